#!/usr/bin/env python3
"""In-process A/B of tracer variants and tunables on the bench workload (interleaved rounds,
median + min of per-launch HIP-event times).  Usage: python scripts/studies/trace_sweep.py [--ao-radius R]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ao-radius", type=float, default=5.0)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--configs", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    tri, pos, cam = scenes.atrium()
    bvh = nt.sah_build(tri, pos)
    d_nodes, d_woop, d_idx = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
    view = nt.BvhView(d_nodes.data_ptr(), bvh.nodes.nbytes, d_woop.data_ptr(), bvh.woop.nbytes, d_idx.data_ptr())
    view.validate(stream)
    w, h = 1920, 1080
    n = w * h
    d_tab = torch.zeros(n, dtype=torch.int32, device=dev)
    nt.pixel_table(w, h, d_tab.data_ptr(), 0, stream)
    d_rays = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_a = torch.zeros(n, dtype=torch.int32, device=dev)
    nt.raygen_primary(d_rays.data_ptr(), d_a.data_ptr(), d_a.data_ptr(), d_tab.data_ptr(), cam["eye"],
                      scenes.nscreen_to_world(cam, w, h), w, h, cam["far"], 0, stream)
    view.trace("fermi_speculative_while_while", n, False, d_rays.data_ptr(), d_res.data_ptr(), stream)
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, per = 8, (1 << 20) // 8
    ao = []
    for lo in range(0, n, per):
        cnt = min(per, n - lo)
        br = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
        bs = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
        ba = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
        nt.raygen_ao(br.data_ptr(), ba.data_ptr(), ba.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(),
                     lo, cnt, ns, args.ao_radius, 0xFFF2D5E4, stream)
        live = nt.count_hits(d_res.data_ptr() + lo * 16, cnt, stream) * ns
        ao.append((br, bs, cnt * ns, live))
    ao_live = sum(a[3] for a in ao)

    print("bvh flags", view.flags)
    configs = [("perray", "fermi_speculative_while_while", {})]
    for ls in (4, 8, 16, 24, 32, 40, 48, 56, 65):
        configs.append(("perray ls%d" % ls, "fermi_speculative_while_while", {"NTR_TRACE_LEAF_SWITCH": ls}))
    for c in (32, 64, 128):
        for t in (0, 24, 40, 56):
            for b in (7,):
                configs.append(("persist c%d t%d b%d" % (c, t, b), "kepler_dynamic_fetch",
                                {"NTR_TRACE_CHUNK": c, "NTR_TRACE_FETCH_THRESHOLD": t, "NTR_TRACE_BLOCKS_PER_CU": b}))
    for (c, t, b, ls) in ((64, 0, 7, 32), (64, 32, 7, 32), (64, 48, 7, 32), (32, 40, 7, 32), (64, 40, 7, 16), (64, 56, 7, 48)):
        configs.append(("persist c%d t%d b%d ls%d" % (c, t, b, ls), "kepler_dynamic_fetch",
                        {"NTR_TRACE_CHUNK": c, "NTR_TRACE_FETCH_THRESHOLD": t, "NTR_TRACE_BLOCKS_PER_CU": b, "NTR_TRACE_LEAF_SWITCH": ls}))
    configs.append(("persist c64 t40 b4", "kepler_dynamic_fetch", {"NTR_TRACE_CHUNK": 64, "NTR_TRACE_FETCH_THRESHOLD": 40, "NTR_TRACE_BLOCKS_PER_CU": 4}))
    configs.append(("persist c64 t40 b5", "kepler_dynamic_fetch", {"NTR_TRACE_CHUNK": 64, "NTR_TRACE_FETCH_THRESHOLD": 40, "NTR_TRACE_BLOCKS_PER_CU": 5}))
    if args.configs:
        keep = set(args.configs.split(","))
        configs = [c for c in configs if c[0] in keep]
    tunables = ("NTR_TRACE_LEAF_SWITCH", "NTR_TRACE_CHUNK", "NTR_TRACE_FETCH_THRESHOLD", "NTR_TRACE_BLOCKS_PER_CU")
    times = {c[0]: ([], []) for c in configs}
    for rnd in range(args.rounds + 1):
        for name, kernel, env in configs:
            for t in tunables:
                os.environ.pop(t, None)
            for k, v in env.items():
                if not k.startswith("_"):
                    os.environ[k] = str(v)
            nt.set_tunables()  # the library reads the environment once: make it re-read
            flags = env.get("_flags", None)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            view.trace(kernel, n, False, d_rays.data_ptr(), d_res.data_ptr(), stream, False, flags)
            e[1].record()
            for (br, bs, cnt, live) in ao:
                view.trace(kernel, cnt, True, br.data_ptr(), bs.data_ptr(), stream, False, flags)
            e[2].record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[name][0].append(e[0].elapsed_time(e[1]))
                times[name][1].append(e[1].elapsed_time(e[2]))
    print("%-24s %10s %10s %12s %12s" % ("config", "prim ms", "ao ms", "prim Mray/s", "ao Mray/s"))
    for name, _, _ in configs:
        p, a = np.median(times[name][0]), np.median(times[name][1])
        print("%-24s %10.3f %10.3f %12.0f %12.0f   (min %.3f / %.3f)" % (name, p, a, n / p / 1e3, ao_live / a / 1e3,
                                                                        min(times[name][0]), min(times[name][1])))


if __name__ == "__main__":
    main()
