#!/usr/bin/env python3
"""Per-wave timelines of the per-ray and the dynamic-fetch kernel on divergent batches (device LBVH scenes): when does the
ray pool run dry (first wave to end), how long is the tail after it, how many waves are resident over time.
Needs the experiment build: make -C ntrace_amd/csrc exp; NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_exp.so python3 scripts/studies/divergence_timeline.py courtyard
usage: divergence_timeline.py <scene> [batch ...]      batches: primary diffuse incoherent"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")


def resident(s, e, bins=40):
    t1 = e.max()
    w = t1 / bins
    out = []
    for k in range(bins):
        a, b = k * w, (k + 1) * w
        out.append(int(round(np.clip(np.minimum(e, b) - np.maximum(s, a), 0, None).sum() / w)))
    return out


def main():
    scene = sys.argv[1]
    want = sys.argv[2:] or ["diffuse", "incoherent"]
    tri, pos, cam = scene_of(scene)
    best, bufs = lbvh(tri, pos, 2)
    view = nt.BvhView(bufs[0].data_ptr(), best.nodesBytes, bufs[1].data_ptr(), best.triWoopBytes, bufs[2].data_ptr())
    view.validate()
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    npr = rays.shape[0]
    d_rays = up(rays)
    d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    view.trace("fermi_speculative_while_while", npr, False, d_rays.data_ptr(), d_res.data_ptr())
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, cnt = 8, (1 << 20) // 8
    first = min(900000, npr - cnt)
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, cam["far"],
                 0xFFF2D5E4)
    nr = 1 << 21
    batches = {"primary": (npr, d_rays), "diffuse": (cnt * ns, b_rays), "incoherent": (nr, up(scenes.box_rays(pos, nr, seed=21)))}
    nt.set_tunables(NTR_TRACE_PREDICT=0)
    for bname in want:
        n, dr = batches[bname]
        d_o = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        for kernel, env in (("fermi_speculative_while_while", {"NTR_TRACE_PERRAY_UNIFIED": "0"}),
                            ("fermi_speculative_while_while", {}),
                            ("kepler_dynamic_fetch", {})):
            nt.set_tunables(**env)
            persistent = not kernel.startswith("fermi")
            per = 6 if persistent else 3
            bpc = int(env.get("NTR_TRACE_BLOCKS_PER_CU", 6))
            nw = (256 * bpc * 4) if persistent else ((n + 255) // 256) * 4
            tl = torch.zeros(nw * per, dtype=torch.int64, device=dev)
            view.trace(kernel, n, False, dr.data_ptr(), d_o.data_ptr())
            plain = min(view.trace(kernel, n, False, dr.data_ptr(), d_o.data_ptr()) for _ in range(3))
            nt.experiment_hooks(timeline=tl.data_ptr())
            sec = view.trace(kernel, n, False, dr.data_ptr(), d_o.data_ptr())
            nt.experiment_hooks()
            t = tl.cpu().numpy().reshape(-1, per)
            t = t[t[:, 0] > 0]
            s, e = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
            t0 = s.min()
            s, e = (s - t0) / 100.0, (e - t0) / 100.0
            life = e - s
            out = dict(scene=scene, batch=bname, kernel=kernel, env=env, us_plain=plain * 1e6, us_instrumented=sec * 1e6, waves=int(len(s)),
                       life_us=dict(mean=float(life.mean()), p50=float(np.median(life)), p99=float(np.percentile(life, 99)), max=float(life.max())),
                       last_start_us=float(s.max()), first_end_us=float(e.min()), p10_end_us=float(np.percentile(e, 10)),
                       p50_end_us=float(np.percentile(e, 50)), p90_end_us=float(np.percentile(e, 90)), last_end_us=float(e.max()),
                       resident_waves_40bins=resident(s, e))
            if persistent:
                out["refill"] = dict(per_wave=float(t[:, 3].mean()), rays_per_refill=float(t[:, 4].sum() / max(t[:, 3].sum(), 1)),
                                     share_of_life=float(t[:, 2].mean() / max(life.mean() * 2400.0, 1)))
            print(json.dumps(out), flush=True)
            nt.set_tunables(**{k: None for k in env})


if __name__ == "__main__":
    main()
