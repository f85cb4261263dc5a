#!/usr/bin/env python3
"""The headline step (atrium-262k SAH, 1080p primary + 16 AO batches) traced by every kernel name, under a list of environment settings:
   python3 scripts/persist_bench.py [--scenes atrium,hairball,courtyard] [VAR=v,VAR2=v ...]...
Each positional argument is one setting (comma-separated VAR=value pairs, "-" = defaults).  Per (setting, kernel): Mrays/s of the step by
the reference's count, primary / AO milliseconds, and whether every record equals the default selector's.  For the LBVH scenes:
the 2^21 incoherent box rays and (hairball) one diffuse batch, kepler_dynamic_fetch and the default selector."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ntrace_amd as nt
from ntrace_amd import scenes
import bench

dev = torch.device("cuda:0")
K0 = "fermi_speculative_while_while"
NAMES = (K0, "tesla_persistent_while_while", "kepler_dynamic_fetch")


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def med(f, n=5):
    f()
    return float(np.median([f() for _ in range(n)]))


def main():
    argv = sys.argv[1:]
    which = "atrium"
    if argv and argv[0] == "--scenes":
        which = argv[1]
        argv = argv[2:]
    settings = argv or ["-"]
    stream = torch.cuda.current_stream().cuda_stream
    args = bench.parse(["--no-extras"])
    out = []
    if "atrium" in which:
        tri, pos, cam = scenes.atrium()
        bvh = nt.sah_build(tri, pos, 1, 1)
        keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
        view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
        view.validate(stream)
        d_nrm = up(scenes.tri_normals(tri, pos))
        from ntrace_amd import dist as ntd
        frame = bench.Frame(nt, torch, view, lambda d: ntd.FramePlan(1920 * 1080, 0, 1, 8, 1 << 20), cam, 1920, 1080, d_nrm, args, dev, stream, scenes)
        for b in frame.batches:
            view.trace(K0, b["n"], b["any_hit"], b["rays"], b["res"], stream)
        torch.cuda.synchronize()
        ref = [(frame.d_res if b["name"] == "primary" else b["res_t"]).clone() for b in frame.batches]
        for st in settings:
            env = {} if st == "-" else dict(kv.split("=") for kv in st.split(","))
            nt.set_tunables(**env)
            for kn in NAMES:
                ms, eq = [], True
                for bi, b in enumerate(frame.batches):
                    ms.append(med(lambda: view.trace(kn, b["n"], b["any_hit"], b["rays"], b["res"], stream, True), 5) * 1e3)
                    got = frame.d_res if b["name"] == "primary" else b["res_t"]
                    eq = eq and bool(torch.equal(got.view(torch.int32).view(-1, 4)[:, :2], ref[bi].view(torch.int32).view(-1, 4)[:, :2]))
                row = dict(scene="atrium", setting=st, kernel=kn, mrays=frame.rays_per_step / (sum(ms) * 1e-3) / 1e6, primary_ms=ms[0], ao_ms=sum(ms[1:]), records_equal=eq)
                out.append(row)
                print(json.dumps(row), flush=True)
            nt.set_tunables(**{k: None for k in env})
        del frame, keep
    for name, fn in (("hairball", scenes.hairball), ("courtyard", scenes.courtyard)):
        if name not in which:
            continue
        tri, pos, cam = fn()
        lview, best, info, keep = bench.device_lbvh(nt, torch, up, dev, stream, tri, pos, 2, 8000.0)
        nr = 1 << 21
        d_rr = up(scenes.box_rays(pos, nr, seed=21))
        d_ro = torch.zeros(nr * 16, dtype=torch.uint8, device=dev)
        # one diffuse batch (config 4's kind): from the primary hits of the first 2^17 pixels... use the middle of the frame
        rays, _ = scenes.primary_rays(cam, 1920, 1080)
        d_pr = up(rays)
        d_pres = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device=dev)
        lview.trace(K0, rays.shape[0], False, d_pr.data_ptr(), d_pres.data_ptr(), stream)
        d_nrm = up(scenes.tri_normals(tri, pos))
        per = (1 << 20) // 8
        lo = (rays.shape[0] // 2) // per * per
        b_rays = torch.zeros(per * 8 * 32, dtype=torch.uint8, device=dev)
        b_res = torch.zeros(per * 8 * 16, dtype=torch.uint8, device=dev)
        b_a = torch.zeros(per * 8, dtype=torch.int32, device=dev)
        nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_pr.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), lo, per, 8, cam["far"], 0xFFF2D5E4, stream)
        refs = {}
        for st in settings:
            env = {} if st == "-" else dict(kv.split("=") for kv in st.split(","))
            nt.set_tunables(**env)
            for kn in (K0, "kepler_dynamic_fetch"):
                t_box = med(lambda: lview.trace(kn, nr, False, d_rr.data_ptr(), d_ro.data_ptr(), stream, True), 3) * 1e3
                e1 = refs.setdefault("box", d_ro.clone())
                eq = bool(torch.equal(d_ro.view(torch.int32).view(-1, 4)[:, :2], e1.view(torch.int32).view(-1, 4)[:, :2]))
                t_dif = med(lambda: lview.trace(kn, per * 8, False, b_rays.data_ptr(), b_res.data_ptr(), stream, True), 3) * 1e3
                e2 = refs.setdefault("dif", b_res.clone())
                eq = eq and bool(torch.equal(b_res.view(torch.int32).view(-1, 4)[:, :2], e2.view(torch.int32).view(-1, 4)[:, :2]))
                t_prim = med(lambda: lview.trace(kn, rays.shape[0], False, d_pr.data_ptr(), d_pres.data_ptr(), stream, True), 3) * 1e3
                row = dict(scene=name, setting=st, kernel=kn, box_rays_ms=t_box, diffuse_batch_ms=t_dif, primary_ms=t_prim, records_equal=eq)
                out.append(row)
                print(json.dumps(row), flush=True)
            nt.set_tunables(**{k: None for k in env})
        del keep, lview


if __name__ == "__main__":
    main()
