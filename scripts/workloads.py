#!/usr/bin/env python3
"""Small fixed workloads for rocprofv3 runs (scripts/profile_round.sh), one per sub-command:

  lbvh  <scene> [reps]            on-device LBVH builds of atrium (262 k) / hairball (2.8 M) / courtyard (10 M)
  trace <scene> <kernel> [reps]   1080p primary batch + one 2^20-ray AO batch + 2^21 incoherent rays (uniform in the
                                  bounding box), SAH BVH for atrium, device LBVH otherwise
                                  (the 10 M-triangle LBVH is 0.75 GB: three times the 256 MB Infinity Cache, the
                                  HBM-resident roofline point)

Each prints one JSON line with what the GPU-side counters have to be compared with (algorithmic bytes from the
instrumented kernel, launch times by HIP events)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402

dev = torch.device("cuda:0")


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def scene_of(name):
    return {"atrium": scenes.atrium, "hairball": scenes.hairball, "courtyard": scenes.courtyard,
            "conference": scenes.conference_room}[name]()


def lbvh(tri, pos, reps):
    n = tri.shape[0]
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    dn = torch.zeros(capn, dtype=torch.uint8, device=dev)
    dw = torch.zeros(capw, dtype=torch.uint8, device=dev)
    di = torch.zeros(capi, dtype=torch.uint8, device=dev)
    best = None
    for _ in range(reps):
        r = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), pos.min(0), pos.max(0), 8, 0.001, dn.data_ptr(), capn,
                          dw.data_ptr(), capw, di.data_ptr(), capi)
        best = r if best is None or r.seconds < best.seconds else best
    return best, (dn, dw, di, d_tri, d_pos)


def main():
    what, scene = sys.argv[1], sys.argv[2]
    tri, pos, cam = scene_of(scene)
    n = tri.shape[0]
    if what == "lbvh":
        reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
        best, _ = lbvh(tri, pos, reps)
        ni, nl = best.numNodes, best.numLeaves
        alg = n * (56 + 4 * 20 + 96 + 112 + 48) + ni * (28 + 48) + nl * 20 + max(ni - 1, 0) * 96
        print(json.dumps(dict(workload="lbvh", scene=scene, triangles=n, reps=reps, best=best.as_dict(), algorithmic_bytes=alg,
                              hbm_frac=alg / best.seconds / 8e12)))
        return
    kernel = sys.argv[3]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    keep = []
    if scene == "atrium":
        bvh = nt.sah_build(tri, pos, 1, 1)
        d_n, d_w, d_i = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
        view = nt.BvhView(d_n.data_ptr(), bvh.nodes.nbytes, d_w.data_ptr(), bvh.woop.nbytes, d_i.data_ptr())
        bvh_bytes = bvh.nodes.nbytes + bvh.woop.nbytes + bvh.tri_index.nbytes
        keep += [d_n, d_w, d_i]
    else:
        best, bufs = lbvh(tri, pos, 2)
        keep += list(bufs)
        view = nt.BvhView(bufs[0].data_ptr(), best.nodesBytes, bufs[1].data_ptr(), best.triWoopBytes, bufs[2].data_ptr())
        bvh_bytes = best.nodesBytes + best.triWoopBytes + best.triIndexBytes
    view.validate()
    w, h = 1920, 1080
    only = os.environ.get("WL_ONLY", "")   # "incoherent": only that batch (the persistent kernels' grid is the same for every batch,
    if only == "incoherent":                # so per-batch PMC means need a run of their own)
        nr = 1 << 21
        d_rr = up(scenes.box_rays(pos, nr, seed=21))
        d_ro = torch.zeros(nr * 16, dtype=torch.uint8, device=dev)
        view.trace(kernel, nr, False, d_rr.data_ptr(), d_ro.data_ptr())
        tr = [view.trace(kernel, nr, False, d_rr.data_ptr(), d_ro.data_ptr()) for _ in range(reps)]
        print(json.dumps(dict(workload="trace", scene=scene, kernel=kernel, only=only, incoherent=dict(rays=nr, ms_mean=float(np.mean(tr)) * 1e3,
                                                                                                     ms_min=float(np.min(tr)) * 1e3))))
        return
    rays, _ = scenes.primary_rays(cam, w, h)
    npr = rays.shape[0]
    d_rays = up(rays)
    d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    view.trace(kernel, npr, False, d_rays.data_ptr(), d_res.data_ptr())
    if only in ("diffuse", "ao_frame"):
        # BASELINE config 4 / 5's secondary batches, the whole frame: 16 batches of <= 2^20 rays from the primary hits, each traced once cold
        # and `reps` times (the reference's protocol re-traces a batch): per-dispatch PMC means of the working kernel symbol x 16 launches =
        # the frame's HBM traffic (bench.py extras.configs[*].roofline.traffic)
        d_nrm = up(scenes.tri_normals(tri, pos))
        ns, per = 8, (1 << 20) // 8
        diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
        any_hit = only == "ao_frame"
        dist_ = 5.0 * diag / 4300.0 if any_hit else cam["far"]
        b_rays = torch.zeros(per * ns * 32, dtype=torch.uint8, device=dev)
        b_res = torch.zeros(per * ns * 16, dtype=torch.uint8, device=dev)
        b_a = torch.zeros(per * ns, dtype=torch.int32, device=dev)
        tot, alg, nb = 0.0, 0, 0
        for lo in range(0, npr, per):
            cnt = min(per, npr - lo)
            nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), lo, cnt, ns, dist_, 0xFFF2D5E4)
            view.trace(kernel, cnt * ns, any_hit, b_rays.data_ptr(), b_res.data_ptr())
            tot += float(np.median([view.trace(kernel, cnt * ns, any_hit, b_rays.data_ptr(), b_res.data_ptr()) for _ in range(reps)]))
            alg += view.trace_stats(kernel, cnt * ns, any_hit, b_rays.data_ptr(), b_res.data_ptr()).algorithmic_bytes()
            nb += 1
        print(json.dumps(dict(workload="trace", scene=scene, kernel=kernel, only=only, batches=nb, launches_per_batch=reps + 1, frame_ms=tot * 1e3,
                              algorithmic_bytes=alg, hbm_frac=alg / tot / 8e12)))
        return
    tp = [view.trace(kernel, npr, False, d_rays.data_ptr(), d_res.data_ptr()) for _ in range(reps)]
    st = view.trace_stats(kernel, npr, False, d_rays.data_ptr(), d_res.data_ptr())
    # one AO batch (2^20 rays, any hit) from the first 131 072 primary hits
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, cnt = 8, (1 << 20) // 8
    diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
    radius = 5.0 if scene == "atrium" else 5.0 * diag / 4300.0
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), 0, cnt, ns, radius,
                 0xFFF2D5E4)
    view.trace(kernel, cnt * ns, True, b_rays.data_ptr(), b_res.data_ptr())
    ta = [view.trace(kernel, cnt * ns, True, b_rays.data_ptr(), b_res.data_ptr()) for _ in range(reps)]
    sa = view.trace_stats(kernel, cnt * ns, True, b_rays.data_ptr(), b_res.data_ptr())
    # incoherent batch: 2^21 random rays through the bounding box (closest hit) -- every ray in its own part of the BVH
    nr = 1 << 21
    d_rr = up(scenes.box_rays(pos, nr, seed=21))
    d_ro = torch.zeros(nr * 16, dtype=torch.uint8, device=dev)
    view.trace(kernel, nr, False, d_rr.data_ptr(), d_ro.data_ptr())
    tr = [view.trace(kernel, nr, False, d_rr.data_ptr(), d_ro.data_ptr()) for _ in range(reps)]
    sr = view.trace_stats(kernel, nr, False, d_rr.data_ptr(), d_ro.data_ptr())
    print(json.dumps(dict(workload="trace", scene=scene, kernel=kernel, triangles=n, bvh_bytes=bvh_bytes, bvh_flags=view.flags,
                          incoherent=dict(rays=nr, ms_mean=float(np.mean(tr)) * 1e3, ms_min=float(np.min(tr)) * 1e3, stats=sr.as_dict(),
                                          algorithmic_bytes=sr.algorithmic_bytes()),
                          primary=dict(rays=npr, ms_mean=float(np.mean(tp)) * 1e3, ms_min=float(np.min(tp)) * 1e3, stats=st.as_dict(),
                                       algorithmic_bytes=st.algorithmic_bytes()),
                          ao=dict(rays=cnt * ns, ms_mean=float(np.mean(ta)) * 1e3, ms_min=float(np.min(ta)) * 1e3, stats=sa.as_dict(),
                                  algorithmic_bytes=sa.algorithmic_bytes()))))


if __name__ == "__main__":
    main()
