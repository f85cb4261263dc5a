#!/usr/bin/env python3
"""Measured numbers for the five BASELINE.json configurations (seeded stand-ins for the named scenes):
BVH build time, primary and 8xAO (radius 5) trace rates at 1920x1080, on one GPU.  Prints a markdown table."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ntrace_amd as nt
from ntrace_amd import scenes
dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"
def up(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

def measure(name, tri, pos, cam, builder):
    n = tri.shape[0]
    keep = []
    if builder == "sah":
        t0 = time.time(); bvh = nt.sah_build(tri, pos, 1, 1); build = "%.1f s (host SAH)" % (time.time() - t0)
        d_n, d_w, d_i = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index); keep += [d_n, d_w, d_i]
        view = nt.BvhView(d_n.data_ptr(), bvh.nodes.nbytes, d_w.data_ptr(), bvh.woop.nbytes, d_i.data_ptr())
    else:
        capn, capw, capi = nt.lbvh_capacity(n)
        d_tri, d_pos = up(tri), up(pos)
        d_n = torch.zeros(capn, dtype=torch.uint8, device=dev); d_w = torch.zeros(capw, dtype=torch.uint8, device=dev); d_i = torch.zeros(capi, dtype=torch.uint8, device=dev)
        keep += [d_tri, d_pos, d_n, d_w, d_i]
        best = None
        for _ in range(4):
            r = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), pos.min(0), pos.max(0), 8, 0.001, d_n.data_ptr(), capn, d_w.data_ptr(), capw, d_i.data_ptr(), capi)
            best = r if best is None or r.seconds < best.seconds else best
        build = "%.2f ms (device LBVH, %.2f Gtris/s)" % (best.seconds * 1e3, n / best.seconds / 1e9)
        view = nt.BvhView(d_n.data_ptr(), best.nodesBytes, d_w.data_ptr(), best.triWoopBytes, d_i.data_ptr())
    view.validate()
    w, h = 1920, 1080
    rays, _ = scenes.primary_rays(cam, w, h)
    npr = rays.shape[0]; d_rays = up(rays); d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    for _ in range(2): view.trace(K, npr, False, d_rays.data_ptr(), d_res.data_ptr())
    tp = np.median([view.trace(K, npr, False, d_rays.data_ptr(), d_res.data_ptr()) for _ in range(7)])
    hits = nt.count_hits(d_res.data_ptr(), npr)
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, per = 8, (1 << 20) // 8
    ao_t, ao_live = 0.0, 0
    diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
    radius = 5.0 * diag / 4300.0  # config.conf's aoRadius 5 is in Sponza units; scaled to the scene's diagonal
    for lo in range(0, npr, per):
        cnt = min(per, npr - lo)
        b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev); b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
        b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
        nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), lo, cnt, ns, radius, 0xFFF2D5E4)
        view.trace(K, cnt * ns, True, b_rays.data_ptr(), b_res.data_ptr())
        ao_t += float(np.median([view.trace(K, cnt * ns, True, b_rays.data_ptr(), b_res.data_ptr()) for _ in range(3)]))
        ao_live += nt.count_hits(d_res.data_ptr() + lo * 16, cnt) * ns
    print("| %s | %d | %s | %.0f | %.0f | %.1f %% |" % (name, n, build, npr / tp / 1e6, ao_live / ao_t / 1e6 if ao_t > 0 else 0, 100.0 * hits / npr), flush=True)

print("| config (stand-in) | triangles | BVH build | primary Mrays/s | 8xAO Mrays/s | primary hit rate |")
print("|---|---|---|---|---|---|")
tri, pos, cam = scenes.cornell_box(); measure("1 Cornell box", tri, pos, cam, "sah")
tri, pos, cam = scenes.atrium(); measure("2 Sponza (atrium-262k), SAH", tri, pos, cam, "sah"); measure("2 Sponza (atrium-262k), LBVH", tri, pos, cam, "lbvh")
tri, pos, cam = scenes.conference_room(); measure("3 Conference (room-331k), SAH", tri, pos, cam, "sah")
tri, pos, cam = scenes.hairball(); measure("4 Hairball (2.8 M), LBVH", tri, pos, cam, "lbvh")
tri, pos, cam = scenes.courtyard(); measure("5 San Miguel (courtyard-10M), LBVH", tri, pos, cam, "lbvh")
if "--sah-10m" in sys.argv:
    measure("5 San Miguel (courtyard-10M), SAH", tri, pos, cam, "sah")
