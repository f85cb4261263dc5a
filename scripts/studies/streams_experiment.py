#!/usr/bin/env python3
"""Experiment: the 16 AO batches of a frame are independent launches; issue them round-robin on k HIP streams
(tails of one batch overlap the start of the next) and compare wall-clock per step with the single-stream order."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch
import ntrace_amd as nt
from ntrace_amd import scenes
dev = torch.device("cuda:0")
def up(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
d_nodes, d_woop, d_idx = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
view = nt.BvhView(d_nodes.data_ptr(), bvh.nodes.nbytes, d_woop.data_ptr(), bvh.woop.nbytes, d_idx.data_ptr()); view.validate()
K = "fermi_speculative_while_while"
rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]; d_rays = up(rays)
d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
batches = [(n, False, d_rays, d_res)]
ns, per = 8, (1 << 20) // 8
for lo in range(0, n, per):
    cnt = min(per, n - lo)
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev); b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), lo, cnt, ns, 5.0, 0xFFF2D5E4)
    batches.append((cnt * ns, True, b_rays, b_res))
for (m, ah, r, o) in batches:
    view.trace(K, m, ah, r.data_ptr(), o.data_ptr())
torch.cuda.synchronize()
ref = [b[3].clone() for b in batches]
for k in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(k)]
    def step():
        for i, (m, ah, r, o) in enumerate(batches):
            view.trace(K, m, ah, r.data_ptr(), o.data_ptr(), streams[i % k].cuda_stream, False)
    for b in batches[1:]: b[3].zero_()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    ok = all(torch.equal(a, b[3]) for a, b in zip(ref, batches))
    print("streams %d: %.3f ms per step (primary + 16 AO), results identical: %s" % (k, dt * 1e3, ok))
