#!/usr/bin/env python3
"""Experiment: block order from an a-priori cost PREDICTOR (no feedback): per block, the number of BVH
nodes of depth <= D whose boxes a sample ray segment intersects.  Predictor computed on the host here
(numpy); the launch is re-timed under the derived order (NTR_TRACE_ORDER)."""
import os, sys
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
nodes = bvh.nodes.view(np.float32).reshape(-1, 16); nodesi = bvh.nodes.view(np.int32).reshape(-1, 16)

def top_boxes(D):
    out, cur = [], [0]
    for dep in range(D):
        nxt = []
        for ni in cur:
            nf, nI = nodes[ni // 64], nodesi[ni // 64]
            for k in range(2):
                out.append((nf[4 * k], nf[4 * k + 1], nf[4 * k + 2], nf[4 * k + 3], nf[8 + 2 * k], nf[9 + 2 * k]))
                if nI[12 + k] >= 0:
                    nxt.append(nI[12 + k])
        cur = nxt
    return np.array(out, dtype=np.float32)

def predict(R, F):
    o = np.stack([R["ox"], R["oy"], R["oz"]], 1); d = np.stack([R["dx"], R["dy"], R["dz"]], 1)
    inv = (1.0 / np.where(d == 0, np.float32(1e-30), d)).astype(np.float32)
    cnt = np.zeros(len(R))
    for s in range(0, len(F), 128):
        f = F[s:s + 128]
        lo, hi = f[:, [0, 2, 4]], f[:, [1, 3, 5]]
        t0 = (lo[None] - o[:, None]) * inv[:, None]; t1 = (hi[None] - o[:, None]) * inv[:, None]
        tn = np.maximum(np.minimum(t0, t1).max(2), R["tmin"][:, None]); tf = np.minimum(np.maximum(t0, t1).min(2), R["tmax"][:, None])
        cnt += (tn <= tf).sum(1)
    return cnt

def experiment(name, rays, d_rays, n, any_hit):
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    nb = (n + 255) // 256
    def timed(reps=9):
        return np.median([view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr()) for _ in range(reps)]) * 1e6
    for _ in range(3): view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr())
    out = {"natural": timed()}
    ref = d_res.clone()
    for D in (8, 10, 12):
        F = top_boxes(D)
        for samples in ((100,), (30, 160)):
            idx = [np.minimum(np.arange(nb) * 256 + s, n - 1) for s in samples]
            p = np.max([predict(rays[i], F) for i in idx], 0)
            for C in (16, 64):
                b = np.minimum((p / max(p.max(), 1) * C).astype(np.int64), C - 1)
                o = np.argsort(-b, kind="stable")
                d_o = up(o.astype(np.uint32))
                nt.experiment_hooks(order=d_o.data_ptr())
                view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr())
                out["D%d_%dbox_s%d_c%d" % (D, len(F), len(samples), C)] = timed()
                nt.experiment_hooks()
                assert torch.equal(d_res, ref)
    print(name, n, {k: round(float(v), 1) for k, v in out.items()})

rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]; d_rays = up(rays)
experiment("primary", rays, d_rays, n, False)
d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
for radius in (5.0, 200.0):
    cnt, ns = (1 << 20) // 8, 8
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), 0, cnt, ns, radius, 0xFFF2D5E4)
    torch.cuda.synchronize()
    experiment("ao_r%g" % radius, b_rays.cpu().numpy().view(nt.RAY_DTYPE), b_rays, cnt * ns, True)
