#!/usr/bin/env python3
"""Diagnostic: where the persistent kernel's waves spend their time (refill vs traversal)."""
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
bvh = nt.sah_build(tri, pos)
d_nodes, d_woop, d_idx = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
view = nt.BvhView(d_nodes.data_ptr(), bvh.nodes.nbytes, d_woop.data_ptr(), bvh.woop.nbytes, d_idx.data_ptr()); view.validate()
rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]; d_rays = up(rays); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
for (c, t, b) in ((64, 0, 7), (64, 40, 7), (32, 48, 7), (64, 56, 4)):
    os.environ.update(NTR_TRACE_CHUNK=str(c), NTR_TRACE_FETCH_THRESHOLD=str(t), NTR_TRACE_BLOCKS_PER_CU=str(b))
    nw = 256 * b * 4
    tl = torch.zeros(nw * 6, dtype=torch.int64, device=dev)
    for _ in range(2): view.trace("kepler_dynamic_fetch", n, False, d_rays.data_ptr(), d_res.data_ptr())
    nt.experiment_hooks(timeline=tl.data_ptr())
    sec = view.trace("kepler_dynamic_fetch", n, False, d_rays.data_ptr(), d_res.data_ptr())
    nt.experiment_hooks()
    t_ = tl.cpu().numpy().reshape(-1, 6)
    t_ = t_[t_[:, 0] > 0]
    s, e = t_[:, 0], t_[:, 1]
    t0 = s.min(); s = (s - t0) / 100.0; e = (e - t0) / 100.0
    life_us = (e - s)
    print("chunk %d thr %d blocks/CU %d: kernel %.0f us; waves %d, life mean %.0f max %.0f min-end %.0f us; refills/wave %.1f rays/refill %.1f refill cycles/wave %.0f (%.1f%% of life at 2.4GHz)" % (
        c, t, b, sec * 1e6, len(s), life_us.mean(), life_us.max(), e.min(), t_[:, 3].mean(), t_[:, 4].sum() / max(t_[:, 3].sum(), 1), t_[:, 2].mean(),
        100.0 * t_[:, 2].mean() / (life_us.mean() * 2400.0)))
