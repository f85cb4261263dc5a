#!/usr/bin/env python3
"""Diagnostic: per-wave start/end stamps of one primary launch -> average / peak resident waves."""
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
nw = (n + 63) // 64
tl = torch.zeros(nw * 3, dtype=torch.int64, device=dev)
for _ in range(3): view.trace("fermi_speculative_while_while", n, False, d_rays.data_ptr(), d_res.data_ptr())
nt.experiment_hooks(timeline=tl.data_ptr())
sec = view.trace("fermi_speculative_while_while", n, False, d_rays.data_ptr(), d_res.data_ptr())
nt.experiment_hooks()
t = tl.cpu().numpy().reshape(-1, 3)
s, e, hw = t[:, 0], t[:, 1], t[:, 2]
t0 = s.min(); s = (s - t0) / 100.0; e = (e - t0) / 100.0   # us (100 MHz realtime counter)
print("kernel %.1f us (event), wave span %.1f us; wave life mean %.1f us p50 %.1f p99 %.1f max %.1f" % (sec * 1e6, e.max(), (e - s).mean(), np.median(e - s), np.percentile(e - s, 99), (e - s).max()))
grid = np.linspace(0, e.max(), 61)
for a, b in zip(grid[:-1:4], grid[4::4]):
    mid = 0.5 * (a + b)
    print("t=%6.1f us resident waves %6d  (%.2f per SIMD)" % (mid, ((s <= mid) & (e > mid)).sum(), ((s <= mid) & (e > mid)).sum() / 1024.0))
print("avg resident waves/SIMD over kernel: %.2f" % ((e - s).sum() / e.max() / 1024.0))
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7
print("last wave start %.1f us" % s.max())
