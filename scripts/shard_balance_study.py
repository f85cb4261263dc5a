#!/usr/bin/env python3
"""Strong-scaling load balance of one 1080p frame, simulated on ONE GPU: for N = 2, 4, 8 the work of every rank (its primary rays,
the AO batches from its own hits) is traced in turn and the per-rank sum of kernel times recorded; the job's rate is
total rays / MAX over ranks.  Contiguous PixelTable ranges (compact screen regions, what FramePlan did) against ranges interleaved in
chunks of C rays (every N-th chunk).  One JSON line per (N, plan)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402

dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos)
d_nodes, d_woop, d_idx = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
view = nt.BvhView(d_nodes.data_ptr(), bvh.nodes.nbytes, d_woop.data_ptr(), bvh.woop.nbytes, d_idx.data_ptr())
view.validate()
w, h, ns = 1920, 1080, 8
rays, _ = scenes.primary_rays(cam, w, h)
n = rays.shape[0]
full = up(rays).view(n, 32)
d_nrm = up(scenes.tri_normals(tri, pos))


def rank_time(idx):
    """Sum of kernel times (primary + AO batches of <= 2^20 rays) for the primary slots `idx` (int64 tensor)."""
    m = idx.numel()
    r = full[idx].contiguous()
    res = torch.zeros(m * 16, dtype=torch.uint8, device=dev)
    best = None
    per = (1 << 20) // ns
    ao = []
    view.trace(K, m, False, r.data_ptr(), res.data_ptr())
    for first in range(0, m, per):
        cnt = min(per, m - first)
        b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
        b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
        b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
        nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), r.data_ptr(), res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, 5.0, 0xFFF2D5E4)
        ao.append((b_rays, b_res, cnt * ns))
    for _ in range(4):
        t = view.trace(K, m, False, r.data_ptr(), res.data_ptr())
        for (br, bs, k) in ao:
            t += view.trace(K, k, True, br.data_ptr(), bs.data_ptr())
        best = t if best is None or t < best else best
    return best


one = rank_time(torch.arange(n, device=dev))
print(json.dumps(dict(ranks=1, plan="whole frame", ms=one * 1e3)), flush=True)
for N in (2, 4, 8):
    plans = {"contiguous": [torch.arange(n * r // N // 64 * 64, (n * (r + 1) // N // 64 * 64) if r < N - 1 else n, device=dev) for r in range(N)]}
    for C in (1024, 4096, 16384):
        chunks = torch.arange(n, device=dev).split(C)
        plans["interleaved-%d" % C] = [torch.cat(chunks[r::N]) for r in range(N)]
    for name, parts in plans.items():
        ts = [rank_time(p) for p in parts]
        print(json.dumps(dict(ranks=N, plan=name, max_ms=max(ts) * 1e3, mean_ms=float(np.mean(ts)) * 1e3, min_ms=min(ts) * 1e3,
                              speedup_vs_one=one / max(ts), efficiency=one / max(ts) / N)), flush=True)
