#!/usr/bin/env python3
"""A/B of the ageing wave priority (NTR_TRACE_AGE_SHIFT) of the per-ray kernel: 1080p primary batch and one 2^20-ray AO batch on
atrium-262k, interleaved rounds, with and without the dispatch-order prediction."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402

dev = torch.device("cuda:0")


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos)
d_nodes, d_woop, d_idx = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
view = nt.BvhView(d_nodes.data_ptr(), bvh.nodes.nbytes, d_woop.data_ptr(), bvh.woop.nbytes, d_idx.data_ptr())
view.validate()
rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]
d_rays = up(rays)
d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
K = "fermi_speculative_while_while"
view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr())
ref = d_res.clone()
d_nrm = up(scenes.tri_normals(tri, pos))
cnt, ns = (1 << 20) // 8, 8
b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), 0, cnt, ns, 5.0, 0xFFF2D5E4)
configs = []
for pred in (1, 0):
    for shift in (0, 4, 5, 6, 7, 8):
        configs.append(dict(NTR_TRACE_PREDICT=pred, NTR_TRACE_AGE_SHIFT=shift))
times = {i: ([], []) for i in range(len(configs))}
for rnd in range(7):
    for i, cfg in enumerate(configs):
        nt.set_tunables(**cfg)
        tp = view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr())
        ta = view.trace(K, cnt * ns, True, b_rays.data_ptr(), b_res.data_ptr())
        if rnd:
            times[i][0].append(tp)
            times[i][1].append(ta)
        if rnd == 1:
            assert torch.equal(d_res, ref), cfg
for i, cfg in enumerate(configs):
    print(json.dumps(dict(cfg=cfg, primary_us=float(np.median(times[i][0])) * 1e6, ao_us=float(np.median(times[i][1])) * 1e6)), flush=True)
