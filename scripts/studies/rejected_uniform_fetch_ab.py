#!/usr/bin/env python3
"""REJECTED (kept for the record; the kernel no longer has the switch -- commit c766c93..): A/B of the scalar wave-uniform node
fetch (NTR_TRACE_UNIFORM) on the bench workload: 1080p primary batch and one 2^20-ray AO batch
on atrium-262k (SAH), interleaved rounds, median kernel time by HIP events; hit records must be identical."""
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
view.trace("fermi_speculative_while_while", n, False, d_rays.data_ptr(), d_res.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
cnt, ns = (1 << 20) // 8, 8
b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), 0, cnt, ns, 5.0, 0xFFF2D5E4)

cfgs = [dict(NTR_TRACE_UNIFORM=0, NTR_TRACE_OCTANT=0), dict(NTR_TRACE_UNIFORM=1, NTR_TRACE_OCTANT=0), dict(NTR_TRACE_UNIFORM=1, NTR_TRACE_OCTANT=1), dict(NTR_TRACE_UNIFORM=0, NTR_TRACE_OCTANT=1)]
kernels = sys.argv[1:] or ["fermi_speculative_while_while", "tesla_persistent_while_while"]
for kernel in kernels:
    times = {i: dict(primary=[], ao=[]) for i in range(len(cfgs))}
    ref = {}
    for rnd in range(7):
        for i, cfg in enumerate(cfgs):
            nt.set_tunables(**cfg)
            times[i]["primary"].append(view.trace(kernel, n, False, d_rays.data_ptr(), d_res.data_ptr()))
            torch.cuda.synchronize()
            h = d_res.cpu().numpy().tobytes()
            times[i]["ao"].append(view.trace(kernel, cnt * ns, True, b_rays.data_ptr(), b_res.data_ptr()))
            torch.cuda.synchronize()
            h2 = b_res.cpu().numpy().view(np.int32).reshape(-1, 4)[:, 0] >= 0   # any-hit: which triangle is free, hit / miss is not
            if rnd == 0:
                ref[i] = (h, h2.tobytes())
    for i, cfg in enumerate(cfgs):
        print(json.dumps(dict(kernel=kernel, cfg=cfg, primary_us=float(np.median(times[i]["primary"][1:])) * 1e6,
                              ao_us=float(np.median(times[i]["ao"][1:])) * 1e6, same_primary_records=ref[i][0] == ref[0][0],
                              same_ao_hits=ref[i][1] == ref[0][1])), flush=True)
nt.set_tunables(NTR_TRACE_UNIFORM=None, NTR_TRACE_OCTANT=None)
