#!/usr/bin/env python3
"""Would the PRIMARY launch's per-block cost predict the cost of the AO blocks generated from the same pixels?  Per-wave lifetimes of
the primary launch and of one 2^20-ray AO batch (experiment build: NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_exp.so), reduced to
per-block costs (longest wave), AO block b (256 rays = 32 pixels) against the primary block of its pixels: rank correlation."""
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
K = "fermi_speculative_while_while"


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def ranks(x):
    r = np.empty(len(x))
    r[np.argsort(x, kind="stable")] = np.arange(len(x))
    return r


tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos)
d_nodes, d_woop, d_idx = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
view = nt.BvhView(d_nodes.data_ptr(), bvh.nodes.nbytes, d_woop.data_ptr(), bvh.woop.nbytes, d_idx.data_ptr())
view.validate()
rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]
d_rays = up(rays)
d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
nt.set_tunables(NTR_TRACE_PREDICT=0)


def wave_life(m, any_hit, r, o):
    nw = ((m + 255) // 256) * 4
    tl = torch.zeros(nw * 3, dtype=torch.int64, device=dev)
    view.trace(K, m, any_hit, r.data_ptr(), o.data_ptr())
    nt.experiment_hooks(timeline=tl.data_ptr())
    view.trace(K, m, any_hit, r.data_ptr(), o.data_ptr())
    nt.experiment_hooks()
    t = tl.cpu().numpy().reshape(-1, 3)
    return (t[:, 1] - t[:, 0]).astype(np.float64) / 100.0   # us


prim = wave_life(n, False, d_rays, d_res).reshape(-1, 4).max(1)          # per primary block of 256 pixels
d_nrm = up(scenes.tri_normals(tri, pos))
ns, cnt = 8, (1 << 20) // 8
out = []
for batch in (0, 5, 10):
    first = batch * cnt
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, 5.0, 0xFFF2D5E4)
    ao = wave_life(cnt * ns, True, b_rays, b_res).reshape(-1, 4).max(1)   # per AO block of 256 rays = 32 pixels
    pb = (first + np.arange(len(ao)) * 32) // 256
    rho = float(np.corrcoef(ranks(ao), ranks(prim[pb]))[0, 1])
    out.append(dict(ao_batch=batch, ao_blocks=int(len(ao)), spearman=rho, ao_cost_us=dict(mean=float(ao.mean()), p99=float(np.percentile(ao, 99)), max=float(ao.max()))))
    print(json.dumps(out[-1]), flush=True)
