#!/usr/bin/env python3
"""Diagnostic for the persistent-waves kernel against the per-ray kernel: per-wave start / end stamps of
one launch of each kernel on the 1080p primary batch and on one 2^20-ray AO batch of atrium-262k, reduced to
  * the number of waves resident over time (10 us bins) -- is the machine full?
  * wave lifetimes, and for the persistent kernel the share of a wave's life spent refilling.
Needs the experiment build: make -C ntrace_amd/csrc exp; NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_exp.so python scripts/studies/persist_diag.py"""
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
nt.set_tunables(NTR_TRACE_PREDICT=0)


def resident(s, e, bin_us=10.0):
    t1 = e.max()
    bins = int(t1 / bin_us) + 1
    out = np.zeros(bins)
    for k in range(bins):
        a, b = k * bin_us, (k + 1) * bin_us
        out[k] = np.clip(np.minimum(e, b) - np.maximum(s, a), 0, None).sum() / bin_us
    return out


def run(kernel, m, any_hit, r, o, tunables=None, label=""):
    nt.set_tunables(**(tunables or {}))
    persistent = not kernel.startswith("fermi")
    per = 6 if persistent else 3
    bpc = int((tunables or {}).get("NTR_TRACE_BLOCKS_PER_CU", 6))
    nw = (256 * bpc * 4) if persistent else ((m + 255) // 256) * 4
    tl = torch.zeros(nw * per, dtype=torch.int64, device=dev)
    for _ in range(2):
        view.trace(kernel, m, any_hit, r.data_ptr(), o.data_ptr())
    plain = float(np.median([view.trace(kernel, m, any_hit, r.data_ptr(), o.data_ptr()) for _ in range(5)]))
    nt.experiment_hooks(timeline=tl.data_ptr())
    sec = view.trace(kernel, m, any_hit, r.data_ptr(), o.data_ptr())
    nt.experiment_hooks()
    t = tl.cpu().numpy().reshape(-1, per)
    t = t[t[:, 0] > 0]
    s, e = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
    t0 = s.min()
    s, e = (s - t0) / 100.0, (e - t0) / 100.0  # s_memrealtime ticks of 10 ns
    res = resident(s, e)
    life = e - s
    out = dict(kernel=kernel, label=label, rays=m, any_hit=any_hit, us_plain=plain * 1e6, us_instrumented=sec * 1e6, waves=int(len(s)),
               life_us=dict(mean=float(life.mean()), p50=float(np.median(life)), p99=float(np.percentile(life, 99)), max=float(life.max())),
               last_start_us=float(s.max()), first_end_us=float(e.min()), last_end_us=float(e.max()),
               resident_waves_per_10us=[int(round(x)) for x in res])
    if persistent:
        out["refill"] = dict(per_wave=float(t[:, 3].mean()), rays_per_refill=float(t[:, 4].sum() / max(t[:, 3].sum(), 1)),
                             cycles_per_wave=float(t[:, 2].mean()), share_of_life=float(t[:, 2].mean() / max(life.mean() * 2400.0, 1)))
    if tunables:
        out["tunables"] = tunables
        nt.set_tunables(**{k: None for k in tunables})
    print(json.dumps(out), flush=True)


for (name, m, ah, r, o) in (("primary", n, False, d_rays, d_res), ("ao", cnt * ns, True, b_rays, b_res)):
    run("fermi_speculative_while_while", m, ah, r, o, label=name)
    run("tesla_persistent_while_while", m, ah, r, o, label=name)
    for tun in ({"NTR_TRACE_POOL_HEADS": 8}, {"NTR_TRACE_POOL_HEADS": 128}, {"NTR_TRACE_POOL_HEADS": 256}, {"NTR_TRACE_POOL_HEADS": 512}, {"NTR_TRACE_POOL_HEADS": 1024},
                {"NTR_TRACE_POOL_HEADS": 256, "NTR_TRACE_BLOCKS_PER_CU": 7}, {"NTR_TRACE_POOL_HEADS": 256, "NTR_TRACE_BLOCKS_PER_CU": 5},
                {"NTR_TRACE_POOL_HEADS": 256, "NTR_TRACE_CHUNK": 128}, {"NTR_TRACE_POOL_HEADS": 256, "NTR_TRACE_FETCH_THRESHOLD": 32},
                {"NTR_TRACE_POOL_HEADS": 64}):
        run("tesla_persistent_while_while", m, ah, r, o, tun, label=name)
