#!/usr/bin/env python3
"""What would starting the long rays first be worth now that a ray is no longer one chain?  The box rays of a scene are permuted on the
host by their TRUE step counts (the oracle's counting tracer), longest first -- the bound of any per-ray cost prediction or feedback --,
by step counts blurred with log-normal noise, and with only the longest tenth moved to the front; each order is traced by
kepler_dynamic_fetch (ray splitting on) and by the per-ray launch.  Records are compared through the permutation.
usage: lpt_upper_bound.py <scene> [rays, default 2^21]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import up, lbvh, scene_of  # noqa: E402
from tail_handoff_model import per_ray_steps  # noqa: E402  (analysis script: the oracle's counting tracer)

dev = torch.device("cuda:0")
scene = sys.argv[1] if len(sys.argv) > 1 else "courtyard"
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
tri, pos, cam = scene_of(scene)
best, bufs = lbvh(tri, pos, 2)
view = nt.BvhView(bufs[0].data_ptr(), best.nodesBytes, bufs[1].data_ptr(), best.triWoopBytes, bufs[2].data_ptr())
view.validate()
rays = scenes.box_rays(pos, nr, seed=21)
host = dict(nodes=bufs[0].cpu().numpy()[:best.nodesBytes].copy(), woop=bufs[1].cpu().numpy()[:best.triWoopBytes].copy(),
            tri_index=bufs[2].cpu().numpy()[:best.triIndexBytes].view(np.int32).copy())
steps = per_ray_steps(host, rays)
rng = np.random.default_rng(5)
orders = {"buffer order": np.arange(nr)}
orders["longest first (true steps)"] = np.argsort(-steps, kind="stable")
for sg in (1.0,):
    orders["longest first, steps known to a factor e^+-%.1f" % sg] = np.argsort(-(steps * np.exp(rng.normal(0.0, sg, nr))), kind="stable")
top = np.argsort(-steps, kind="stable")[: nr // 10]
rest = np.setdiff1d(np.arange(nr), top, assume_unique=True)
orders["longest tenth first, the rest in buffer order"] = np.concatenate([top, rest])
# the same costs DEALT over the persistent kernel's 128 pool heads (buffer-order pool: head h owns the contiguous range
# ((h & 7) * 16 + (h >> 3)) of nr / 128 rays and hands it out front to back in chunks of 64): sorted chunk c goes to head c % 128
if nr % (128 * 64) == 0:
    heads, shard = 128, nr // 128
    for label, key in (("true steps", steps.astype(np.float64)), ("steps known to a factor e^+-1.0", steps * np.exp(rng.normal(0.0, 1.0, nr)))):
        srt = np.argsort(-key, kind="stable")
        p_ = np.arange(nr)
        c = p_ // 64
        h = c % heads
        pos = ((h & 7) * (heads >> 3) + (h >> 3)) * shard + (c // heads) * 64 + (p_ % 64)
        perm = np.empty(nr, dtype=np.int64)
        perm[pos] = srt
        orders["longest first DEALT over the pool heads (%s; prediction off, 3 workgroups per CU)" % label] = perm
    orders["buffer order (prediction off, 3 workgroups per CU)"] = np.arange(nr)
ref = None
for name, perm in orders.items():
    d_r = up(rays[perm])
    out = {}
    for kernel in ("kepler_dynamic_fetch", "fermi_speculative_while_while"):
        res = torch.zeros(nr * 16, dtype=torch.uint8, device=dev)
        if "prediction off" in name:
            nt.set_tunables(NTR_TRACE_PREDICT_PERSISTENT=0, NTR_TRACE_BLOCKS_PER_CU=3)
        else:
            nt.set_tunables(NTR_TRACE_PREDICT_PERSISTENT=None, NTR_TRACE_BLOCKS_PER_CU=None)
        view.trace(kernel, nr, False, d_r.data_ptr(), res.data_ptr())
        ts = [view.trace(kernel, nr, False, d_r.data_ptr(), res.data_ptr()) for _ in range(5)]
        out[kernel.split("_")[0] + "_ms_min"] = round(min(ts) * 1e3, 3)
        out[kernel.split("_")[0] + "_ms_mean"] = round(float(np.mean(ts)) * 1e3, 3)
        got = res.cpu().numpy().view(np.int32).reshape(-1, 4)
        back = np.empty_like(got)
        back[perm] = got
        if ref is None:
            ref = back.copy()
        out[kernel.split("_")[0] + "_records_differing"] = int((back != ref).any(axis=1).sum())
    print(json.dumps(dict(scene=scene, rays=nr, order=name, steps_mean=round(float(steps.mean()), 1), steps_max=int(steps.max()),
                          steps_p99=int(np.percentile(steps, 99)), **out)), flush=True)
