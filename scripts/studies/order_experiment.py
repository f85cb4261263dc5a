#!/usr/bin/env python3
"""Experiment: how much does the dispatch order of the per-ray kernel's workgroups matter?
Measures per-wave lifetimes of one launch (NTR_TRACE_TIMELINE), derives block orders (heavy first,
reversed, random) and re-times the same launch under each (NTR_TRACE_ORDER)."""
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

def experiment(name, d_rays, n, any_hit):
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    nw = (n + 63) // 64; nb = (n + 255) // 256
    tl = torch.zeros(nw * 3, dtype=torch.int64, device=dev)
    def timed(reps=7):
        return np.median([view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr()) for _ in range(reps)]) * 1e6
    for _ in range(3): view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr())
    base = timed()
    ref = d_res.clone()
    nt.experiment_hooks(timeline=tl.data_ptr())
    view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr())
    nt.experiment_hooks()
    t = tl.cpu().numpy().reshape(-1, 3)
    life = np.zeros(nb * 4); life[:nw] = (t[:, 1] - t[:, 0])
    cost = life.reshape(nb, 4).max(1)
    out = {"natural": base}
    rng = np.random.default_rng(0)
    orders = {"heavy_first": np.argsort(-cost, kind="stable"), "reversed": np.arange(nb)[::-1], "random": rng.permutation(nb),
              "light_first": np.argsort(cost, kind="stable")}
    # coarse LPT: 64 cost buckets only (what a cheap device-side counting sort would give)
    b = np.minimum((cost / max(cost.max(), 1) * 63).astype(np.int64), 63)
    orders["heavy_first_64_buckets"] = np.argsort(-b, kind="stable")
    for k, o in orders.items():
        d_o = up(o.astype(np.uint32))
        nt.experiment_hooks(order=d_o.data_ptr())
        view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr())
        out[k] = timed()
        nt.experiment_hooks()
        assert torch.equal(d_res, ref), k
    # second-generation feedback: costs measured under the heavy-first order
    print(name, "rays", n, {k: round(float(v), 1) for k, v in out.items()}, "us; cost ticks p50/p99/max", np.percentile(cost, 50), np.percentile(cost, 99), cost.max())

rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]
d_rays = up(rays)
experiment("primary", d_rays, n, False)
# one AO batch (radius 5, 8 samples) generated from the first 2^17 primary hits, as bench.py does
d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
for radius in (5.0, 200.0):
    cnt, ns = (1 << 20) // 8, 8
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), 0, cnt, ns, radius, 0xFFF2D5E4)
    torch.cuda.synchronize()
    experiment("ao_r%g" % radius, b_rays, cnt * ns, True)
