#!/usr/bin/env python3
"""Would sorting an incoherent batch before tracing it pay?  Per scene: 2^21 box rays traced as they are (per-ray kernel, mini-pool
left to the device) against ntr_ray_morton_sort (the reference's 192-bit key; its time includes allocations of its temporaries,
so the figure is an upper bound of what a pre-pass inside the launch costs) + the trace of the sorted rays with K forced 1 / 2 / 4.

usage: sort_prepass_study.py <scene>[,<scene>...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"


def best_of(view, n, d_rays, d_res, reps=4):
    view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr())
    return min(view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr()) for _ in range(reps)) * 1e3


for scene in sys.argv[1].split(","):
    tri, pos, cam = scene_of(scene)
    if scene in ("atrium", "conference"):
        bvh = nt.sah_build(tri, pos, 1, 1)
        keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
        view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
    else:
        best, keep = lbvh(tri, pos, 2)
        view = nt.BvhView(keep[0].data_ptr(), best.nodesBytes, keep[1].data_ptr(), best.triWoopBytes, keep[2].data_ptr())
    view.validate()
    n = 1 << 21
    rays = scenes.box_rays(pos, n, seed=21)
    d_rays = up(rays)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    out = dict(scene=scene, rays=n)
    out["unsorted_ms"] = best_of(view, n, d_rays, d_res)
    ref = d_res.cpu().numpy().view(nt.RESULT_DTYPE).copy()
    d_sorted = torch.zeros_like(d_rays)
    ident = torch.arange(n, dtype=torch.int32, device=dev)
    id2slot = torch.zeros(n, dtype=torch.int32, device=dev)
    slot2id = torch.zeros(n, dtype=torch.int32, device=dev)
    nt.ray_morton_sort(n, d_rays.data_ptr(), ident.data_ptr(), d_sorted.data_ptr(), id2slot.data_ptr(), slot2id.data_ptr())
    out["sort192_ms"] = min(nt.ray_morton_sort(n, d_rays.data_ptr(), ident.data_ptr(), d_sorted.data_ptr(), id2slot.data_ptr(), slot2id.data_ptr())
                            for _ in range(3)) * 1e3
    for k in ("1", "2", "4"):
        nt.set_tunables(NTR_TRACE_MINIPOOL=k)
        out["sorted_k%s_ms" % k] = best_of(view, n, d_sorted, d_res)
    nt.set_tunables(NTR_TRACE_MINIPOOL=None)
    got = d_res.cpu().numpy().view(nt.RESULT_DTYPE)
    s2i = slot2id.cpu().numpy()
    out["records_equal"] = bool((got["id"] == ref["id"][s2i]).all() and (got["t"].view(np.uint32) == ref["t"].view(np.uint32)[s2i]).all())
    print(json.dumps(out), flush=True)
