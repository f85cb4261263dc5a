#!/usr/bin/env python3
"""From which batch size does the wave-private mini-pool pay?  Box rays (incoherent) of 2^17 .. 2^22 rays per scene, K forced 1 / 2 / 4
(and left to the device): best-of launch times.  A batch that does not oversubscribe the machine (rays / 64 waves against 7 168 wave
slots) is critical-path bound, and fewer, longer-lived waves only lengthen that path.
usage: small_batch_minipool.py <scene>[,<scene>...]"""
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
    prim = scenes.primary_rays(cam, 1920, 1080)[0]
    for n in (1 << 19, 1 << 20, 3 << 19, 1 << 21, 3 << 20, 1 << 22):
        rays = scenes.box_rays(pos, n, seed=21)
        out = dict(scene=scene, batch="incoherent", rays=n)
        ref = None
        for mode in ("1", "2", "3", "4", "5", "6", "8", "10", None):
            nt.set_tunables(NTR_TRACE_MINIPOOL=mode)
            d_rays = up(rays)   # (a new buffer: a new hint)
            d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
            ts = [view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr()) * 1e3 for _ in range(6)]
            out["ms_k%s" % (mode or "auto")] = round(min(ts[1:]), 4)
            got = d_res.cpu().numpy().view(nt.RESULT_DTYPE).copy()
            if ref is None:
                ref = got
            else:
                out["records_equal"] = out.get("records_equal", True) and bool((got["id"] == ref["id"]).all() and (got["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
        print(json.dumps(out), flush=True)
