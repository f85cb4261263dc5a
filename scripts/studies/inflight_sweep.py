#!/usr/bin/env python3
"""How many rays should be in flight?  The memory system beyond the L2 serves ~56 G requests/s from ~64 k requests in flight on; more
in flight only adds queueing delay (scripts/microbench/chase64.hip), which every ray -- the longest one included -- pays per step.
kepler_dynamic_fetch with fewer persistent waves (NTR_TRACE_BLOCKS_PER_CU 1 .. 6: 1 024 .. 6 144 waves of 64 lanes) on the divergent
batches, against the per-ray kernel.  usage: inflight_sweep.py <scene>[,<scene>]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
for scene in sys.argv[1].split(","):
    tri, pos, cam = scene_of(scene)
    best, keep = lbvh(tri, pos, 1)
    view = nt.BvhView(keep[0].data_ptr(), best.nodesBytes, keep[1].data_ptr(), best.triWoopBytes, keep[2].data_ptr())
    view.validate()
    prim = scenes.primary_rays(cam, 1920, 1080)[0]
    npr = prim.shape[0]
    d_prim = up(prim)
    d_pres = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    view.trace("fermi_speculative_while_while", npr, False, d_prim.data_ptr(), d_pres.data_ptr())
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, cnt = 8, (1 << 20) // 8
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_prim.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), min(900000, npr - cnt), cnt, ns,
                 cam["far"], 0xFFF2D5E4)
    torch.cuda.synchronize()
    batches = [("incoherent", up(scenes.box_rays(pos, 1 << 21, seed=21)), 1 << 21), ("diffuse", b_rays, cnt * ns)]
    for bname, d_rays, n in batches:
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        ref = None
        for kernel, env in [("fermi_speculative_while_while", {})] + [("kepler_dynamic_fetch", {"NTR_TRACE_BLOCKS_PER_CU": str(b), "NTR_TRACE_FETCH_THRESHOLD": str(t)})
                                                                      for b in (1, 2, 3, 4, 6) for t in (48, 60)]:
            nt.set_tunables(NTR_TRACE_BLOCKS_PER_CU=None, NTR_TRACE_FETCH_THRESHOLD=None)
            nt.set_tunables(**env)
            ts = [view.trace(kernel, n, False, d_rays.data_ptr(), d_res.data_ptr()) * 1e3 for _ in range(6)]
            got = d_res.cpu().numpy().view(nt.RESULT_DTYPE).copy()
            if ref is None:
                ref = got
            eq = bool((got["id"] == ref["id"]).all() and (got["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
            print(json.dumps(dict(scene=scene, batch=bname, kernel=kernel, env=env, ms_min=round(min(ts[2:]), 4), records_equal=eq)), flush=True)
    nt.set_tunables(NTR_TRACE_BLOCKS_PER_CU=None, NTR_TRACE_FETCH_THRESHOLD=None)
