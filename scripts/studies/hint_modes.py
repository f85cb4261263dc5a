#!/usr/bin/env python3
"""Primary batch of the bench scene traced repeatedly: no feedback, the library's automatic feedback, a caller-owned NtrSchedHint.
Per mode the per-launch times by torch events (asynchronous launches, as bench.py issues them)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch
import ntrace_amd as nt
from ntrace_amd import scenes
from workloads import up
dev = torch.device("cuda:0")
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
d_n, d_w, d_i = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
view = nt.BvhView(d_n.data_ptr(), bvh.nodes.nbytes, d_w.data_ptr(), bvh.woop.nbytes, d_i.data_ptr())
view.validate()
rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]
d_r = up(rays)
d_o = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
E = torch.cuda.Event
st = torch.cuda.current_stream().cuda_stream
for mode in ("none", "auto", "explicit", "auto", "explicit", "none"):
    nt.set_tunables(NTR_TRACE_AUTO_HINT=0 if mode != "auto" else 1)
    hint = nt.SchedHint() if mode == "explicit" else None
    ts = []
    for i in range(40):
        e0, e1 = E(enable_timing=True), E(enable_timing=True)
        e0.record()
        view.trace("fermi_speculative_while_while", n, False, d_r.data_ptr(), d_o.data_ptr(), st, False, hint=hint)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(json.dumps(dict(mode=mode, first5=[round(x, 4) for x in ts[:5]], last24_mean=float(np.mean(ts[16:])), last24_min=float(np.min(ts[16:])),
                          all=[round(x, 3) for x in ts])), flush=True)
    if hint is not None:
        hint.close()
