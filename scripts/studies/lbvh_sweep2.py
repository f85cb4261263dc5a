#!/usr/bin/env python3
"""LBVH build time of the three scene sizes over the subtree size (NTR_LBVH_SPLIT), the subtree workgroup size and the
round-1 sort / top pass, one JSON line per configuration (best of 6 builds)."""
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


which = sys.argv[1:] or ["atrium", "hairball", "courtyard"]
for name in which:
    tri, pos, cam = {"atrium": scenes.atrium, "hairball": scenes.hairball, "courtyard": scenes.courtyard}[name]()
    n = tri.shape[0]
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    dn = torch.zeros(capn, dtype=torch.uint8, device=dev)
    dw = torch.zeros(capw, dtype=torch.uint8, device=dev)
    di = torch.zeros(capi, dtype=torch.uint8, device=dev)
    mn, mx = pos.min(0), pos.max(0)
    configs = [dict(NTR_LBVH_LEGACY_TOP=1, NTR_LBVH_LEGACY_SORT=1), dict(NTR_LBVH_LEGACY_TOP=1), dict(NTR_LBVH_LEGACY_SORT=1)]
    for spill in (256, 512, 1024, 2048, 3072, 5120):
        for thr in (64, 128, 256):
            configs.append(dict(NTR_LBVH_SPLIT=spill, NTR_LBVH_SUB_THREADS=thr))
    for cfg in configs:
        nt.set_tunables(**cfg)
        best = None
        for _ in range(6):
            r = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, dn.data_ptr(), capn, dw.data_ptr(), capw,
                              di.data_ptr(), capi)
            best = r if best is None or r.seconds < best.seconds else best
        nt.set_tunables(**{k: None for k in cfg})
        print(json.dumps(dict(scene=name, triangles=n, cfg=cfg, ms=best.seconds * 1e3, phases=dict(morton=best.mortonMs, sort=best.sortMs, box=best.woopMs,
                                                                                               top=best.emitMs, rest=best.refitMs), nodes=best.numNodes)), flush=True)
