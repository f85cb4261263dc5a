#!/usr/bin/env python3
"""LBVH build phases under values of one environment tunable:  python3 scripts/studies/lbvh_knob.py VAR v1 v2 ... [-- scene ...]
("-" = unset).  Best of 8 builds per value; the three output buffers are compared byte for byte with the first value's."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402

dev = torch.device("cuda:0")


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


argv = sys.argv[1:]
which = ["atrium", "hairball", "courtyard"]
if "--" in argv:
    which = argv[argv.index("--") + 1:]
    argv = argv[:argv.index("--")]
var, vals = argv[0], argv[1:]
for name in which:
    tri, pos, cam = {"atrium": scenes.atrium, "hairball": scenes.hairball, "courtyard": scenes.courtyard}[name]()
    n = tri.shape[0]
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    dn = torch.zeros(capn, dtype=torch.uint8, device=dev)
    dw = torch.zeros(capw, dtype=torch.uint8, device=dev)
    di = torch.zeros(capi, dtype=torch.uint8, device=dev)
    mn, mx = pos.min(0), pos.max(0)
    ref = None
    for v in vals:
        nt.set_tunables(**{var: None if v == "-" else v})
        best = None
        for _ in range(8):
            r = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, dn.data_ptr(), capn, dw.data_ptr(), capw, di.data_ptr(), capi)
            best = r if best is None or r.seconds < best.seconds else best
        torch.cuda.synchronize()
        got = (dn[:best.nodesBytes].clone(), dw[:best.triWoopBytes].clone(), di[:best.triIndexBytes].clone())
        ref = ref or got
        same = all(bool(torch.equal(a, b)) for a, b in zip(got, ref))
        print(json.dumps(dict(scene=name, triangles=n, var=var, value=v, ms=round(best.seconds * 1e3, 4), morton_ms=round(best.mortonMs, 4), sort_ms=round(best.sortMs, 4),
                              marks_ms=round(best.emitMs, 4), emit_ms=round(best.refitMs, 4), same_bytes=same)), flush=True)
    nt.set_tunables(**{var: None})
