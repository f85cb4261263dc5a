#!/usr/bin/env python3
"""ntr_lbvh_build: device time (the build's own event bracket) against host wall clock per call."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402

dev = torch.device("cuda:0")


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


for name in sys.argv[1:] or ["atrium", "hairball"]:
    tri, pos, cam = {"atrium": scenes.atrium, "hairball": scenes.hairball, "courtyard": scenes.courtyard}[name]()
    n = tri.shape[0]
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    dn = torch.zeros(capn, dtype=torch.uint8, device=dev)
    dw = torch.zeros(capw, dtype=torch.uint8, device=dev)
    di = torch.zeros(capi, dtype=torch.uint8, device=dev)
    mn, mx = pos.min(0), pos.max(0)
    args = (n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, dn.data_ptr(), capn, dw.data_ptr(), capw, di.data_ptr(), capi)
    for _ in range(3):
        nt.lbvh_build(*args)
    devs, walls = [], []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = nt.lbvh_build(*args)
        walls.append(time.perf_counter() - t0)
        devs.append(r.seconds)
    print(json.dumps(dict(scene=name, triangles=n, device_ms=round(float(np.median(devs)) * 1e3, 4), wall_ms=round(float(np.median(walls)) * 1e3, 4),
                          wall_min_ms=round(float(np.min(walls)) * 1e3, 4))), flush=True)
