#!/usr/bin/env python3
"""ntr_ray_morton_sort: device time (the call's own event bracket) against host wall clock per call, 2^20 and 2^17 random rays.
   [NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_b.so] python3 scripts/studies/ray_sort_wall.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for n in (1 << 20, 1 << 17):
    rays = scenes.random_rays(n, seed=3, tmax=4.0)
    d_in = torch.from_numpy(rays.view(np.uint8).reshape(-1).copy()).to(dev)
    ident = torch.arange(n, dtype=torch.int32, device=dev)
    d_out = torch.zeros_like(d_in)
    a = torch.zeros(n, dtype=torch.int32, device=dev)
    b = torch.zeros(n, dtype=torch.int32, device=dev)
    for _ in range(3):
        nt.ray_morton_sort(n, d_in.data_ptr(), ident.data_ptr(), d_out.data_ptr(), a.data_ptr(), b.data_ptr(), stream)
    torch.cuda.synchronize()
    devs, walls = [], []
    for _ in range(20):
        t0 = time.perf_counter()
        devs.append(nt.ray_morton_sort(n, d_in.data_ptr(), ident.data_ptr(), d_out.data_ptr(), a.data_ptr(), b.data_ptr(), stream))
        walls.append(time.perf_counter() - t0)
    print(json.dumps(dict(lib=os.environ.get("NTR_LIB_OVERRIDE", "product"), rays=n, device_ms=round(float(np.median(devs)) * 1e3, 4),
                          wall_ms=round(float(np.median(walls)) * 1e3, 4))), flush=True)
