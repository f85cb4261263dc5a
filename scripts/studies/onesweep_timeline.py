#!/usr/bin/env python3
"""Where does a one-sweep pass spend its time?  Needs a study build with the time stamps of rejected_patches/onesweep_timeline.patch
(the product sources carry no instrumentation):
   git apply scripts/studies/rejected_patches/onesweep_timeline.patch && make -C ntrace_amd/csrc clean && \
   make -C ntrace_amd/csrc -j8 EXTRA=-DNTR_OS_TIMELINE LIB=../libntrace_amd_diag.so ; git apply -R scripts/studies/rejected_patches/onesweep_timeline.patch && \
   make -C ntrace_amd/csrc clean && make -C ntrace_amd/csrc -j8
   NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_diag.so python3 scripts/studies/onesweep_timeline.py [atrium hairball courtyard]
Per scene and pass: the pass's span, tiles in flight on average, and the mean duration of a tile's phases (100 MHz stamps by thread 0):
entry -> ticket + digit scan -> keys loaded and ranked -> published + tile scan -> staged -> look-back of digit 0 -> all digits -> written."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes, _capi  # noqa: E402

dev = torch.device("cuda:0")


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


items_env = int(os.environ.get("NTR_LBVH_SORT_ITEMS", "0"))
which = sys.argv[1:] or ["atrium", "hairball", "courtyard"]
for name in which:
    tri, pos, cam = {"atrium": scenes.atrium, "hairball": scenes.hairball, "courtyard": scenes.courtyard}[name]()
    n = tri.shape[0]
    items = items_env if items_env else (32 if n >= (1 << 23) else (24 if n >= (1 << 21) else 8))
    tiles = (n + 256 * items - 1) // (256 * items)
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    dn = torch.zeros(capn, dtype=torch.uint8, device=dev)
    dw = torch.zeros(capw, dtype=torch.uint8, device=dev)
    di = torch.zeros(capi, dtype=torch.uint8, device=dev)
    mn, mx = pos.min(0), pos.max(0)
    tl = torch.zeros(4 * tiles * 8, dtype=torch.int64, device=dev)
    for rep in range(3):
        if rep == 2:
            assert _capi.lib().ntr_debug_os_timeline(C.c_void_p(tl.data_ptr())) == 0
        r = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, dn.data_ptr(), capn, dw.data_ptr(), capw, di.data_ptr(), capi)
    torch.cuda.synchronize()
    _capi.lib().ntr_debug_os_timeline(C.c_void_p(0))
    t = tl.cpu().numpy().reshape(4, tiles, 8).astype(np.float64) * 0.01   # us
    for p in range(4):
        a = t[p]
        t0 = a[:, 0].min()
        span = a[:, 7].max() - t0
        life = (a[:, 7] - a[:, 0])
        ph = [float((a[:, k + 1] - a[:, k]).mean()) for k in range(7)]
        # when did tile i enter, relative to the pass's start: quartiles over the tile index
        q = [float(a[int(f * (tiles - 1)), 0] - t0) for f in (0.0, 0.25, 0.5, 0.75, 1.0)]
        qe = [float(a[int(f * (tiles - 1)), 7] - t0) for f in (0.0, 0.25, 0.5, 0.75, 1.0)]
        print(json.dumps(dict(scene=name, n=n, items=items, tiles=tiles, pass_=p, sort_ms=round(r.sortMs, 4), span_us=round(span, 2), in_flight=round(float(life.sum() / span), 1),
                              life_us=round(float(life.mean()), 2), phases_us=[round(x, 2) for x in ph], entry_at=[round(x, 1) for x in q],
                              end_at=[round(x, 1) for x in qe])), flush=True)
