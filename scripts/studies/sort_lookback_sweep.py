#!/usr/bin/env python3
"""One-sweep look-back: keys per thread of a tile (NTR_LBVH_SORT_ITEMS) x state words per round trip (NTR_SORT_LOOK) on the three LBVH
scene sizes, and the ray sort of 2^20 AO rays under NTR_SORT_LOOK.  One JSON line per configuration (best of 6 builds / 5 sorts); every
build's three output buffers are compared byte for byte with the first configuration's.
   python3 scripts/studies/sort_lookback_sweep.py [atrium hairball courtyard rays]"""
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


which = sys.argv[1:] or ["atrium", "hairball", "courtyard", "rays"]
ITEMS = [int(x) for x in os.environ.get("SWEEP_ITEMS", "0,8,16,24,32").split(",")]
LOOKS = [int(x) for x in os.environ.get("SWEEP_LOOKS", "1,2,4,8").split(",")]
for name in which:
    if name == "rays":
        continue
    tri, pos, cam = {"atrium": scenes.atrium, "hairball": scenes.hairball, "courtyard": scenes.courtyard}[name]()
    n = tri.shape[0]
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    dn = torch.zeros(capn, dtype=torch.uint8, device=dev)
    dw = torch.zeros(capw, dtype=torch.uint8, device=dev)
    di = torch.zeros(capi, dtype=torch.uint8, device=dev)
    mn, mx = pos.min(0), pos.max(0)
    ref = None
    for items in ITEMS:
        for look in LOOKS:
            cfg = dict(NTR_LBVH_SORT_ITEMS=items, NTR_SORT_LOOK=look)
            nt.set_tunables(**cfg)
            best = None
            for _ in range(6):
                r = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, dn.data_ptr(), capn, dw.data_ptr(), capw,
                                  di.data_ptr(), capi)
                best = r if best is None or r.seconds < best.seconds else best
            torch.cuda.synchronize()
            got = (dn[:best.nodesBytes].clone(), dw[:best.triWoopBytes].clone(), di[:best.triIndexBytes].clone())
            if ref is None:
                ref = got
            same = all(bool(torch.equal(a, b)) for a, b in zip(got, ref))
            nt.set_tunables(**{k: None for k in cfg})
            print(json.dumps(dict(scene=name, triangles=n, items=items, look=look, ms=round(best.seconds * 1e3, 4), sort_ms=round(best.sortMs, 4),
                                  per_pass_us=round(best.sortMs * 250, 2), morton_ms=round(best.mortonMs, 4), emit_ms=round(best.emitMs, 4),
                                  rest_ms=round(best.refitMs, 4), same_bytes=same)), flush=True)
    del d_tri, d_pos, dn, dw, di, ref

if "rays" in which:
    tri, pos, cam = scenes.atrium()
    bvh = nt.sah_build(tri, pos, 1, 1)
    keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
    view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
    stream = torch.cuda.current_stream().cuda_stream
    view.validate(stream)
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    d_pr = up(rays)
    d_pres = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device=dev)
    view.trace("fermi_speculative_while_while", rays.shape[0], False, d_pr.data_ptr(), d_pres.data_ptr(), stream)
    d_nrm = up(scenes.tri_normals(tri, pos))
    per = (1 << 20) // 8
    lo = (rays.shape[0] // 2) // per * per
    nr = per * 8
    b_rays = torch.zeros(nr * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(nr, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_pr.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), lo, per, 8, 5.0, 0xFFF2D5E4, stream)
    slot = torch.arange(nr, dtype=torch.int32, device=dev)
    o_rays = torch.zeros_like(b_rays)
    o_i2s = torch.zeros(nr, dtype=torch.int32, device=dev)
    o_s2i = torch.zeros(nr, dtype=torch.int32, device=dev)
    ref = None
    for look in LOOKS:
        nt.set_tunables(NTR_SORT_LOOK=look)
        ts = [nt.ray_morton_sort(nr, b_rays.data_ptr(), slot.data_ptr(), o_rays.data_ptr(), o_i2s.data_ptr(), o_s2i.data_ptr(), stream) for _ in range(6)]
        torch.cuda.synchronize()
        if ref is None:
            ref = o_s2i.clone()
        print(json.dumps(dict(what="ray_sort", rays=nr, look=look, ms=round(min(ts[1:]) * 1e3, 4), same_order=bool(torch.equal(ref, o_s2i)))), flush=True)
    nt.set_tunables(NTR_SORT_LOOK=None)
