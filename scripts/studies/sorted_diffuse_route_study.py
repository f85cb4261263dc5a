#!/usr/bin/env python3
"""Sorted secondary rays and the routing ("what comes next" (3)): one diffuse batch of the hairball frame (config 4's kind), in
generation order and Morton-sorted (ntr_ray_morton_sort, the reference's default), traced by the default selector (routed by the batch
word) and by kepler_dynamic_fetch's own body with the refill policy and the grid forced.  One JSON line per (setting, kernel)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
K0 = "fermi_speculative_while_while"


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def med(f, n=5):
    f()
    return float(np.median([f() for _ in range(n)]))


stream = torch.cuda.current_stream().cuda_stream
which = sys.argv[1] if len(sys.argv) > 1 else "hairball"
tri, pos, cam = {"hairball": scenes.hairball, "courtyard": scenes.courtyard}[which]()
lview, best, info, keep = bench.device_lbvh(nt, torch, up, dev, stream, tri, pos, 2, 8000.0)
rays, _ = scenes.primary_rays(cam, 1920, 1080)
d_pr = up(rays)
d_pres = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device=dev)
lview.trace(K0, rays.shape[0], False, d_pr.data_ptr(), d_pres.data_ptr(), stream)
d_nrm = up(scenes.tri_normals(tri, pos))
per = (1 << 20) // 8
nb = per * 8
settings = os.environ["STUDY_SETTINGS"].split("|") if os.environ.get("STUDY_SETTINGS") else ["-", "NTR_TRACE_ROUTE=0", "NTR_TRACE_ROUTE=0,NTR_TRACE_WHOLE_WAVE=0,NTR_TRACE_BLOCKS_PER_CU=3",
            "NTR_TRACE_ROUTE=0,NTR_TRACE_WHOLE_WAVE=0,NTR_TRACE_BLOCKS_PER_CU=4", "NTR_TRACE_ROUTE=0,NTR_TRACE_WHOLE_WAVE=0,NTR_TRACE_BLOCKS_PER_CU=7"]
for frac in (0.5, 0.3):
    lo = int(rays.shape[0] * frac) // per * per
    b_rays = torch.zeros(nb * 32, dtype=torch.uint8, device=dev)
    b_res = torch.zeros(nb * 16, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(nb, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_pr.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), lo, per, 8, cam["far"], 0xFFF2D5E4, stream)
    s_rays = torch.zeros_like(b_rays)
    s_i2s = torch.zeros(nb, dtype=torch.int32, device=dev)
    s_s2i = torch.zeros(nb, dtype=torch.int32, device=dev)
    ident = torch.arange(nb, dtype=torch.int32, device=dev)
    sort_ms = nt.ray_morton_sort(nb, b_rays.data_ptr(), ident.data_ptr(), s_rays.data_ptr(), s_i2s.data_ptr(), s_s2i.data_ptr(), stream) * 1e3
    live = int((b_rays.view(torch.float32).view(-1, 8)[:, 3] < b_rays.view(torch.float32).view(-1, 8)[:, 7]).sum().item())
    ref = None
    for st in settings:
        env = {} if st == "-" else dict(kv.split("=") for kv in st.split(","))
        nt.set_tunables(**env)
        for kn in (K0, "kepler_dynamic_fetch"):
            if st != "-" and kn == K0 and "WHOLE_WAVE" in st:
                continue
            tu = med(lambda: lview.trace(kn, nb, False, b_rays.data_ptr(), b_res.data_ptr(), stream, True)) * 1e3
            got_u = b_res.view(torch.int32).view(-1, 4)[:, :2].clone()
            ts = med(lambda: lview.trace(kn, nb, False, s_rays.data_ptr(), b_res.data_ptr(), stream, True)) * 1e3
            got_s = b_res.view(torch.int32).view(-1, 4)[:, :2].clone()
            if ref is None:
                ref = got_u
            ok = bool(torch.equal(got_u, ref)) and bool(torch.equal(got_s, ref[s_s2i.long()]))
            print(json.dumps(dict(scene=which, first_pixel=lo, live_rays=live, sort_ms=round(sort_ms, 3), setting=st, kernel=kn, unsorted_ms=round(tu, 4),
                                  sorted_ms=round(ts, 4), records_equal=ok)), flush=True)
        nt.set_tunables(**{k: None for k in env})
