#!/usr/bin/env python3
"""Does the secondary-ray Morton sort (ntr_ray_morton_sort) pay?  Sort time and trace time of the unsorted / sorted batch for
  * one 2^20-ray AO batch on atrium-262k (SAH BVH, any hit, radius 5),
  * 2^20 diffuse rays on hairball-2.8M (device LBVH, closest hit, extent = camera far: BASELINE configuration 4),
  * 2^20 diffuse rays on courtyard-10M (device LBVH),
  * bench.py's fully incoherent 2^21-ray batch on courtyard-10M.
One JSON line per case."""
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
K = "fermi_speculative_while_while"


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def case(name, tri, pos, cam, builder, any_hit, radius, box_rays=0):
    n = tri.shape[0]
    keep = []
    if builder == "sah":
        bvh = nt.sah_build(tri, pos, 1, 1)
        d_n, d_w, d_i = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
        keep += [d_n, d_w, d_i]
        view = nt.BvhView(d_n.data_ptr(), bvh.nodes.nbytes, d_w.data_ptr(), bvh.woop.nbytes, d_i.data_ptr())
    else:
        capn, capw, capi = nt.lbvh_capacity(n)
        d_tri, d_pos = up(tri), up(pos)
        d_n = torch.zeros(capn, dtype=torch.uint8, device=dev)
        d_w = torch.zeros(capw, dtype=torch.uint8, device=dev)
        d_i = torch.zeros(capi, dtype=torch.uint8, device=dev)
        keep += [d_tri, d_pos, d_n, d_w, d_i]
        r = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), pos.min(0), pos.max(0), 8, 0.001, d_n.data_ptr(), capn, d_w.data_ptr(), capw,
                          d_i.data_ptr(), capi)
        view = nt.BvhView(d_n.data_ptr(), r.nodesBytes, d_w.data_ptr(), r.triWoopBytes, d_i.data_ptr())
    view.validate()
    w, h = 1920, 1080
    rays, _ = scenes.primary_rays(cam, w, h)
    d_rays = up(rays)
    d_res = torch.zeros(w * h * 16, dtype=torch.uint8, device=dev)
    view.trace(K, w * h, False, d_rays.data_ptr(), d_res.data_ptr())
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, cnt = 8, (1 << 20) // 8
    first = (w * h) // 2 - cnt // 2
    m = cnt * ns
    b_rays = torch.zeros(m * 32, dtype=torch.uint8, device=dev)
    b_res = torch.zeros(m * 16, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(m, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, radius, 0xFFF2D5E4)
    if box_rays:   # fully incoherent batch: origins uniform in the bounding box, directions uniform on the sphere (bench.py's HBM point)
        m = box_rays
        b_rays = up(scenes.box_rays(pos, m, 7))
        b_res = torch.zeros(m * 16, dtype=torch.uint8, device=dev)
    so = torch.zeros_like(b_rays)
    sa = torch.zeros(m, dtype=torch.int32, device=dev)
    sb = torch.zeros(m, dtype=torch.int32, device=dev)
    ident = torch.arange(m, dtype=torch.int32, device=dev)
    ssec = min(nt.ray_morton_sort(m, b_rays.data_ptr(), ident.data_ptr(), so.data_ptr(), sa.data_ptr(), sb.data_ptr()) for _ in range(4))
    s_res = torch.zeros_like(b_res)
    tu = min(view.trace(K, m, any_hit, b_rays.data_ptr(), b_res.data_ptr()) for _ in range(6))
    ts = min(view.trace(K, m, any_hit, so.data_ptr(), s_res.data_ptr()) for _ in range(6))
    # same hit records, permuted: sorted slot k holds the ray of id sb[k]
    a = b_res.view(torch.int32).view(-1, 4)[:, :2]
    b = s_res.view(torch.int32).view(-1, 4)[:, :2]
    same = bool(torch.equal(a[sb.long()], b))
    print(json.dumps(dict(case=name, triangles=n, rays=m, any_hit=any_hit, sort_ms=ssec * 1e3, trace_unsorted_ms=tu * 1e3, trace_sorted_ms=ts * 1e3,
                          sort_pays=bool(ssec + ts < tu), same_records=same)), flush=True)


tri, pos, cam = scenes.atrium()
case("atrium-262k SAH, AO radius 5 (any hit)", tri, pos, cam, "sah", True, 5.0)
tri, pos, cam = scenes.hairball()
case("hairball-2.8M LBVH, diffuse (closest hit, extent = far)", tri, pos, cam, "lbvh", False, cam["far"])
tri, pos, cam = scenes.courtyard()
case("courtyard-10M LBVH, diffuse (closest hit, extent = far)", tri, pos, cam, "lbvh", False, cam["far"])
case("courtyard-10M LBVH, 2^21 incoherent rays (origins uniform in the box, directions on the sphere)", tri, pos, cam, "lbvh", False, cam["far"], box_rays=1 << 21)
