#!/usr/bin/env python3
"""Ray splitting (trace_split.h): launch times of closest-hit batches with NTR_TRACE_SPLIT_SLICE off / on, every record compared with the
first setting's (split off).  Batches: 2^21 box rays, 1080p primary, one 2^20-ray diffuse batch; device LBVH (SAH tree for atrium).
usage: split_study.py <scene> [settings, comma separated: slice[/blocksPerCUIncoherent][+NTR_NAME=value...]] [kernels, comma separated] [batches]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import up, lbvh, scene_of  # noqa: E402

dev = torch.device("cuda:0")
scene = sys.argv[1] if len(sys.argv) > 1 else "courtyard"
settings = (sys.argv[2] if len(sys.argv) > 2 else "0,8,16,32,0").split(",")
kernels = (sys.argv[3] if len(sys.argv) > 3 else "kepler_dynamic_fetch").split(",")
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
tri, pos, cam = scene_of(scene)
if scene in ("atrium", "conference"):
    bvh = nt.sah_build(tri, pos, 1, 1)
    keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
    view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
else:
    best, bufs = lbvh(tri, pos, 2)
    view = nt.BvhView(bufs[0].data_ptr(), best.nodesBytes, bufs[1].data_ptr(), best.triWoopBytes, bufs[2].data_ptr())
view.validate()
batches = {}
nr = 1 << 21
batches["box_rays_2^21"] = (up(scenes.box_rays(pos, nr, seed=21)), nr)
rays, _ = scenes.primary_rays(cam, 1920, 1080)
npr = rays.shape[0]
d_rays = up(rays)
batches["primary_1080p"] = (d_rays, npr)
d_res0 = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
view.trace("fermi_speculative_while_while", npr, False, d_rays.data_ptr(), d_res0.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
ns, cnt = 8, (1 << 20) // 8
b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res0.data_ptr(), d_nrm.data_ptr(), min(900000, npr - cnt), cnt, ns,
             cam["far"], 0xFFF2D5E4)
batches["diffuse_2^20"] = (b_rays, cnt * ns)
# the same rays as an any-hit batch (shadow-ray like: long any-hit rays), and the box rays as any-hit
batches["anyhit_diffuse_2^20"] = (b_rays, cnt * ns)
batches["anyhit_box_rays_2^21"] = batches["box_rays_2^21"]
# ... and a short-ray AO batch (radius as in scripts/workloads.py)
diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
a_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
nt.raygen_ao(a_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res0.data_ptr(), d_nrm.data_ptr(), min(900000, npr - cnt), cnt, ns,
             5.0 if scene == "atrium" else 5.0 * diag / 4300.0, 0xFFF2D5E4)
batches["anyhit_ao_2^20"] = (a_rays, cnt * ns)
for kernel in kernels:
    for name, (d_r, n) in batches.items():
        if only and not any(name.startswith(o) for o in only):
            continue
        ref = None
        for sl in settings:
            # a setting: slice[/blocksPerCUIncoherent][+NTR_NAME=value...]
            extra = dict(kv.split("=") for kv in sl.split("+")[1:])
            f = (sl.split("+")[0].split("/") + ["3"])[:2]
            nt.set_tunables(NTR_TRACE_SPLIT_SLICE=f[0], NTR_TRACE_BLOCKS_PER_CU_INCOHERENT=f[1], **extra)
            d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
            anyh = name.startswith("anyhit")
            view.trace(kernel, n, anyh, d_r.data_ptr(), d_res.data_ptr())
            ts = [view.trace(kernel, n, anyh, d_r.data_ptr(), d_res.data_ptr()) for _ in range(5)]
            torch.cuda.synchronize()
            out = d_res.cpu().numpy().view(np.int32).reshape(-1, 4)
            nt.set_tunables(**{k: None for k in extra})
            if ref is None:
                ref = out.copy()
            diff = int((out != ref).any(axis=1).sum())
            print(json.dumps(dict(scene=scene, kernel=kernel, batch=name, rays=n, split=sl, ms_min=round(min(ts) * 1e3, 4),
                                  ms_mean=round(float(np.mean(ts)) * 1e3, 4), records_differing_from_split_off=diff,
                                  hits=int((out[:, 0] >= 0).sum()))), flush=True)
nt.set_tunables(NTR_TRACE_SPLIT_SLICE=None, NTR_TRACE_BLOCKS_PER_CU_INCOHERENT=None)
