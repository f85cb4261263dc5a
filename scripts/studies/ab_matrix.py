#!/usr/bin/env python3
"""Same-box A/B of two builds of the library over the batch kinds of kernel_matrix.py: libntrace_amd.so against libntrace_amd_ab.so
(`make -C ntrace_amd/csrc ab ABFLAGS=-D...`), launches alternated, best of N each; per-ray kernel name, records compared.
usage: ab_matrix.py <scene>[,<scene>...] [reps]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import _capi, scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"
LIBS = {"a": _capi.lib_path(), "b": os.path.join(os.path.dirname(_capi.lib_path()), "libntrace_amd_ab.so")}
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for scene in sys.argv[1].split(","):
    tri, pos, cam = scene_of(scene)
    nt.use_library(LIBS["a"])
    if scene in ("atrium", "conference"):
        bvh = nt.sah_build(tri, pos, 1, 1)
        keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
        view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
    else:
        best, keep = lbvh(tri, pos, 2)
        view = nt.BvhView(keep[0].data_ptr(), best.nodesBytes, keep[1].data_ptr(), best.triWoopBytes, keep[2].data_ptr())
    for l in LIBS.values():
        nt.use_library(l)
        view.validate()
    nt.use_library(LIBS["a"])
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    npr = rays.shape[0]
    d_rays = up(rays)
    d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    view.trace(K, npr, False, d_rays.data_ptr(), d_res.data_ptr())
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, cnt = 8, (1 << 20) // 8
    diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
    radius = 5.0 if scene == "atrium" else 5.0 * diag / 4300.0
    first = min(900000, npr - cnt)
    batches = [("primary", npr, False, d_rays)]
    for nm, dist_, anyh in (("ao", radius, True), ("diffuse", cam["far"], False)):
        b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
        b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
        nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, dist_, 0xFFF2D5E4)
        batches.append((nm, cnt * ns, anyh, b_rays))
    batches.append(("incoherent", 1 << 21, False, up(scenes.box_rays(pos, 1 << 21, seed=21))))
    torch.cuda.synchronize()
    for nm, n, anyh, dr in batches:
        out = dict(scene=scene, batch=nm, rays=n)
        res = {k: torch.zeros(n * 16, dtype=torch.uint8, device=dev) for k in LIBS}
        ts = {k: [] for k in LIBS}
        for rep in range(reps + 2):
            for k, l in LIBS.items():
                nt.use_library(l)
                t = view.trace(K, n, anyh, dr.data_ptr(), res[k].data_ptr())
                if rep >= 2:
                    ts[k].append(t * 1e3)
        for k in LIBS:
            out["%s_ms" % k] = round(min(ts[k]), 4)
            out["%s_median_ms" % k] = round(float(np.median(ts[k])), 4)
        ga, gb = (res[k].cpu().numpy().view(nt.RESULT_DTYPE) for k in ("a", "b"))
        out["records_equal"] = bool((ga["id"] == gb["id"]).all() and (ga["t"].view(np.uint32) == gb["t"].view(np.uint32)).all())
        out["b_over_a"] = round(out["b_ms"] / out["a_ms"], 4)
        print(json.dumps(out), flush=True)
