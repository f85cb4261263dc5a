#!/usr/bin/env python3
"""ntr_predict_batch_coherence on the four batch kinds of scripts/kernel_matrix.py: [origin-incoherent blocks, direction-incoherent blocks, K].
usage: coherence_probe.py <scene>[,<scene>...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
for scene in sys.argv[1].split(","):
    tri, pos, cam = scene_of(scene)
    if scene in ("atrium", "conference"):
        bvh = nt.sah_build(tri, pos, 1, 1)
        keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
        nb = bvh.nodes.nbytes
        view = nt.BvhView(keep[0].data_ptr(), nb, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
    else:
        best, keep = lbvh(tri, pos, 2)
        nb = best.nodesBytes
        view = nt.BvhView(keep[0].data_ptr(), nb, keep[1].data_ptr(), best.triWoopBytes, keep[2].data_ptr())
    view.validate()
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    npr = rays.shape[0]
    d_rays = up(rays)
    d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    view.trace("fermi_speculative_while_while", npr, False, d_rays.data_ptr(), d_res.data_ptr())
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, cnt = 8, (1 << 20) // 8
    first = min(900000, npr - cnt)
    out = torch.zeros(3, dtype=torch.int32, device=dev)
    res = dict(scene=scene, nodes_mb=nb / 2**20)
    for name, maxd in (("ao", 5.0), ("diffuse", cam["far"])):
        b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
        b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
        nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, maxd, 0xFFF2D5E4)
        nt.predict_batch_coherence(cnt * ns, b_rays.data_ptr(), keep[0].data_ptr(), nb, out.data_ptr())
        torch.cuda.synchronize()
        res[name] = out.cpu().tolist() + [cnt * ns // 256]
        h = b_rays.cpu().numpy().view(nt.RAY_DTYPE)
        res[name + "_sample"] = [[float(h[k][i]) for k in ("ox", "oy", "oz", "dx", "dy", "dz", "tmin", "tmax")] for i in (100, 227)]
    nt.predict_batch_coherence(npr, d_rays.data_ptr(), keep[0].data_ptr(), nb, out.data_ptr())
    torch.cuda.synchronize()
    res["primary"] = out.cpu().tolist() + [npr // 256]
    inc = up(scenes.box_rays(pos, 1 << 21, seed=21))
    nt.predict_batch_coherence(1 << 21, inc.data_ptr(), keep[0].data_ptr(), nb, out.data_ptr())
    torch.cuda.synchronize()
    res["incoherent"] = out.cpu().tolist() + [(1 << 21) // 256]
    print(json.dumps(res), flush=True)
