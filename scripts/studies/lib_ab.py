#!/usr/bin/env python3
"""Same-box A/B of two builds of the library on the trace batches: run as
   NTR_LIB_OVERRIDE=ntrace_amd/<lib>.so python3 scripts/studies/lib_ab.py <scene>[,<scene>] [kernel]
once per library (scripts/studies/lib_ab.sh alternates them).  Prints min-of-last-4 launch times of 8 launches per batch."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
names = sys.argv[1].split(",")
kernels = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fermi_speculative_while_while", "kepler_dynamic_fetch"]
out = {}
sha = {}   # a digest of every batch's hit records: two builds must agree
for scene in names:
    tri, pos, cam = scene_of(scene)
    if scene in ("atrium", "conference"):
        bvh = nt.sah_build(tri, pos, 1, 1)
        keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
        view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
    else:
        best, keep = lbvh(tri, pos, 1)
        view = nt.BvhView(keep[0].data_ptr(), best.nodesBytes, keep[1].data_ptr(), best.triWoopBytes, keep[2].data_ptr())
    view.validate()
    prim = scenes.primary_rays(cam, 1920, 1080)[0]
    npr = prim.shape[0]
    d_prim = up(prim)
    d_pres = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    view.trace("fermi_speculative_while_while", npr, False, d_prim.data_ptr(), d_pres.data_ptr())
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, cnt = 8, (1 << 20) // 8
    diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
    b_diff = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_ao = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    first = min(900000, npr - cnt)
    nt.raygen_ao(b_diff.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_prim.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, cam["far"], 0xFFF2D5E4)
    nt.raygen_ao(b_ao.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_prim.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), first, cnt, ns,
                 5.0 if scene == "atrium" else 5.0 * diag / 4300.0, 0xFFF2D5E4)
    torch.cuda.synchronize()
    batches = [("primary", d_prim, npr, False), ("ao", b_ao, cnt * ns, True), ("diffuse", b_diff, cnt * ns, False)]
    if scene not in ("atrium", "conference"):
        batches.append(("incoherent", up(scenes.box_rays(pos, 1 << 21, seed=21)), 1 << 21, False))
    for bname, d_rays, n, any_hit in batches:
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        for kernel in kernels:
            ts = [view.trace(kernel, n, any_hit, d_rays.data_ptr(), d_res.data_ptr()) * 1e3 for _ in range(8)]
            out["%s %s %s" % (scene, bname, kernel.split("_")[0])] = round(min(ts[4:]), 4)
            sha["%s %s %s" % (scene, bname, kernel.split("_")[0])] = hashlib.sha1(d_res.cpu().numpy().tobytes()).hexdigest()[:12]
print(json.dumps(dict(lib=os.environ.get("NTR_LIB_OVERRIDE", "product"), ms=out, sha=sha)))
