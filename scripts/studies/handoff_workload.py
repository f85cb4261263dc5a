#!/usr/bin/env python3
"""One divergent batch, traced `reps` times under whatever NTR_* tunables the environment holds: the workload of the rocprofv3 passes of
the tail hand-off study (scripts/studies/handoff_pmc.sh).  usage: handoff_workload.py <scene> <incoherent|diffuse> [reps]"""
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
KERNEL = "fermi_speculative_while_while"
scene, batch = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
tri, pos, cam = scene_of(scene)
best, keep = lbvh(tri, pos, 1)
view = nt.BvhView(keep[0].data_ptr(), best.nodesBytes, keep[1].data_ptr(), best.triWoopBytes, keep[2].data_ptr())
view.validate()
if batch == "incoherent":
    rays = scenes.box_rays(pos, 1 << 21, seed=21)
    d_rays = up(rays)
else:
    prim = scenes.primary_rays(cam, 1920, 1080)[0]
    npr = prim.shape[0]
    d_prim = up(prim)
    d_pres = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    nt.set_tunables(NTR_TRACE_AUTO_HINT="0")
    view.trace(KERNEL, npr, False, d_prim.data_ptr(), d_pres.data_ptr())
    nt.set_tunables(NTR_TRACE_AUTO_HINT=None)
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, cnt = 8, (1 << 20) // 8
    first = min(900000, npr - cnt)
    d_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(d_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_prim.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, cam["far"], 0xFFF2D5E4)
    torch.cuda.synchronize()
n = d_rays.numel() // 32
d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
ts = [view.trace(KERNEL, n, False, d_rays.data_ptr(), d_res.data_ptr()) * 1e3 for _ in range(reps)]
st = view.trace_stats(KERNEL, n, False, d_rays.data_ptr(), d_res.data_ptr())
print(json.dumps(dict(scene=scene, batch=batch, rays=n, ms=[round(t, 4) for t in ts], handoff=nt.trace_handoff_counts(0), stats=st.as_dict(),
                      algorithmic_bytes=st.algorithmic_bytes(), env={k: v for k, v in os.environ.items() if k.startswith("NTR_TRACE")})))
