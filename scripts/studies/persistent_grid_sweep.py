#!/usr/bin/env python3
"""Grid of the persistent kernels (NTR_TRACE_BLOCKS_PER_CU) on coherent batches: atrium 1080p primary + one AO batch, both persistent
selectors.  usage: persistent_grid_sweep.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import up  # noqa: E402

dev = torch.device("cuda:0")
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
view.validate()
prim = scenes.primary_rays(cam, 1920, 1080)[0]
npr = prim.shape[0]
d_prim = up(prim)
d_pres = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
view.trace("fermi_speculative_while_while", npr, False, d_prim.data_ptr(), d_pres.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
ns, cnt = 8, (1 << 20) // 8
b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_prim.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), 900000, cnt, ns, 5.0, 0xFFF2D5E4)
d_ares = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for kernel in ("kepler_dynamic_fetch", "tesla_persistent_while_while"):
    for b in (2, 3, 4, 5, 6, 8):
        nt.set_tunables(NTR_TRACE_BLOCKS_PER_CU=str(b))
        tp = [view.trace(kernel, npr, False, d_prim.data_ptr(), d_pres.data_ptr()) * 1e3 for _ in range(6)]
        ta = [view.trace(kernel, cnt * ns, True, b_rays.data_ptr(), d_ares.data_ptr()) * 1e3 for _ in range(6)]
        print(json.dumps(dict(kernel=kernel, blocks_per_cu=b, primary_ms=round(min(tp[2:]), 4), ao_ms=round(min(ta[2:]), 4))), flush=True)
nt.set_tunables(NTR_TRACE_BLOCKS_PER_CU=None)
