#!/usr/bin/env python3
"""AO batches of one frame traced (a) from a buffer of their own each (bench.py: the library's automatic scheduling feedback learns every
batch's own order) and (b) through ONE secondary ray buffer that every batch is generated into, as the reference's Renderer does
(m_secondaryRays, src/rt/cuda/Renderer.cpp:501-564): the feedback then hands batch b + 1 the order batch b measured.  Sum of the 16
launch times, feedback on / off.  usage: ao_shared_buffer.py [scene]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import up  # noqa: E402

dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
view.validate()
prim = scenes.primary_rays(cam, 1920, 1080)[0]
npr = prim.shape[0]
d_prim = up(prim)
d_pres = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
view.trace(K, npr, False, d_prim.data_ptr(), d_pres.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
ns, per = 8, (1 << 20) // 8
own = []
for first in range(0, npr, per):
    cnt = min(per, npr - first)
    b = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b.data_ptr(), a.data_ptr(), a.data_ptr(), d_prim.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, 5.0, 0xFFF2D5E4)
    own.append((b, cnt * ns))
torch.cuda.synchronize()
shared = torch.zeros(per * ns * 32, dtype=torch.uint8, device=dev)
res = torch.zeros(per * ns * 16, dtype=torch.uint8, device=dev)
for hint in ("1", "0"):
    nt.set_tunables(NTR_TRACE_AUTO_HINT=hint)
    for mode in ("own buffers", "one shared buffer"):
        frames = []
        for frame in range(6):
            tot = 0.0
            for (b, n) in own:
                if mode == "one shared buffer":
                    shared[: n * 32].copy_(b)
                    torch.cuda.synchronize()
                    tot += view.trace(K, n, True, shared.data_ptr(), res.data_ptr())
                else:
                    tot += view.trace(K, n, True, b.data_ptr(), res.data_ptr())
            frames.append(tot * 1e3)
        print(json.dumps(dict(auto_hint=hint, mode=mode, ao_ms_per_frame=[round(f, 4) for f in frames])), flush=True)
