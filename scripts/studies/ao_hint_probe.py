import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scripts'))
import numpy as np, torch
import ntrace_amd as nt
from ntrace_amd import scenes
from workloads import up
dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
view.validate()
prim = scenes.primary_rays(cam, 1920, 1080)[0]
npr = prim.shape[0]
d_prim = up(prim); d_pres = torch.zeros(npr*16, dtype=torch.uint8, device=dev)
view.trace(K, npr, False, d_prim.data_ptr(), d_pres.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
ns, cnt = 8, (1 << 20)//8
batches = []
for first in (0, 131072*4, 131072*9):
    b_rays = torch.zeros(cnt*ns*32, dtype=torch.uint8, device=dev); b_a = torch.zeros(cnt*ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_prim.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, 5.0, 0xFFF2D5E4)
    batches.append(b_rays)
torch.cuda.synchronize()
d_res = torch.zeros(cnt*ns*16, dtype=torch.uint8, device=dev)
for mode in ("1", "0", "1"):
    nt.set_tunables(NTR_TRACE_AUTO_HINT=mode)
    for b in batches:
        bb = b.clone()
        ts = [view.trace(K, cnt*ns, True, bb.data_ptr(), d_res.data_ptr())*1e6 for _ in range(10)]
        print("auto_hint", mode, [round(t,1) for t in ts])
