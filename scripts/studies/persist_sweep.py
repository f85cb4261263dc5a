import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
import ntrace_amd as nt
from ntrace_amd import scenes
dev = torch.device("cuda:0")
def up(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
k3 = (up(bvh.nodes), up(bvh.woop), up(bvh.tri_index))
view = nt.BvhView(k3[0].data_ptr(), bvh.nodes.nbytes, k3[1].data_ptr(), bvh.woop.nbytes, k3[2].data_ptr()); view.validate()
rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]; d_rays = up(rays); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
view.trace("fermi_speculative_while_while", n, False, d_rays.data_ptr(), d_res.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos)); cnt, ns = (1 << 20) // 8, 8
b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev); b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev); b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), 0, cnt, ns, 5.0, 0xFFF2D5E4)
def timed(kernel, m, ah, r, o, reps=7):
    for _ in range(2): view.trace(kernel, m, ah, r.data_ptr(), o.data_ptr())
    return float(np.median([view.trace(kernel, m, ah, r.data_ptr(), o.data_ptr()) for _ in range(reps)])) * 1e6
nt.set_tunables(NTR_TRACE_PREDICT=0)
print("per-ray primary %.1f ao %.1f" % (timed("fermi_speculative_while_while", n, False, d_rays, d_res), timed("fermi_speculative_while_while", cnt * ns, True, b_rays, b_res)), flush=True)
for chunk in (64, 128):
    for bpc in (5, 6, 7, 8):
        for thr in (0, 16, 24, 32):
            nt.set_tunables(NTR_TRACE_CHUNK=chunk, NTR_TRACE_BLOCKS_PER_CU=bpc, NTR_TRACE_FETCH_THRESHOLD=thr)
            print("chunk %3d blocks/CU %d threshold %2d: primary %.1f us, AO batch %.1f us" % (chunk, bpc, thr, timed("kepler_dynamic_fetch", n, False, d_rays, d_res), timed("kepler_dynamic_fetch", cnt * ns, True, b_rays, b_res)), flush=True)
