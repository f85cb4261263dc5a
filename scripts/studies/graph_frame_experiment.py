#!/usr/bin/env python3
"""Experiment: one frame (primary, then 16 AO batches on three streams) captured into a HIP graph and replayed,
against the same frame issued launch by launch.  Wall clock per frame."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch
import ntrace_amd as nt
from ntrace_amd import scenes
dev = torch.device("cuda:0")
def up(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
d_nodes, d_woop, d_idx = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
view = nt.BvhView(d_nodes.data_ptr(), bvh.nodes.nbytes, d_woop.data_ptr(), bvh.woop.nbytes, d_idx.data_ptr()); view.validate()
K = "fermi_speculative_while_while"
rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]; d_rays = up(rays); d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
batches = [(n, False, d_rays, d_res)]
ns, per = 8, (1 << 20) // 8
for lo in range(0, n, per):
    cnt = min(per, n - lo)
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev); b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), lo, cnt, ns, 5.0, 0xFFF2D5E4)
    batches.append((cnt * ns, True, b_rays, b_res))
for (m, ah, r, o) in batches: view.trace(K, m, ah, r.data_ptr(), o.data_ptr())
torch.cuda.synchronize()
ref = [b[3].clone() for b in batches]
ao_streams = [torch.cuda.Stream() for _ in range(3)]
def frame(main):
    m, ah, r, o = batches[0]
    view.trace(K, m, ah, r.data_ptr(), o.data_ptr(), main.cuda_stream, False)
    e = torch.cuda.Event(); e.record(main)
    for st in ao_streams: st.wait_event(e)
    for i, (m, ah, r, o) in enumerate(batches[1:]):
        view.trace(K, m, ah, r.data_ptr(), o.data_ptr(), ao_streams[i % 3].cuda_stream, False)
    for st in ao_streams:
        e2 = torch.cuda.Event(); e2.record(st); main.wait_event(e2)
main = torch.cuda.Stream()
with torch.cuda.stream(main):
    for _ in range(3): frame(main)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(main):
    for _ in range(30): frame(main)
torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) / 30
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=main):
    frame(torch.cuda.current_stream())
for b in batches: b[3].zero_()
g.replay(); torch.cuda.synchronize()
ok = all(torch.equal(a, b[3]) for a, b in zip(ref, batches))
t0 = time.perf_counter()
for _ in range(30): g.replay()
torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 30
print("frame launch by launch %.3f ms, graph replay %.3f ms, replay results identical: %s" % (t_plain * 1e3, t_graph * 1e3, ok))
