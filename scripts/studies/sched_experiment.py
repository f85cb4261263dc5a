#!/usr/bin/env python3
"""Scheduling-hint experiment: kernel time per generation and per number of cost classes, and the
wall-clock cost of the bookkeeping, on the bench workload's primary batch and one AO batch."""
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

def run(name, d_rays, n, any_hit):
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    for _ in range(3): view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr())
    base = np.median([view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr()) for _ in range(9)]) * 1e6
    ref = d_res.clone()
    print("%s: %d rays, no hint %.1f us" % (name, n, base))
    for classes in (4, 8, 16, 32, 64):
        nt.set_tunables(NTR_SCHED_CLASSES=classes)
        h = nt.SchedHint()
        ts = []
        for g in range(20):
            d_res.zero_()
            ts.append(view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr(), hint=h) * 1e6)
            assert torch.equal(d_res, ref), (classes, g)
        print("  classes %2d: gen0..5 %s | steady median %.1f us (min %.1f)" % (classes, " ".join("%.0f" % t for t in ts[:6]), np.median(ts[6:]), min(ts[6:])))
        h.close()
    nt.set_tunables(NTR_SCHED_CLASSES=None)
    # wall-clock per launch, asynchronous launches back to back
    stream = torch.cuda.current_stream().cuda_stream
    for label, hint in (("no hint", None), ("hint", nt.SchedHint())):
        for _ in range(4): view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr(), stream, False, hint=hint)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): view.trace(K, n, any_hit, d_rays.data_ptr(), d_res.data_ptr(), stream, False, hint=hint)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40 * 1e6
        print("  wall per launch, 40 async launches, %s: %.1f us" % (label, dt))

rays, _ = scenes.primary_rays(cam, 1920, 1080)
n = rays.shape[0]; d_rays = up(rays)
run("primary", d_rays, n, False)
d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
for radius in (5.0, 200.0):
    cnt, ns = (1 << 20) // 8, 8
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), 0, cnt, ns, radius, 0xFFF2D5E4)
    torch.cuda.synchronize()
    run("ao_r%g" % radius, b_rays, cnt * ns, True)

# ---- moving camera: the hint comes from the PREVIOUS frame's (different) rays -------------------------
w, h = 1920, 1080
d_tab = torch.zeros(n, dtype=torch.int32, device=dev)
nt.pixel_table(w, h, d_tab.data_ptr(), 0)
d_i2s = torch.zeros(n, dtype=torch.int32, device=dev); d_s2i = torch.zeros(n, dtype=torch.int32, device=dev)
d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
for step_units in (5.0, 25.0, 100.0):   # the hall is 3600 units long; 25 units per frame = 1.5 m/frame walk
    hint = nt.SchedHint()
    t_plain, t_hint = [], []
    for f in range(24):
        c = dict(cam)
        eye = np.array(c["eye"], dtype=np.float64); tgt = np.array(c["target"], dtype=np.float64)
        d = (tgt - eye) / np.linalg.norm(tgt - eye)
        eye2 = eye + d * step_units * f + np.array([0.0, 0.0, 3.0]) * np.sin(f * 0.7) * step_units / 5
        c["eye"] = tuple(eye2); c["target"] = tuple(eye2 + (tgt - eye) + np.array([0.0, 0.0, 40.0]) * np.sin(f * 0.3))
        nt.raygen_primary(d_rays.data_ptr(), d_i2s.data_ptr(), d_s2i.data_ptr(), d_tab.data_ptr(), c["eye"], scenes.nscreen_to_world(c, w, h), w, h, c["far"], 0)
        torch.cuda.synchronize()
        t_plain.append(min(view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr()) for _ in range(3)) * 1e6)
        t_hint.append(view.trace(K, n, False, d_rays.data_ptr(), d_res.data_ptr(), hint=hint) * 1e6)   # ONE hinted trace per frame
    print("moving camera, %g units/frame: plain mean %.1f us, hinted (hint from previous frame) mean %.1f us over frames 3.. ; per frame hinted/plain %s"
          % (step_units, np.mean(t_plain[3:]), np.mean(t_hint[3:]), " ".join("%.2f" % (a / b) for a, b in zip(t_hint, t_plain))))
    hint.close()
