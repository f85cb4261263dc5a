#!/usr/bin/env python3
"""Launch times of consecutive fermi launches of one incoherent batch (automatic scheduling feedback: first sighting, hint allocation,
refresh launches, steady state), and the same after the batch's primary rays were traced through the same BVH first.
usage: fermi_launch_sequence.py <scene>"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ntrace_amd as nt
from ntrace_amd import scenes
from workloads import up, lbvh, scene_of
dev = torch.device("cuda:0")
scene = sys.argv[1] if len(sys.argv) > 1 else "hairball"
K = "fermi_speculative_while_while"
tri, pos, cam = scene_of(scene)
best, bufs = lbvh(tri, pos, 2)
view = nt.BvhView(bufs[0].data_ptr(), best.nodesBytes, bufs[1].data_ptr(), best.triWoopBytes, bufs[2].data_ptr())
view.validate()
nr = 1 << 21
d_rr = up(scenes.box_rays(pos, nr, seed=21))
for prime in (False, True):
    nt.set_tunables()
    if prime:
        rays, _ = scenes.primary_rays(cam, 1920, 1080)
        d_rays = up(rays)
        d_res = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device=dev)
        view.trace(K, rays.shape[0], False, d_rays.data_ptr(), d_res.data_ptr())
    res = torch.zeros(nr * 16, dtype=torch.uint8, device=dev)
    ts = [round(view.trace(K, nr, False, d_rr.data_ptr(), res.data_ptr()) * 1e3, 3) for _ in range(20)]
    print(json.dumps(dict(scene=scene, primary_traced_first=prime, launch_ms=ts)), flush=True)
