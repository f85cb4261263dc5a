#!/usr/bin/env python3
"""Prints the records that differ between split-off and split-on launches (atrium SAH tree, box rays)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ntrace_amd as nt
from ntrace_amd import scenes
from workloads import up, scene_of
dev = torch.device("cuda:0")
kernel = sys.argv[1] if len(sys.argv) > 1 else "fermi_speculative_while_while"
sl = sys.argv[2] if len(sys.argv) > 2 else "32"
tri, pos, cam = scene_of("atrium")
bvh = nt.sah_build(tri, pos, 1, 1)
keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
view.validate()
nr = 1 << 21
rays = scenes.box_rays(pos, nr, seed=21)
d_r = up(rays)
outs = []
for s in ("0", sl):
    nt.set_tunables(NTR_TRACE_SPLIT_SLICE=s)
    d_res = torch.zeros(nr * 16, dtype=torch.uint8, device=dev)
    view.trace(kernel, nr, False, d_r.data_ptr(), d_res.data_ptr())
    torch.cuda.synchronize()
    outs.append(d_res.cpu().numpy().view(np.int32).reshape(-1, 4).copy())
bad = np.nonzero((outs[0] != outs[1]).any(axis=1))[0]
rv = np.asarray(rays).view(np.float32).reshape(-1, 8)
for i in bad[:40]:
    a, b = outs[0][i], outs[1][i]
    print(json.dumps(dict(ray=int(i), lane=int(i % 64), off=dict(id=int(a[0]), t=float(a[1:2].view(np.float32)[0]), tbits=int(a[1]), u=int(a[2]), v=int(a[3])),
                          on=dict(id=int(b[0]), t=float(b[1:2].view(np.float32)[0]), tbits=int(b[1]), u=int(b[2]), v=int(b[3])),
                          o=[float(x) for x in rv[i][:4]], d=[float(x) for x in rv[i][4:]])))
print("differing", len(bad))
