#!/usr/bin/env python3
"""Round 6 probe: what would XCD-local work assignment be worth on the headline frame?  Workgroups are dealt round-robin to the eight XCDs
(workgroup i -> XCD i % 8), so in buffer order every XCD's 4 MB L2 sees the whole 34 MB tree.  Permuting the 64-ray chunks of a batch so
that chunk i holds the (i / 8)-th chunk of the (i % 8)-th contiguous eighth gives XCD k the k-th eighth of the batch (one screen region)
without touching the kernel.  Buffer-order dispatch (automatic hints and prediction off); times by the library's events, best of 7."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ntrace_amd as nt
from ntrace_amd import scenes, dist as ntd
import bench

dev = torch.device("cuda:0")


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


stream = torch.cuda.current_stream().cuda_stream
args = bench.parse(["--no-extras"])
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
blob = torch.zeros(bvh.nodes.nbytes + bvh.woop.nbytes + bvh.tri_index.nbytes + 512, dtype=torch.uint8, device=dev)
o1 = (bvh.nodes.nbytes + 255) // 256 * 256
o2 = o1 + (bvh.woop.nbytes + 255) // 256 * 256
blob[:bvh.nodes.nbytes].copy_(up(bvh.nodes)); blob[o1:o1 + bvh.woop.nbytes].copy_(up(bvh.woop)); blob[o2:o2 + bvh.tri_index.nbytes].copy_(up(bvh.tri_index))
view = nt.BvhView(blob.data_ptr(), bvh.nodes.nbytes, blob.data_ptr() + o1, bvh.woop.nbytes, blob.data_ptr() + o2)
view.validate(stream)
d_nrm = up(scenes.tri_normals(tri, pos))
frame = bench.Frame(nt, torch, view, lambda d: ntd.FramePlan(1920 * 1080, 0, 1, 8, 1 << 20), cam, 1920, 1080, d_nrm, args, dev, stream, scenes)
nt.set_tunables(NTR_TRACE_AUTO_HINT=0, NTR_TRACE_PREDICT=0)
K = "fermi_speculative_while_while"
rng = np.random.default_rng(1)
for bi in (0, 2, 5, 9, 13):
    b = frame.batches[bi]
    n = b["n"]
    rays = (frame.d_rays if bi == 0 else b["rays_t"]).view(torch.uint8)[: n * 32].reshape(-1, 32)
    nch = n // 64
    base = torch.arange(nch, device=dev)
    per = (nch + 7) // 8
    src = (base % 8) * per + base // 8          # chunk i <- chunk (i % 8) * per + i / 8 (XCD k gets the k-th eighth)
    ok = src < nch
    src = torch.where(ok, src, base)             # (a ragged tail keeps its place: only whole eighths matter for the probe)
    # make it a permutation: fall back to identity unless it is one
    perm_ok = bool(torch.equal(torch.sort(src).values, base))
    row = {"batch": bi, "rays": n, "any_hit": b["any_hit"], "permutation": perm_ok}
    variants = {"buffer_order": None}
    if perm_ok:
        variants["xcd_contiguous_eighths"] = src
    variants["random_chunks"] = torch.from_numpy(rng.permutation(nch)).to(dev)
    res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    for name, p in variants.items():
        if p is None:
            r2 = rays
        else:
            idx = (p[:, None] * 64 + torch.arange(64, device=dev)[None, :]).reshape(-1)
            r2 = torch.cat([rays[: nch * 64][idx], rays[nch * 64:]])
        r2 = r2.contiguous()
        for _ in range(2):
            view.trace(K, n, b["any_hit"], r2.data_ptr(), res.data_ptr(), stream, True)
        row[name + "_ms"] = min(view.trace(K, n, b["any_hit"], r2.data_ptr(), res.data_ptr(), stream, True) for _ in range(7)) * 1e3
    print(json.dumps(row), flush=True)
