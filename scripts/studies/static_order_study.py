#!/usr/bin/env python3
"""Cold AO batches: does a STATIC dispatch order (no history) recover what the learned order gives?  The learned order (heaviest cost class
first) is worth ~11 % on the frame's AO batches but needs a previous launch of the same batch.  Buffer order = PixelTable order: heavy
screen regions are contiguous, so the second round of waves of a launch may be all heavy.  Strided orders deal the blocks of s contiguous
regions of the batch round-robin (block position g -> block (g % s) * (nb / s) + g / s): every round of waves then holds an even mix while
each region still advances sequentially (neighbouring blocks of a region stay close in time).
Also: a PREDICTED order without history -- a block's cost class from the depth in the tree of the leaves its pixels' primary rays hit
(short AO rays mostly pay for descending to where they start): blocks of deeper leaves first.
Needs the experiment build (NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_exp.so: the order hook).  usage: static_order_study.py [batches=6]"""
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
nbatches = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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
# depth of the leaf of every triangle (host walk of the Compact buffers)
h_nodes = bvh.nodes.view(np.int32).reshape(-1, 16)
h_woop = bvh.woop.view(np.uint32).reshape(-1, 4)
h_tidx = bvh.tri_index.view(np.int32)
depth_of_tri = np.zeros(tri.shape[0], np.int32)
todo = [(0, 0)]
while todo:
    ofs, dpt = todo.pop()
    rec = h_nodes[ofs // 64]
    for ch in (int(rec[12]), int(rec[13])):
        if ch >= 0:
            todo.append((ch, dpt + 1))
        else:
            a4 = ~ch
            while h_woop[a4][0] != 0x80000000:
                depth_of_tri[h_tidx[a4]] = dpt + 1
                a4 += 3
h_pres = d_pres.cpu().numpy().view(nt.RESULT_DTYPE)
ns, cnt = 8, (1 << 20) // 8
n = cnt * ns
nb = n // 256
tot = {}
for b in range(nbatches):
    first = b * ((npr - cnt) // max(nbatches - 1, 1))
    b_rays = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(n, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_prim.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, 5.0, 0xFFF2D5E4)
    torch.cuda.synchronize()
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    nt.set_tunables(NTR_TRACE_AUTO_HINT="0")

    def timed(reps=6):
        return min(view.trace(K, n, True, b_rays.data_ptr(), d_res.data_ptr()) for _ in range(reps)) * 1e6
    row = {"first": first, "buffer_order": timed()}
    ref = d_res.clone()
    g = np.arange(nb)
    for s in (2, 4, 8, 16, 64, 256):
        order = ((g % s) * (nb // s) + g // s).astype(np.uint32)
        assert np.array_equal(np.sort(order), g)
        d_o = up(order)
        nt.experiment_hooks(order=d_o.data_ptr())
        row["stride_%d" % s] = timed()
        nt.experiment_hooks()
        assert torch.equal(d_res, ref), s
    ids = h_pres["id"][first:first + cnt]
    dep = np.where(ids >= 0, depth_of_tri[np.maximum(ids, 0)], 0).reshape(nb, 256 // ns).max(1)
    for classes in (64, 8):
        cls = (dep.astype(np.int64) * classes) // (int(dep.max()) + 1)
        d_o = up(np.argsort(-cls, kind="stable").astype(np.uint32))
        nt.experiment_hooks(order=d_o.data_ptr())
        row["leaf_depth_order_%d_classes" % classes] = timed()
        nt.experiment_hooks()
        assert torch.equal(d_res, ref)
    # the same through the product API: leaf depths and block costs on the device, a hint restarted from the prediction before every launch
    if b == 0:
        d_depth = torch.zeros(tri.shape[0], dtype=torch.int32, device=dev)
        nt.bvh_leaf_depths(view.d_nodes, view.nodes_bytes, view.d_woop, view.woop_bytes, view.d_tri_index, tri.shape[0], d_depth.data_ptr())
        assert np.array_equal(d_depth.cpu().numpy(), depth_of_tri)
    d_cost = torch.zeros(nb, dtype=torch.int32, device=dev)
    nt.secondary_block_costs(d_pres.data_ptr(), first, cnt, ns, d_depth.data_ptr(), tri.shape[0], d_cost.data_ptr())
    assert np.array_equal(d_cost.cpu().numpy(), dep)
    hint = nt.SchedHint()
    ts = []
    for _ in range(6):
        hint.predict(d_cost.data_ptr(), nb)
        torch.cuda.synchronize()
        ts.append(view.trace(K, n, True, b_rays.data_ptr(), d_res.data_ptr(), hint=hint))
    row["api_predicted_hint"] = min(ts) * 1e6
    assert torch.equal(d_res, ref)
    hint.close()
    nt.set_tunables(NTR_TRACE_AUTO_HINT=None)
    rr = b_rays.clone()
    for _ in range(3):
        view.trace(K, n, True, rr.data_ptr(), d_res.data_ptr())
    row["learned"] = min(view.trace(K, n, True, rr.data_ptr(), d_res.data_ptr()) for _ in range(6)) * 1e6
    assert torch.equal(d_res, ref)
    for k, v in row.items():
        if k != "first":
            tot[k] = tot.get(k, 0.0) + v
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
print(json.dumps({"sum_us": {k: round(v, 1) for k, v in tot.items()}, "relative_to_buffer_order": {k: round(v / tot["buffer_order"], 3) for k, v in tot.items()}}))
