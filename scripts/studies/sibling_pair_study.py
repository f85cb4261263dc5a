#!/usr/bin/env python3
"""Would 128-byte sibling pairs pay?  The device LBVH numbers a node by the rank of its split position; here its nodes are moved (on the
host, for the study) so that the two inner children of node i sit in the aligned pair of slots 2 i + 2, 2 i + 3 (root at slot 0): a ray
that visits both children then touches one 128-byte line instead of two.  Same tree, same visiting order, same records; per batch kind
the launch time of the per-ray kernel on the original and on the paired node buffer.  Also applies to a host SAH tree (atrium), whose
createCompact layout already puts siblings in adjacent slots but not in aligned pairs.

usage: sibling_pair_study.py <scene>[,<scene>...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"


def pair_nodes(nodes_u8):
    nd = nodes_u8.view(np.int32).reshape(-1, 16)
    n = nd.shape[0]
    l0, l1 = nd[:, 12].astype(np.int64), nd[:, 13].astype(np.int64)
    slot = np.full(n, -1, dtype=np.int64)
    slot[0] = 0
    # reachable nodes only; parent -> children slots 2 * slot-rank... use the node's own index for its children's pair
    idx = np.arange(n, dtype=np.int64)
    in0, in1 = l0 >= 0, l1 >= 0
    c0, c1 = l0[in0] // 64, l1[in1] // 64
    slot[c0] = 2 * idx[in0] + 2
    slot[c1] = 2 * idx[in1] + 3
    out = np.zeros((2 * n + 2, 16), dtype=np.int32)
    live = slot >= 0
    rec = nd[live].copy()
    li = idx[live]
    rec[:, 12] = np.where(rec[:, 12] >= 0, (2 * li + 2) * 64, rec[:, 12])
    rec[:, 13] = np.where(rec[:, 13] >= 0, (2 * li + 3) * 64, rec[:, 13])
    out[slot[live]] = rec
    return out.reshape(-1).view(np.uint8), int(live.sum())


for scene in sys.argv[1].split(","):
    tri, pos, cam = scene_of(scene)
    if scene in ("atrium", "conference"):
        bvh = nt.sah_build(tri, pos, 1, 1)
        h_nodes, h_woop, h_idx = bvh.nodes, bvh.woop, bvh.tri_index
    else:
        best, keep = lbvh(tri, pos, 2)
        h_nodes = keep[0].cpu().numpy()[:best.nodesBytes].copy()
        h_woop = keep[1].cpu().numpy()[:best.triWoopBytes].copy()
        h_idx = keep[2].cpu().numpy()[:best.triIndexBytes].copy()
        del keep
    paired, live = pair_nodes(np.ascontiguousarray(h_nodes).view(np.uint8).reshape(-1))
    d_w, d_i = up(h_woop), up(h_idx)
    trees = {}
    for name, nodes in (("original", np.ascontiguousarray(h_nodes).view(np.uint8).reshape(-1)), ("paired", paired)):
        d_n = up(nodes)
        view = nt.BvhView(d_n.data_ptr(), nodes.nbytes, d_w.data_ptr(), np.ascontiguousarray(h_woop).view(np.uint8).nbytes, d_i.data_ptr())
        view.validate()
        trees[name] = (view, d_n)
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    npr = rays.shape[0]
    d_rays = up(rays)
    d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    trees["original"][0].trace(K, npr, False, d_rays.data_ptr(), d_res.data_ptr())
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, cnt = 8, (1 << 20) // 8
    diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
    radius = 5.0 if scene == "atrium" else 5.0 * diag / 4300.0
    first = min(900000, npr - cnt)
    batches = [("primary", npr, False, d_rays)]
    for nm, dist_, anyh in (("ao", radius, True), ("diffuse", cam["far"], False)):
        b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
        b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
        nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, dist_, 0xFFF2D5E4)
        batches.append((nm, cnt * ns, anyh, b_rays))
    batches.append(("incoherent", 1 << 21, False, up(scenes.box_rays(pos, 1 << 21, seed=21))))
    torch.cuda.synchronize()
    for nm, n, anyh, dr in batches:
        out = dict(scene=scene, batch=nm, rays=n, nodes=int(h_nodes.nbytes // 64), reachable=live)
        ref = None
        for name, (view, _) in trees.items():
            res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
            for kernel in (K, "kepler_dynamic_fetch"):
                view.trace(kernel, n, anyh, dr.data_ptr(), res.data_ptr())
                out["%s_%s_ms" % (name, kernel.split("_")[0])] = round(min(view.trace(kernel, n, anyh, dr.data_ptr(), res.data_ptr()) for _ in range(4)) * 1e3, 4)
            got = res.cpu().numpy().view(nt.RESULT_DTYPE).copy()
            if ref is None:
                ref = got
            else:
                out["records_equal"] = bool((got["id"] == ref["id"]).all() and (got["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
        print(json.dumps(out), flush=True)
