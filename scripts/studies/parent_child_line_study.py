#!/usr/bin/env python3
"""Would a node in the same 128-byte line as its parent pay?  Every L2 miss of a divergent traversal fills a whole
128-byte line for a 64-byte node record (EXPERIMENTS.md, "the roof is per 128-byte LINE"); the second half of the line is a free ride
if the NEXT node the ray needs sits there.  Here the nodes of a tree are moved on the host (the study's stand-in for a numbering the
builder could produce) into aligned pairs (parent, preferred inner child): the root heads a pair with its preferred child as the tail;
a tail's inner children and a head's other inner child head pairs of their own.  Preferred = the inner child with the larger box
surface (the likelier entered by an arbitrary ray).  Layouts measured on the same tree, same visiting order, same records:
  original   the builder's numbering (LBVH: rank of the split position; SAH: createCompact's order)
  pc_pairs   aligned (parent, preferred child) pairs as above; a head without an inner child leaves the other half of its line empty
  dfs        depth-first preorder (node, then its first inner child's subtree): the cheap numbering that puts a node next to a child
             about half of the time, aligned or not
With `@sah` appended to a scene name (e.g. courtyard@sah) the tree is the host SAH build instead of the device LBVH (item 4c: what the
longest chain of dependent steps is on a better tree).

usage: parent_child_line_study.py <scene>[@sah][,<scene>...] [batches=incoherent,primary,diffuse]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"


def box_area(nf, k):
    """surface of child k's box of every node record (Compact layout: [c0.lo.x c0.hi.x c0.lo.y c0.hi.y][c1 ...][c0.lo.z c0.hi.z c1.lo.z c1.hi.z])"""
    dx = nf[:, 4 * k + 1] - nf[:, 4 * k + 0]
    dy = nf[:, 4 * k + 3] - nf[:, 4 * k + 2]
    dz = nf[:, 8 + 2 * k + 1] - nf[:, 8 + 2 * k + 0]
    return dx * dy + dy * dz + dz * dx


def relocate(nodes_u8, mode):
    nd = nodes_u8.view(np.int32).reshape(-1, 16)
    nf = nodes_u8.view(np.float32).reshape(-1, 16)
    n = nd.shape[0]
    c = np.stack([nd[:, 12].astype(np.int64), nd[:, 13].astype(np.int64)], axis=1)
    inner = c >= 0
    child = np.where(inner, c // 64, -1)
    slot = np.full(n, -1, dtype=np.int64)
    if mode == "dfs":
        # preorder by an explicit stack, vectorised per "wave" of the stack is awkward: sizes first, then offsets top-down
        order = []            # BFS levels
        level = np.array([0], dtype=np.int64)
        while level.size:
            order.append(level)
            ch = child[level].reshape(-1)
            level = ch[ch >= 0]
        size = np.ones(n, dtype=np.int64)
        for lv in reversed(order):
            for k in (0, 1):
                ck = child[lv, k]
                m = ck >= 0
                size[lv[m]] += size[ck[m]]
        slot[0] = 0
        for lv in order:
            c0, c1 = child[lv, 0], child[lv, 1]
            m0, m1 = c0 >= 0, c1 >= 0
            slot[c0[m0]] = slot[lv[m0]] + 1
            s0 = np.where(m0, size[np.where(m0, c0, 0)], 0)
            slot[c1[m1]] = slot[lv[m1]] + 1 + s0[m1]
        total = n
    else:
        area = np.stack([box_area(nf, 0), box_area(nf, 1)], axis=1)
        pref = np.where(inner[:, 0] & inner[:, 1], (area[:, 1] > area[:, 0]).astype(np.int64), np.where(inner[:, 1], 1, 0))
        slot[0] = 0
        next_pair = 1
        heads = np.array([0], dtype=np.int64)
        tails = np.zeros(0, dtype=np.int64)
        while heads.size or tails.size:
            new_heads = []
            # heads: the preferred inner child becomes the tail of the same line, the other inner child a new head
            if heads.size:
                p = pref[heads]
                pc = child[heads, p]
                has = pc >= 0
                slot[pc[has]] = slot[heads[has]] + 1
                oc = child[heads, 1 - p]
                new_heads.append(oc[oc >= 0])
                new_tails = pc[has]
            else:
                new_tails = np.zeros(0, dtype=np.int64)
            if tails.size:
                ch = child[tails].reshape(-1)
                new_heads.append(ch[ch >= 0])
            nh = np.concatenate(new_heads) if new_heads else np.zeros(0, dtype=np.int64)
            slot[nh] = 2 * (next_pair + np.arange(nh.size, dtype=np.int64))
            next_pair += nh.size
            heads, tails = nh, new_tails
        total = 2 * next_pair
    live = slot >= 0
    out = np.zeros((total, 16), dtype=np.int32)
    rec = nd[live].copy()
    for k in (12, 13):
        ck = rec[:, k].astype(np.int64)
        m = ck >= 0
        rec[m, k] = (slot[ck[m] // 64] * 64).astype(np.int32)
    out[slot[live]] = rec
    # share of inner->inner descents that stay inside the parent's 128-byte line
    li = np.nonzero(live)[0]
    same = 0
    tot = 0
    for k in (0, 1):
        ck = child[li, k]
        m = ck >= 0
        tot += int(m.sum())
        same += int(((slot[li[m]] >> 1) == (slot[ck[m]] >> 1)).sum())
    return out.reshape(-1).view(np.uint8), int(live.sum()), same / max(tot, 1)


def main():
    want = sys.argv[2].split(",") if len(sys.argv) > 2 else ["incoherent", "primary", "diffuse"]
    for spec in sys.argv[1].split(","):
        scene, _, builder = spec.partition("@")
        tri, pos, cam = scene_of(scene)
        t0 = time.time()
        if builder == "sah" or scene in ("atrium", "conference"):
            bvh = nt.sah_build(tri, pos, 1, 1) if scene in ("atrium", "conference") else nt.sah_build(tri, pos, 1, 8)
            h_nodes, h_woop, h_idx = bvh.nodes, bvh.woop, bvh.tri_index
            tree = "host SAH"
        else:
            best, keep = lbvh(tri, pos, 2)
            h_nodes = keep[0].cpu().numpy()[:best.nodesBytes].copy()
            h_woop = keep[1].cpu().numpy()[:best.triWoopBytes].copy()
            h_idx = keep[2].cpu().numpy()[:best.triIndexBytes].copy()
            del keep
            tree = "device LBVH"
        build_s = time.time() - t0
        base = np.ascontiguousarray(h_nodes).view(np.uint8).reshape(-1)
        d_w, d_i = up(h_woop), up(h_idx)
        woop_bytes = np.ascontiguousarray(h_woop).view(np.uint8).nbytes
        trees = {}
        for name in ("original", "pc_pairs", "dfs"):
            if name == "original":
                nodes, share = base, relocate_share_only(base)
            else:
                nodes, live, share = relocate(base, name)
            d_n = up(nodes)
            view = nt.BvhView(d_n.data_ptr(), nodes.nbytes, d_w.data_ptr(), woop_bytes, d_i.data_ptr())
            view.validate()
            trees[name] = (view, d_n, nodes.nbytes, share)
        rays, _ = scenes.primary_rays(cam, 1920, 1080)
        npr = rays.shape[0]
        d_rays = up(rays)
        d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
        trees["original"][0].trace(K, npr, False, d_rays.data_ptr(), d_res.data_ptr())
        batches = []
        if "primary" in want:
            batches.append(("primary", npr, False, d_rays))
        if "diffuse" in want:
            d_nrm = up(scenes.tri_normals(tri, pos))
            ns, cnt = 8, (1 << 20) // 8
            first = min(900000, npr - cnt)
            b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
            b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
            nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, cam["far"], 0xFFF2D5E4)
            batches.append(("diffuse", cnt * ns, False, b_rays))
        if "incoherent" in want:
            batches.append(("incoherent", 1 << 21, False, up(scenes.box_rays(pos, 1 << 21, seed=21))))
        torch.cuda.synchronize()
        for nm, n, anyh, dr in batches:
            out = dict(scene=scene, tree=tree, build_s=round(build_s, 2), batch=nm, rays=n, nodes=int(base.nbytes // 64))
            ref = None
            for name, (view, _, nbytes, share) in trees.items():
                res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
                out["%s_node_MB" % name] = round(nbytes / 1e6, 1)
                out["%s_same_line_share" % name] = None if share is None else round(share, 3)
                for kernel in (K, "kepler_dynamic_fetch"):
                    view.trace(kernel, n, anyh, dr.data_ptr(), res.data_ptr())
                    out["%s_%s_ms" % (name, kernel.split("_")[0])] = round(min(view.trace(kernel, n, anyh, dr.data_ptr(), res.data_ptr()) for _ in range(4)) * 1e3, 4)
                got = res.cpu().numpy().view(nt.RESULT_DTYPE).copy()
                if ref is None:
                    ref = got
                    st = view.trace_stats(K, n, anyh, dr.data_ptr(), res.data_ptr())
                    out["steps_per_ray"] = round((st.numInnerVisits + st.numTriTests) / n, 1)
                else:
                    out["%s_records_equal" % name] = bool((got["id"] == ref["id"]).all() and (got["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
            print(json.dumps(out), flush=True)


def relocate_share_only(nodes_u8):
    nd = nodes_u8.view(np.int32).reshape(-1, 16)
    idx = np.arange(nd.shape[0], dtype=np.int64)
    same = tot = 0
    for k in (12, 13):
        ck = nd[:, k].astype(np.int64)
        m = ck >= 0
        tot += int(m.sum())
        same += int(((idx[m] >> 1) == ((ck[m] // 64) >> 1)).sum())
    return same / max(tot, 1)


if __name__ == "__main__":
    main()
