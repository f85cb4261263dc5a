#!/usr/bin/env python3
"""How well does the depth of the leaf a pixel's primary ray hit predict what the pixel's secondary rays cost?  CPU only (oracle step counts):
Spearman rank correlation, per 256-ray block, between the deepest leaf among the block's 32 pixels and the block's longest wave, for one AO and
one diffuse batch at two places of the 1080p frame.  AO (short any-hit rays): 0.6-0.9 -- the predictor behind ntr_secondary_block_costs;
diffuse (long closest-hit rays): 0.0-0.5 -- not used there.   usage: leaf_depth_predictor_corr.py atrium,hairball"""
import sys, os, json, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntrace_amd as nt
from ntrace_amd import scenes
from oracle import oracle
import np_raygen
from scipy.stats import spearmanr
L = oracle.lib()
vp = C.c_void_p
L.orc_trace_compact_counts.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp]; L.orc_trace_compact_counts.restype = C.c_int

def leaf_depths(nodes_u8, woop_u8, tidx, ntris):
    nodes = np.ascontiguousarray(nodes_u8).view(np.uint8).reshape(-1).view(np.int32).reshape(-1, 16)
    woop = np.ascontiguousarray(woop_u8).view(np.uint8).reshape(-1).view(np.uint32).reshape(-1, 4)
    tidx = np.ascontiguousarray(tidx).view(np.uint8).reshape(-1).view(np.int32)
    depth = np.zeros(ntris, np.int32)
    todo = [(0, 0)]
    while todo:
        ofs, d = todo.pop()
        rec = nodes[ofs // 64]
        for ch in (int(rec[12]), int(rec[13])):
            if ch >= 0: todo.append((ch, d + 1))
            else:
                a = ~ch
                while woop[a][0] != 0x80000000:
                    depth[tidx[a]] = d + 1; a += 3
    return depth

for scene in sys.argv[1].split(","):
    tri, pos, cam = {"atrium": scenes.atrium, "hairball": scenes.hairball, "conference": scenes.conference_room}[scene]()
    if scene in ("atrium", "conference"):
        b = nt.sah_build(tri, pos, 1, 1); N, W, T = b.nodes, b.woop, b.tri_index
    else:
        r = oracle.lbvh_build(tri, pos, 8, 0.001); N, W, T = r["nodes"], r["woop"], r["tri_index"]
    N = np.ascontiguousarray(N); W = np.ascontiguousarray(W); T = np.ascontiguousarray(T)
    dep_tri = leaf_depths(N, W, T, tri.shape[0])
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    res, _ = oracle.trace(N, W, T, rays, any_hit=False, threads=8)
    nrm = scenes.tri_normals(tri, pos)
    cnt, ns = (1 << 20) // 8, 8
    diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
    for kind, maxd, anyhit in (("ao", 5.0 if scene == "atrium" else 5.0 * diag / 4300.0, 1), ("diffuse", cam["far"], 0)):
        for first in (388505, 1165515):
            o, d, tmax = np_raygen.ao_rays(rays, res, nrm, ns, maxd, 0xFFF2D5E4, first, cnt)
            ao = np.zeros(cnt * ns, dtype=nt.RAY_DTYPE)
            ao["ox"], ao["oy"], ao["oz"] = o[:, 0], o[:, 1], o[:, 2]
            ao["dx"], ao["dy"], ao["dz"] = d[:, 0], d[:, 1], d[:, 2]
            ao["tmin"] = 0.0; ao["tmax"] = tmax
            n = ao.shape[0]
            r2 = np.zeros(n, dtype=oracle.RESULT_DTYPE); inner = np.zeros(n, np.int32); tris = np.zeros(n, np.int32)
            ao = np.ascontiguousarray(ao)
            assert L.orc_trace_compact_counts(N.ctypes.data, W.ctypes.data, T.ctypes.data, ao.ctypes.data, r2.ctypes.data, n, anyhit, inner.ctypes.data, tris.ctypes.data) == 0
            steps = (inner + tris).astype(np.int64)
            ids = res["id"][first:first + cnt]
            dep = np.where(ids >= 0, dep_tri[np.maximum(ids, 0)], 0)
            blk_cost = steps.reshape(-1, 4, 64).max(2).max(1)
            blk_dep = dep.reshape(-1, 32).max(1)
            print(json.dumps(dict(scene=scene, kind=kind, first=first, mean_steps=round(float(steps.mean()), 1), hit_fraction=round(float((ids >= 0).mean()), 3),
                                  spearman_block_cost_vs_leaf_depth=round(float(spearmanr(blk_cost, blk_dep)[0]), 3))), flush=True)
