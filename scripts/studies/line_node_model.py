#!/usr/bin/env python3
"""Model of a one-line node record (round-5 idea; EXPERIMENTS.md, "the roof is per 128-byte line"): a fetch brings a node's record AND the
record of one of its children chosen at build time, so a ray that goes on to that child needs no new fetch.  How many of a divergent ray's
dependent fetches does that remove?  CPU only: the numpy tracer (tests/np_tracer.py, lock step) on a sample of the 2^21 box rays of an LBVH
scene, with a hook that sees which node follows which.  Rules for the stored child: child 0; the child with the larger surface area.
Reported: fetches per ray (inner + triangle steps today), the share of inner steps that become free, and the longest ray's chain.

usage: line_node_model.py <scene> [rays=4096]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import np_tracer  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402  (analysis script, not product)


def main():
    scene = sys.argv[1]
    nrays = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    tri, pos, cam = {"hairball": scenes.hairball, "courtyard": scenes.courtyard}[scene]()
    b = oracle.lbvh_build(tri, pos, 8, 0.001)
    nodes = np.ascontiguousarray(b["nodes"])
    nf = np.frombuffer(nodes.tobytes(), dtype=np.float32)
    ni = nf.view(np.int32)
    nn = nf.size // 16
    rec = nf.reshape(nn, 16)
    reci = ni.reshape(nn, 16)
    area = lambda lx, hx, ly, hy, lz, hz: 2.0 * ((hx - lx) * (hy - ly) + (hy - ly) * (hz - lz) + (hz - lz) * (hx - lx))
    a0 = area(rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 8], rec[:, 9])
    a1 = area(rec[:, 4], rec[:, 5], rec[:, 6], rec[:, 7], rec[:, 10], rec[:, 11])
    c0, c1 = reci[:, 12].astype(np.int64), reci[:, 13].astype(np.int64)
    rules = {
        "child0": np.where(c0 >= 0, c0, np.where(c1 >= 0, c1, -1)),
        "larger_area": np.where((c0 >= 0) & ((a0 >= a1) | (c1 < 0)), c0, np.where(c1 >= 0, c1, np.where(c0 >= 0, c0, -1))),
    }
    rays = scenes.box_rays(pos, 1 << 21, seed=21)[:nrays]
    inner_n = np.zeros(nrays, np.int64)
    free = {k: np.zeros(nrays, np.int64) for k in rules}

    def hook(idx, visited, after):
        inner_n[idx] += 1
        vi = visited // 64
        for k, pref in rules.items():
            free[k][idx] += (after >= 0) & (after == pref[vi])
    rid, rt, st = np_tracer.trace(b["nodes"], b["woop"], b["tri_index"], rays, any_hit=False, return_stats=True, inner_hook=hook)
    # triangle steps per ray: from the oracle's per-ray counters
    import ctypes as C
    L = oracle.lib()
    vp = C.c_void_p
    L.orc_trace_compact_counts.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp]
    res = np.zeros(nrays, dtype=oracle.RESULT_DTYPE)
    inn = np.zeros(nrays, np.int32)
    trs = np.zeros(nrays, np.int32)
    r2 = np.ascontiguousarray(rays)
    assert L.orc_trace_compact_counts(b["nodes"].ctypes.data, b["woop"].ctypes.data, b["tri_index"].ctypes.data, r2.ctypes.data, res.ctypes.data, nrays, 0,
                                      inn.ctypes.data, trs.ctypes.data) == 0
    assert np.array_equal(inn.astype(np.int64), inner_n), "hook count != oracle inner count"
    steps = inner_n + trs
    out = dict(scene=scene, rays=nrays, inner_per_ray=float(inner_n.mean()), tri_steps_per_ray=float(trs.mean()), fetches_per_ray_today=float(steps.mean()),
               longest_chain_today=int(steps.max()))
    # triangles: a one-line record holds two of a leaf's triangles (96 bytes + the look-ahead word), so a leaf of k triangles costs ceil(k / 2)
    # fetches instead of k: between tris / 2 (every leaf even) and (tris + leaf visits) / 2 (every leaf odd)
    _, stats = oracle.trace(b["nodes"], b["woop"], b["tri_index"], rays, any_hit=False, threads=8)
    out["triangles_per_leaf_visit"] = stats.numTriTests / max(stats.numLeafVisits, 1)
    tri_lo, tri_hi = trs.sum() / 2.0, (trs.sum() + stats.numLeafVisits) / 2.0
    for k in rules:
        f = steps - free[k]
        out.setdefault("with_paired_triangles", {})[k] = dict(fetch_ratio_low=float((inner_n.sum() - free[k].sum() + tri_lo) / steps.sum()),
                                                              fetch_ratio_high=float((inner_n.sum() - free[k].sum() + tri_hi) / steps.sum()))
        out[k] = dict(free_share_of_inner_steps=float(free[k].sum() / inner_n.sum()), fetches_per_ray=float(f.mean()), fetch_ratio=float(f.sum() / steps.sum()),
                      longest_chain=int(f.max()), longest_chain_ratio=float(f.max() / steps.max()))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
