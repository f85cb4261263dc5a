"""Generates tests/golden/speculative_counterexample.npz: a minimal tree + two rays on which the reference's SPECULATIVE
traversal order (postponed leaves, fermi_speculative_while_while.cu:170-186) returns a different hit record than its CPU
tracer -- the reason the shipped HIP kernels do not postpone leaves.

Geometry: two coplanar, overlapping triangles A and B in general position (a duplicated / z-fighting surface) and a
far-away triangle C; tree  root -> (leaf A, N),  N -> (leaf B, leaf C), boxes = exact bounds of the triangles, Woop rows
from this repo's host builder (CudaBVH::woopifyTri).  Seeded random search for a ray through both A and B with
    t_B (Woop)  <  t_A (Woop)  <  entry distance of B's box (slab test),
all three being roundings of the same exact distance.  CPU order: A is intersected first (ties keep child 0), tmax = t_A,
then N's children are tested: B's box starts beyond tmax -> culled -> record (A, t_A).  Speculative order with a second
lane still searching: A is postponed, N's children are tested against the OLD tmax, B is entered, then A and B are
intersected: t_B < t_A -> record (B, t_B).  The second ray (through C only) is the lane that keeps the warp searching.

Expected records come from tests/np_tracer.py (CPU order) and tests/spec_emulation.py (speculative order); the C oracle
must agree with the former.  Re-run: python tests/golden/make_spec_counterexample.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import ntrace_amd as nt  # noqa: E402
import np_tracer  # noqa: E402
import spec_emulation  # noqa: E402
from oracle import oracle  # noqa: E402

F = np.float32


def woop_rows(tri, pos):
    """Woop rows per triangle id from a host build of the scene (CudaBVH::woopifyTri)."""
    b = nt.sah_build(tri, pos)
    w = b.woop.view(np.float32).reshape(-1, 4)
    wu = w.view(np.uint32)
    out, a = {}, 0
    while a < w.shape[0]:
        if wu[a, 0] == 0x80000000:
            a += 1
            continue
        out[int(b.tri_index[a])] = w[a:a + 3].copy()
        a += 3
    return out


def assemble(tris, rows):
    """root -> (leaf 0, N); N -> (leaf 1, leaf 2); exact triangle bounds as boxes."""
    lo = [t.min(0) for t in tris]
    hi = [t.max(0) for t in tris]
    nodes = np.zeros(32, dtype=np.float32)
    ni = nodes.view(np.int32)

    def put(base, c, blo, bhi):
        nodes[base + 4 * c + 0], nodes[base + 4 * c + 1] = blo[0], bhi[0]
        nodes[base + 4 * c + 2], nodes[base + 4 * c + 3] = blo[1], bhi[1]
        nodes[base + 8 + 2 * c + 0], nodes[base + 8 + 2 * c + 1] = blo[2], bhi[2]
    put(0, 0, lo[0], hi[0])
    put(0, 1, np.minimum(lo[1], lo[2]), np.maximum(hi[1], hi[2]))
    ni[12], ni[13] = ~0, 64                       # leaf at float4 0; inner node N at byte 64
    put(16, 0, lo[1], hi[1])
    put(16, 1, lo[2], hi[2])
    ni[16 + 12], ni[16 + 13] = ~4, ~8
    woop = np.zeros((12, 4), dtype=np.float32)
    wu = woop.view(np.uint32)
    tri_index = np.zeros(12, dtype=np.int32)
    for k in range(3):
        woop[4 * k:4 * k + 3] = rows[k]
        wu[4 * k + 3, :] = 0x80000000
        tri_index[4 * k] = k
    return nodes.view(np.uint8).copy(), woop.view(np.uint8).reshape(-1).copy(), tri_index


def main():
    L = oracle.lib()
    L.orc_ray_triangle_woop.restype = C.c_float
    L.orc_ray_triangle_woop.argtypes = [C.c_void_p] * 5
    L.orc_ray_box.argtypes = [C.c_void_p] * 4
    rng = np.random.default_rng(20261003)
    far = np.array([[50, 50, 20], [52, 50, 20], [50, 52, 20]], dtype=F)
    for attempt in range(100000):
        z = F(rng.uniform(4, 8))
        P = rng.uniform(-1, 1, 2)

        def tri_around():
            while True:
                v = rng.uniform(-3, 3, (3, 2))
                M = np.array([v[1] - v[0], v[2] - v[0]]).T
                if abs(np.linalg.det(M)) < 0.5:
                    continue
                uv = np.linalg.solve(M, P - v[0])
                if uv.min() > 0.1 and uv.sum() < 0.9:
                    return np.column_stack([v, np.full(3, z)]).astype(F)
        A, B = tri_around(), tri_around()
        pos = np.concatenate([A, B, far]).astype(F)
        tri = np.arange(9, dtype=np.int32).reshape(3, 3)
        rows = woop_rows(tri, pos)
        o = np.array([P[0] + rng.uniform(-.5, .5), P[1] + rng.uniform(-.5, .5), 0], dtype=F)
        d = np.array([P[0], P[1], float(z)]) - o.astype(np.float64)
        d = (d / np.linalg.norm(d)).astype(F)
        ray = np.zeros(1, dtype=oracle.RAY_DTYPE)
        ray["ox"], ray["oy"], ray["oz"] = o
        ray["dx"], ray["dy"], ray["dz"] = d
        ray["tmin"], ray["tmax"] = 0.0, 1e30

        def wt(r):
            r = np.ascontiguousarray(r)
            return F(L.orc_ray_triangle_woop(r[0].ctypes.data, r[1].ctypes.data, r[2].ctypes.data, ray.ctypes.data, None))
        tA, tB = wt(rows[0]), wt(rows[1])
        if not (tA < 1e30 and tB < 1e30):
            continue
        lo, hi, out = B.min(0).astype(F), B.max(0).astype(F), np.zeros(2, dtype=F)
        L.orc_ray_box(lo.ctypes.data, hi.ctypes.data, ray.ctypes.data, out.ctypes.data)
        if not (out[0] > tA and tB < tA):
            continue
        nodes, woop, tri_index = assemble([A, B, far], rows)
        rays = np.zeros(2, dtype=oracle.RAY_DTYPE)
        rays[0] = ray[0]
        rays[1] = (51.0, 50.5, 0.0, 0.0, 0.0, 0.0, 1.0, 1e30)   # through C only: keeps the warp searching
        cid, ct = np_tracer.trace(nodes, woop, tri_index, rays)
        ref, _ = oracle.trace(nodes, woop, tri_index, rays)
        assert np.array_equal(cid, ref["id"]) and np.array_equal(ct.view(np.uint32), ref["t"].view(np.uint32))
        sid, st = spec_emulation.trace_warp(nodes, woop, tri_index, rays)
        alone_id, alone_t = spec_emulation.trace_warp(nodes, woop, tri_index, rays[:1])
        if not (cid[0] == 0 and sid[0] == 1 and cid[1] == 2 and sid[1] == 2):
            continue
        # a lone lane never speculates (the vote ends the loop at once): the same kernel then agrees with the CPU
        assert alone_id[0] == cid[0] and alone_t.view(np.uint32)[0] == ct.view(np.uint32)[0]
        gold = {}   # the keys of make_golden.py's fixtures, so that the golden-fixture tests (oracle, every HIP kernel) cover this case too
        for any_hit, key in ((False, "closest"), (True, "any")):
            nid, ntt, nst = np_tracer.trace(nodes, woop, tri_index, rays, any_hit=any_hit, return_stats=True)
            packed = np.zeros(rays.shape[0], dtype=nt.RESULT_DTYPE)
            packed["id"], packed["t"] = nid, ntt
            gold["res_" + key] = packed.view(np.int32).reshape(-1, 4)
            gold["stats_" + key] = np.array([nst["numInnerVisits"], nst["numTriTests"], nst["numLeafVisits"], nst["numHits"]], dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, "speculative_counterexample.npz"), nodes=nodes, woop=woop, tri_index=tri_index,
                            rays=rays.view(np.float32).reshape(-1, 8), tri=tri, pos=pos, **gold,
                            cpu_id=cid, cpu_t_bits=ct.view(np.uint32), spec_id=sid, spec_t_bits=st.view(np.uint32),
                            box_entry_B_bits=np.array([out[0]], dtype=F).view(np.uint32))
        print("attempt", attempt, "CPU order:", cid, ct, "speculative order:", sid, st, "B's box entry", out[0])
        return
    raise SystemExit("no counter-example found")


if __name__ == "__main__":
    main()
