"""Hand-derived known-answer vectors for the trace path.

The reference ships no golden vectors for this path and cannot be built in this image, so every other fixture in
this repo is produced by a restatement of it.  These are not: each case is a two-leaf BVHLayout_Compact tree written
out by hand in exact binary fractions, and every expected (id, t) below was worked out on paper from the reference's
source expressions -- one case per rule that decides bits:

  accept rule      tmin<=tmax && tmax>=ray.tmin && tmin<=ray.tmax        src/rt/cuda/CudaBVH.cpp:742-743
  visiting order   nearer child first, ties keep child 0                 src/rt/cuda/CudaBVH.cpp:755-764
  strict hit test  t>ray.tmin && t<ray.tmax, first accepted wins ties    src/rt/cuda/CudaBVH.cpp:1200-1215
  FLT_MAX retest   a missed Woop test returns FW_F32_MAX, which updateHit accepts when ray.tmax = +inf
                                                                         src/rt/Util.cpp:122-126, CudaBVH.cpp:1121-1124, 1200
  select min/max   (a<b)?a:b / (a>b)?a:b and their NaN behaviour         src/framework/base/Defs.hpp:212-213, Util.cpp:39-45
  1.f / dot        t = Oz * (1.f / Dz), not Oz / Dz                      src/rt/Util.cpp:107-109
  miss record      (-1, ray.tmax)                                        src/rt/cuda/CudaBVH.cpp:273-274

Woop rows of a right triangle with corner v2 = (x2, y2, z2) and legs a along +x (v0 = v2 + (a,0,0)) and b along +y
(v1 = v2 + (0,b,0)): the matrix with columns (v0-v2, v1-v2, n = (0,0,ab), v2) has the inverse rows
(1/a, 0, 0, -x2/a), (0, 1/b, 0, -y2/b), (0, 0, 1/(ab), -z2/(ab)), so (CudaBVH.cpp:680-686)
  woopZ = (0, 0, 1/(ab), z2/(ab))    woopU = (1/a, 0, 0, -x2/a)    woopV = (0, 1/b, 0, -y2/b)
and for a ray (o, d): Oz = z2/(ab) - oz/(ab), Dz = dz/(ab), t = Oz * (1/Dz), u = (ox + t dx - x2)/a, v likewise.
With a = b = 2 everything below is exact in binary32.

Used by tests/test_kat_cpu.py (oracle, numpy restatement, the mirror's host tracer) and tests/test_kat_gpu.py (HIP)."""
import struct

import numpy as np

RAY_DTYPE = np.dtype([("ox", "<f4"), ("oy", "<f4"), ("oz", "<f4"), ("tmin", "<f4"), ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"), ("tmax", "<f4")])
FLT_MAX_BITS = 0x7F7FFFFF
INF = float("inf")


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def from_bits(u):
    return struct.unpack("<f", struct.pack("<I", u))[0]


def tri_rows(x2, y2, z2, a=2.0, b=2.0):
    ab = a * b
    return [(0.0, 0.0, 1.0 / ab, z2 / ab), (1.0 / a, 0.0, 0.0, -x2 / a), (0.0, 1.0 / b, 0.0, -y2 / b)]


def two_leaf_bvh(box0, box1, rows0, rows1, id0, id1):
    """box = (lo.x, hi.x, lo.y, hi.y, lo.z, hi.z).  Root at byte 0, child 0 = leaf at float4 0, child 1 = leaf at float4 4."""
    nodes = np.zeros(16, dtype=np.float32)
    nodes[0:4] = box0[0:4]
    nodes[4:8] = box1[0:4]
    nodes[8:12] = (box0[4], box0[5], box1[4], box1[5])
    ni = nodes.view(np.int32)
    ni[12], ni[13], ni[14], ni[15] = ~0, ~4, 0, 0
    woop = np.zeros((8, 4), dtype=np.float32)
    woop[0:3] = rows0
    woop[4:7] = rows1
    wu = woop.view(np.uint32)
    wu[3, :] = 0x80000000
    wu[7, :] = 0x80000000
    tri_index = np.array([id0, 0, 0, 0, id1, 0, 0, 0], dtype=np.int32)
    return nodes.view(np.uint8).copy(), woop.view(np.uint8).reshape(-1).copy(), tri_index


def ray(o, d, tmin, tmax):
    return (o[0], o[1], o[2], tmin, d[0], d[1], d[2], tmax)


def cases():
    """[(name, nodes, woop, tri_index, rays, expected)], expected = list of dicts with
    closest=(id, t bits), any=(id, t bits), and optionally inner / tris = counters of the closest-hit trace."""
    out = []
    up, down = (0.0, 0.0, 1.0), (0.0, 0.0, -1.0)
    pz = np.float32
    after4 = float(np.nextafter(pz(4.0), pz(INF)))      # 0x40800001
    after45 = float(np.nextafter(pz(4.5), pz(INF)))
    before35 = float(np.nextafter(pz(3.5), pz(-INF)))   # 3.4999998

    # ---- A: two triangles behind each other on the z axis -------------------------------------------------------------
    boxA0, boxA1 = (0, 2, 0, 2, 3.5, 4.5), (0, 2, 0, 2, 7.5, 8.5)
    bvhA = two_leaf_bvh(boxA0, boxA1, tri_rows(0, 0, 4), tri_rows(0, 0, 8), 10, 20)
    o = (0.5, 0.5, 0.0)
    raysA, expA = [], []

    def add(r, closest, any_, inner=None, tris=None):
        raysA.append(r)
        expA.append(dict(closest=closest, any=any_, inner=inner, tris=tris))
    # Oz = 4/4 - 0 = 1, Dz = 1/4, t = 1 * 4 = 4; u = v = 0.25.  Child 0 is nearer (3.5 < 7.5); its hit shrinks tmax to 4; child 1
    # was pushed with the old tmax and is a leaf, so its triangle is still tested (t = 8, rejected): 1 inner node, 2 tests.
    add(ray(o, up, 0.0, 100.0), (10, bits(4.0)), (10, bits(4.0)), 1, 2)
    # from the far side: Oz = 8/4 - 12/4 = -1, Dz = -1/4, 1/Dz = -4, t = 4 on triangle 20 (child 1 is nearer: span [3.5, 4.5])
    add(ray((0.5, 0.5, 12.0), down, 0.0, 100.0), (20, bits(4.0)), (20, bits(4.0)), 1, 2)
    # strict t < tmax: a hit at exactly tmax is no hit; child 1's box starts beyond tmax and is culled: miss = (-1, ray.tmax)
    add(ray(o, up, 0.0, 4.0), (-1, bits(4.0)), (-1, bits(4.0)), 1, 1)
    add(ray(o, up, 0.0, after4), (10, bits(4.0)), (10, bits(4.0)), 1, 1)
    # strict t > tmin: the triangle at exactly tmin is skipped, the next one is hit
    add(ray(o, up, 4.0, 100.0), (20, bits(8.0)), (20, bits(8.0)), 1, 2)
    # accept rule, tmax >= ray.tmin: box 0 ends exactly at ray.tmin = 4.5 -> entered (its triangle is tested and rejected) ...
    add(ray(o, up, 4.5, 100.0), (20, bits(8.0)), (20, bits(8.0)), 1, 2)
    # ... one ulp later it is culled: a single triangle test
    add(ray(o, up, after45, 100.0), (20, bits(8.0)), (20, bits(8.0)), 1, 1)
    # accept rule, tmin <= ray.tmax: box 0 starts exactly at ray.tmax = 3.5 -> entered; one ulp shorter -> culled
    add(ray(o, up, 0.0, 3.5), (-1, bits(3.5)), (-1, bits(3.5)), 1, 1)
    add(ray(o, up, 0.0, before35), (-1, bits(before35)), (-1, bits(before35)), 1, 0)
    # FLT_MAX retest: through both boxes but outside both triangles (u = v = 0.75).  RayTriangleWoop returns FW_F32_MAX, and
    # updateHit accepts it because FLT_MAX < +inf: the record is (first triangle tested, FLT_MAX).  The second test returns
    # FLT_MAX again, which is not < tmax = FLT_MAX.
    add(ray((1.5, 1.5, 0.0), up, 0.0, INF), (10, FLT_MAX_BITS), (10, FLT_MAX_BITS), 1, 2)
    add(ray((1.5, 1.5, 0.0), up, 0.0, 100.0), (-1, bits(100.0)), (-1, bits(100.0)), 1, 2)
    # degenerate ray (tmax < tmin, Ray::degenerate): no box passes tmax >= ray.tmin && tmin <= ray.tmax
    add(ray(o, up, 5.0, 4.0), (-1, bits(4.0)), (-1, bits(4.0)), None, None)
    out.append(("A_depth_order_and_strict_tests", bvhA[0], bvhA[1], bvhA[2], np.array(raysA, dtype=np.float32).view(RAY_DTYPE).reshape(-1), expA))

    # ---- B: ties between the children -----------------------------------------------------------------------------------
    same = (0, 2, 0, 2, 3.5, 4.5)
    r1 = np.array([ray(o, up, 0.0, 100.0)], dtype=np.float32).view(RAY_DTYPE).reshape(-1)
    # equal entry distances: no swap, child 0 first; child 1's identical triangle has t = 4, not < tmax = 4
    b = two_leaf_bvh(same, same, tri_rows(0, 0, 4), tri_rows(0, 0, 4), 7, 9)
    out.append(("B1_tie_keeps_child0", b[0], b[1], b[2], r1, [dict(closest=(7, bits(4.0)), any=(7, bits(4.0)), inner=1, tris=2)]))
    # child 1 nearer by one ulp: it is visited first and keeps the hit
    b = two_leaf_bvh(same, (0, 2, 0, 2, before35, 4.5), tri_rows(0, 0, 4), tri_rows(0, 0, 4), 7, 9)
    out.append(("B2_child1_one_ulp_nearer", b[0], b[1], b[2], r1, [dict(closest=(9, bits(4.0)), any=(9, bits(4.0)), inner=1, tris=2)]))
    b = two_leaf_bvh((0, 2, 0, 2, before35, 4.5), same, tri_rows(0, 0, 4), tri_rows(0, 0, 4), 7, 9)
    out.append(("B3_child0_one_ulp_nearer", b[0], b[1], b[2], r1, [dict(closest=(7, bits(4.0)), any=(7, bits(4.0)), inner=1, tris=2)]))

    # ---- E: select-form min / max with NaN (ray in a slab plane, direction component 0) ----------------------------------
    # Origin on the LOW x plane of box 0 (x in [0.5, 2]), d = (+0, +0, 1):
    #   t0 = ((0.5-0.5)/0, (0-0.5)/0, 3.5) = (NaN, -inf, 3.5)    t1 = (1.5/0, 1.5/0, 4.5) = (+inf, +inf, 4.5)
    #   min(t0,t1) = ((NaN<inf)?NaN:inf, -inf, 3.5) = (inf, -inf, 3.5) -> .max() = inf
    #   max(t0,t1) = ((NaN>inf)?NaN:inf, inf, 4.5) -> .min() = 4.5         inf <= 4.5 is false: box 0 is NOT entered,
    # although the ray grazes its triangle (u = 0, v = 0.25 would be accepted); fminf/fmaxf would enter it.  Box 1 (x in [0, 2])
    # gives (-inf, -inf, 7.5) / (+inf, +inf, 8.5) and is entered: the record is triangle 20 at t = 8.
    e = two_leaf_bvh((0.5, 2, 0, 2, 3.5, 4.5), boxA1, tri_rows(0.5, 0, 4), tri_rows(0, 0, 8), 10, 20)
    re = np.array([ray(o, (0.0, 0.0, 1.0), 0.0, 100.0), ray(o, (-0.0, 0.0, 1.0), 0.0, 100.0)], dtype=np.float32).view(RAY_DTYPE).reshape(-1)
    # with d.x = -0: t0.x = NaN, t1.x = -inf -> min = -inf, max = -inf -> tmax = -inf < tmin = 3.5: not entered either
    out.append(("E1_origin_on_low_slab_plane", e[0], e[1], e[2], re,
                [dict(closest=(20, bits(8.0)), any=(20, bits(8.0)), inner=1, tris=1), dict(closest=(20, bits(8.0)), any=(20, bits(8.0)), inner=1, tris=1)]))
    # Origin on the HIGH x plane of box 0 (x in [-1, 0.5]), d.x = +0:
    #   t0.x = -1.5/0 = -inf, t1.x = 0/0 = NaN: min = (-inf<NaN)?-inf:NaN = NaN, max = NaN; a NaN in the FIRST slot of the folds
    #   is dropped ((NaN>y)?NaN:y = y), so tmin = max(-inf, 3.5) = 3.5, tmax = min(inf, 4.5) = 4.5: box 0 IS entered, and its
    #   triangle (corner (-0.5, 0, 4)) is hit at u = (0.5+0.5)/2 = 0.5, v = 0.25, t = 4.
    e = two_leaf_bvh((-1, 0.5, 0, 2, 3.5, 4.5), boxA1, tri_rows(-0.5, 0, 4), tri_rows(0, 0, 8), 10, 20)
    out.append(("E2_origin_on_high_slab_plane", e[0], e[1], e[2], re[:1], [dict(closest=(10, bits(4.0)), any=(10, bits(4.0)), inner=1, tris=2)]))

    # ---- F: t = Oz * (1.f / Dz) ---------------------------------------------------------------------------------------------
    # woopZ = (0, 0, 3, 5), ray o = (1, 1, 0), d = (0, 0, 1): Oz = 5, Dz = 3.  1.f/3 = 0x3EAAAAAB = 0.3333333432674408;
    # 5 * that = 1.6666667163372040, which lies above the midpoint 1.6666666865348816 of the neighbouring floats 0x3FD55555 and
    # 0x3FD55556, so t = 0x3FD55556.  The true quotient 5/3 rounds to 0x3FD55555: an implementation that divides is one ulp off.
    f = two_leaf_bvh((0, 4, 0, 4, 1, 2), (100, 101, 100, 101, 100, 101), [(0, 0, 3, 5), (0.25, 0, 0, 0), (0, 0.25, 0, 0)], tri_rows(100, 100, 100), 3, 4)
    rf = np.array([ray((1.0, 1.0, 0.0), up, 0.0, 100.0)], dtype=np.float32).view(RAY_DTYPE).reshape(-1)
    out.append(("F_reciprocal_then_multiply", f[0], f[1], f[2], rf, [dict(closest=(3, 0x3FD55556), any=(3, 0x3FD55556), inner=1, tris=1)]))
    return out
