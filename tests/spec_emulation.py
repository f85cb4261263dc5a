"""Warp-level emulation of the CONTROL FLOW of the reference's speculative kernel
(src/rt/kernels/fermi_speculative_while_while.cu:110-254) with the CPU tracer's ARITHMETIC (test helper).

The reference's speculative kernels postpone the first leaf a ray reaches and keep traversing inner nodes -- with the
hit distance the ray had BEFORE that leaf was intersected -- for as long as any other lane of the warp is still looking
for a leaf (:170-186).  The CPU tracer (src/rt/cuda/CudaBVH.cpp:721-775) intersects the leaf first and tests the next
boxes against the shrunken ray.tmax.  In exact arithmetic both find the same closest hit; in binary32 they need not: a
box whose rounded entry distance exceeds the new tmax is culled by the CPU order, while the speculative order has
already entered it, and a triangle inside whose rounded t undercuts tmax is then accepted.  This module reproduces the
order only -- box and triangle tests are the CPU tracer's own expressions (Util.cpp:34-46, 99-127) -- so that a differing
record is due to the order and nothing else.  tests/golden/speculative_counterexample.npz holds such a case."""
import numpy as np

F = np.float32
SENTINEL = 0x76543210
FLT_MAX = F(3.4028234663852886e38)


def _smin(a, b):
    return a if a < b else b


def _smax(a, b):
    return a if a > b else b


def _box(lo, hi, o, d):
    with np.errstate(all="ignore"):
        t0 = [F(F(lo[k] - o[k]) / d[k]) for k in range(3)]
        t1 = [F(F(hi[k] - o[k]) / d[k]) for k in range(3)]
    mn = _smax(_smax(_smin(t0[0], t1[0]), _smin(t0[1], t1[1])), _smin(t0[2], t1[2]))
    mx = _smin(_smin(_smax(t0[0], t1[0]), _smax(t0[1], t1[1])), _smax(t0[2], t1[2]))
    return mn, mx


def _dot4(a, b):
    r = F(0.0)
    for k in range(4):
        r = F(r + F(a[k] * b[k]))
    return r


def _woop(rows, o, d, tmin, tmax):
    """Intersect::RayTriangleWoop (Util.cpp:99-127): t, or FLT_MAX for a miss."""
    z, u4, v4 = rows
    with np.errstate(all="ignore"):
        Oz = F(F(F(z[3] - F(o[0] * z[0])) - F(o[1] * z[1])) - F(o[2] * z[2]))
        t = F(Oz * F(F(1.0) / _dot4(z, (d[0], d[1], d[2], F(0.0)))))
        if t > tmin and t < tmax:
            u = F(_dot4(u4, (o[0], o[1], o[2], F(1.0))) + F(t * _dot4(u4, (d[0], d[1], d[2], F(0.0)))))
            if u >= 0:
                v = F(_dot4(v4, (o[0], o[1], o[2], F(1.0))) + F(t * _dot4(v4, (d[0], d[1], d[2], F(0.0)))))
                if v >= 0 and F(u + v) <= 1:
                    return t
    return FLT_MAX


def trace_warp(nodes, woop, tri_index, rays):
    """All `rays` (<= 32) as ONE warp of the speculative kernel, closest hit.  Returns (ids, t)."""
    nf = np.frombuffer(np.ascontiguousarray(nodes).tobytes(), dtype=F)
    ni = nf.view(np.int32)
    wf = np.frombuffer(np.ascontiguousarray(woop).tobytes(), dtype=F).reshape(-1, 4)
    wu = wf.view(np.uint32)
    n = rays.shape[0]
    o = [(F(r["ox"]), F(r["oy"]), F(r["oz"])) for r in rays]
    d = [(F(r["dx"]), F(r["dy"]), F(r["dz"])) for r in rays]
    tmin = [F(r["tmin"]) for r in rays]
    hit_t = [F(r["tmax"]) for r in rays]
    hit_i = [-1] * n
    stack = [[SENTINEL] for _ in range(n)]           # traversalStack[0] = EntrypointSentinel (:100)
    node = [0] * n                                   # nodeAddr = 0: the root (:103)
    leaf = [0] * n                                   # leafAddr = 0: no postponed leaf (:102)

    def inner_step(i):
        b = node[i] // 4
        g = lambda k: nf[b + k]
        mn0, mx0 = _box((g(0), g(2), g(8)), (g(1), g(3), g(9)), o[i], d[i])
        mn1, mx1 = _box((g(4), g(6), g(10)), (g(5), g(7), g(11)), o[i], d[i])
        i0 = (mn0 <= mx0) and (mx0 >= tmin[i]) and (mn0 <= hit_t[i])     # CudaBVH.cpp:742-743
        i1 = (mn1 <= mx1) and (mx1 >= tmin[i]) and (mn1 <= hit_t[i])
        c0, c1 = int(ni[b + 12]), int(ni[b + 13])
        if not i0 and not i1:
            node[i] = stack[i].pop()
        else:
            node[i] = c0 if i0 else c1
            if i0 and i1:
                far = c1
                if mn1 < mn0:                                             # ties keep child 0 (CudaBVH.cpp:761)
                    node[i], far = c1, c0
                stack[i].append(far)

    def leaf_step(i, leaf_addr):
        a = ~leaf_addr
        while wu[a, 0] != 0x80000000:
            tt = _woop((wf[a], wf[a + 1], wf[a + 2]), o[i], d[i], tmin[i], hit_t[i])
            if tt > tmin[i] and tt < hit_t[i]:                            # updateHit (CudaBVH.cpp:1200)
                hit_t[i] = tt
                hit_i[i] = int(tri_index[a])
            a += 3

    while any(nd != SENTINEL for nd in node):                             # :110 (per lane; lanes that are done idle)
        searching = [True] * n                                            # :114
        while True:
            active = [i for i in range(n) if node[i] >= 0 and node[i] != SENTINEL]   # :115
            if not active:
                break
            for i in active:
                inner_step(i)
                if node[i] < 0 and leaf[i] >= 0:                          # first leaf: postpone, keep traversing (:170-176)
                    searching[i] = False
                    leaf[i] = node[i]
                    node[i] = stack[i].pop()
            if not any(searching[i] for i in active):                     # all voting lanes have found a leaf (:180-181)
                break
        for i in range(n):
            while leaf[i] < 0:                                            # postponed leaves (:186-249)
                leaf_step(i, leaf[i])
                leaf[i] = node[i]                                         # a second leaf found while speculating
                if node[i] < 0:
                    node[i] = stack[i].pop()
    return np.array(hit_i, dtype=np.int32), np.array(hit_t, dtype=F)
