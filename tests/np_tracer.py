"""Independent numpy binary32 restatement of the reference CPU tracer (test helper).

Written separately from oracle/ntr_oracle_trace.c (different language, lock-step
vectorised structure) to cross-check it: the reference ships no golden vectors for this
path and cannot be compiled in this image, so two independent restatements that agree bit
for bit are the strongest pin available.  Follows src/rt/cuda/CudaBVH.cpp:698-784,
1084-1126, 1183-1225, 1250-1265 and src/rt/Util.cpp:34-46, 99-127.
"""
import numpy as np

F = np.float32
FLT_MAX = np.float32(3.4028234663852886e38)


def _smin(a, b):  # FW::min: (a < b) ? a : b   (Defs.hpp:212)
    return np.where(a < b, a, b)


def _smax(a, b):  # FW::max: (a > b) ? a : b   (Defs.hpp:213)
    return np.where(a > b, a, b)


def _dot4(a, bx, by, bz, bw):
    r = np.zeros_like(bx, dtype=F)      # r = 0
    r = r + a[:, 0] * bx                # r += a[i] * b[i], left to right (Math.hpp:185)
    r = r + a[:, 1] * by
    r = r + a[:, 2] * bz
    r = r + a[:, 3] * bw
    return r


def trace(nodes, woop, tri_index, rays, any_hit=False, max_stack=100, return_stats=False, inner_hook=None):
    """Returns (id, t); with return_stats also the counters (inner nodes visited, triangle tests, leaf terminators
    read, hits) as the reference's RayStats defines them (CudaBVH.cpp:746-749, 1107-1111).
    inner_hook(ray indices, node visited (byte offset), node the ray holds afterwards): called once per lock-step inner step
    (analysis scripts: which node follows which)."""
    n_inner = n_tri = n_leaf = 0
    nodes_f = np.frombuffer(np.ascontiguousarray(nodes).tobytes(), dtype=F)
    nodes_i = nodes_f.view(np.int32)
    woop_f = np.frombuffer(np.ascontiguousarray(woop).tobytes(), dtype=F).reshape(-1, 4)
    woop_u = woop_f.view(np.uint32)
    tri_index = np.asarray(tri_index, dtype=np.int32)
    n = rays.shape[0]
    ox, oy, oz = rays["ox"].astype(F), rays["oy"].astype(F), rays["oz"].astype(F)
    dx, dy, dz = rays["dx"].astype(F), rays["dy"].astype(F), rays["dz"].astype(F)
    tmin = rays["tmin"].astype(F)
    tmax = rays["tmax"].astype(F).copy()

    res_id = np.full(n, -1, dtype=np.int32)
    res_t = tmax.copy()                                   # result.t = ray.tmax (CudaBVH.cpp:274)
    node = np.zeros(n, dtype=np.int64)                    # trace<Compact>(0, ...)
    stack = np.zeros((n, max_stack), dtype=np.int64)
    sp = np.ones(n, dtype=np.int64)                       # stackIndex = 1
    tri_cur = np.full(n, -1, dtype=np.int64)              # >= 0 while inside a leaf
    done = np.zeros(n, dtype=bool)
    one, zero = F(1.0), F(0.0)

    def pop(mask):
        idx = np.nonzero(mask)[0]
        sp[idx] -= 1
        node[idx] = stack[idx, sp[idx]]
        finished = idx[sp[idx] <= 0]                      # while (stackIndex > 0)
        done[finished] = True

    with np.errstate(all="ignore"):
        while not done.all():
            act = ~done
            # ---- leaf step: one triangle per iteration ---------------------------------
            enter = act & (tri_cur < 0) & (node < 0)
            tri_cur[enter] = -node[enter] - 1
            inleaf = np.nonzero(act & (tri_cur >= 0))[0]
            if inleaf.size:
                ta = tri_cur[inleaf]
                term = woop_u[ta, 0] == 0x80000000
                # terminator -> leave leaf, pop
                tl = inleaf[term]
                n_leaf += int(tl.size)
                tri_cur[tl] = -1
                m = np.zeros(n, dtype=bool)
                m[tl] = True
                pop(m)
                # real triangle
                ti = inleaf[~term]
                if ti.size:
                    n_tri += int(ti.size)
                    a = tri_cur[ti]
                    z, u4, v4 = woop_f[a], woop_f[a + 1], woop_f[a + 2]
                    rx, ry, rz = ox[ti], oy[ti], oz[ti]
                    ex, ey, ez = dx[ti], dy[ti], dz[ti]
                    Oz = z[:, 3] - rx * z[:, 0] - ry * z[:, 1] - rz * z[:, 2]
                    ooDz = one / _dot4(np.stack([ex, ey, ez, np.zeros_like(ex)], 1), z[:, 0], z[:, 1], z[:, 2], z[:, 3])
                    t = Oz * ooDz
                    ok = (t > tmin[ti]) & (t < tmax[ti])
                    Ou = _dot4(u4, rx, ry, rz, np.full_like(rx, one))
                    Du = _dot4(u4, ex, ey, ez, np.full_like(rx, zero))
                    u = Ou + t * Du
                    ok &= (u >= 0)
                    Ov = _dot4(v4, rx, ry, rz, np.full_like(rx, one))
                    Dv = _dot4(v4, ex, ey, ez, np.full_like(rx, zero))
                    v = Ov + t * Dv
                    ok &= (v >= 0) & ((u + v) <= one)
                    tt = np.where(ok, t, FLT_MAX)          # miss returns FW_F32_MAX
                    acc = (tt > tmin[ti]) & (tt < tmax[ti])  # updateHit (CudaBVH.cpp:1200)
                    hi = ti[acc]
                    tmax[hi] = tt[acc]
                    res_t[hi] = tt[acc]
                    res_id[hi] = tri_index[a[acc]]
                    tri_cur[ti] += 3
                    if any_hit:
                        done[hi] = True                    # return true -> trace returns
                        tri_cur[hi] = -1
            # ---- inner step --------------------------------------------------------------
            inner = np.nonzero(~done & (tri_cur < 0) & (node >= 0) & ~enter)[0]
            if inner.size:
                n_inner += int(inner.size)
                b = node[inner] // 4                        # byte offset -> float index
                g = lambda k: nodes_f[b + k]
                rx, ry, rz = ox[inner], oy[inner], oz[inner]
                ex, ey, ez = dx[inner], dy[inner], dz[inner]

                def box(lox, hix, loy, hiy, loz, hiz):
                    t0x, t0y, t0z = (lox - rx) / ex, (loy - ry) / ey, (loz - rz) / ez
                    t1x, t1y, t1z = (hix - rx) / ex, (hiy - ry) / ey, (hiz - rz) / ez
                    mn = _smax(_smax(_smin(t0x, t1x), _smin(t0y, t1y)), _smin(t0z, t1z))
                    mx = _smin(_smin(_smax(t0x, t1x), _smax(t0y, t1y)), _smax(t0z, t1z))
                    return mn, mx
                mn0, mx0 = box(g(0), g(1), g(2), g(3), g(8), g(9))
                mn1, mx1 = box(g(4), g(5), g(6), g(7), g(10), g(11))
                c0 = nodes_i[b + 12].astype(np.int64)
                c1 = nodes_i[b + 13].astype(np.int64)
                i0 = (mn0 <= mx0) & (mx0 >= tmin[inner]) & (mn0 <= tmax[inner])
                i1 = (mn1 <= mx1) & (mx1 >= tmin[inner]) & (mn1 <= tmax[inner])
                both = i0 & i1
                swp = both & (mn0 > mn1)
                near = np.where(swp, c1, c0)
                far = np.where(swp, c0, c1)
                bi = inner[both]
                if (sp[bi] >= max_stack).any():
                    raise RuntimeError("np_tracer: stack overflow")
                node[bi] = near[both]
                stack[bi, sp[bi]] = far[both]
                sp[bi] += 1
                o0 = i0 & ~i1
                node[inner[o0]] = c0[o0]
                o1 = i1 & ~i0
                node[inner[o1]] = c1[o1]
                none = ~i0 & ~i1
                m = np.zeros(n, dtype=bool)
                m[inner[none]] = True
                pop(m)
                if inner_hook is not None:
                    inner_hook(inner, b * 4, node[inner].copy())
    if return_stats:
        return res_id, res_t, dict(numInnerVisits=n_inner, numTriTests=n_tri, numLeafVisits=n_leaf, numHits=int((res_id >= 0).sum()))
    return res_id, res_t
