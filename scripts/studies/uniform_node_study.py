#!/usr/bin/env python3
"""How often do the lanes of a wave sit on the SAME inner node?  CPU simulation of the per-ray kernel's while-while schedule
(inner loop while any lane holds an inner node, leaf-switch threshold 24, whole leaf per leaf phase) on sampled 64-ray waves of
the bench workload (atrium-262k SAH, 1080p primary batch; optionally one AO batch).  Prints, per batch, the histogram of distinct
nodes per inner wave-iteration: a wave-uniform fetch could go through the scalar cache (no TA cycles).  No GPU needed."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402

F = np.float32
FLT_MAX = np.float32(3.4028234663852886e38)


def simulate(nodes, woop, rays, any_hit, switch_below=24):
    nodes_f = np.frombuffer(np.ascontiguousarray(nodes).tobytes(), dtype=F)
    nodes_i = nodes_f.view(np.int32)
    woop_f = np.frombuffer(np.ascontiguousarray(woop).tobytes(), dtype=F).reshape(-1, 4)
    woop_u = woop_f.view(np.uint32)
    n = rays.shape[0]
    W = n // 64
    ox, oy, oz = rays["ox"].astype(F), rays["oy"].astype(F), rays["oz"].astype(F)
    dx, dy, dz = rays["dx"].astype(F), rays["dy"].astype(F), rays["dz"].astype(F)
    tmin, tmax = rays["tmin"].astype(F), rays["tmax"].astype(F).copy()
    SENT = np.int64(0x76543210)
    node = np.where(tmin < tmax, 0, SENT).astype(np.int64)
    stack = np.zeros((n, 128), dtype=np.int64)
    sp = np.zeros(n, dtype=np.int64)
    hist = np.zeros(65, dtype=np.int64)
    lanes_hist = np.zeros(65, dtype=np.int64)
    wave_of = np.arange(n) // 64

    def pop(idx):
        has = sp[idx] > 0
        i1 = idx[has]
        sp[i1] -= 1
        node[i1] = stack[i1, sp[i1]]
        node[idx[~has]] = SENT

    with np.errstate(all="ignore"):
        while True:
            inner = (node >= 0) & (node < SENT)
            leaf = node < 0
            ic = np.bincount(wave_of[inner], minlength=W)
            lc = np.bincount(wave_of[leaf], minlength=W)
            if (ic + lc).sum() == 0:
                break
            inner_mode = (ic > 0) & ~((ic < switch_below) & (lc > 0))
            # ---- inner step for the waves in inner mode
            step = np.nonzero(inner & inner_mode[wave_of])[0]
            if step.size:
                w = wave_of[step]
                # distinct nodes per wave
                key = w * (1 << 40) + node[step]
                uq = np.unique(key)
                dist = np.bincount((uq >> 40).astype(np.int64), minlength=W)
                lanes = np.bincount(w, minlength=W)
                ws = np.nonzero(dist)[0]
                np.add.at(hist, np.minimum(dist[ws], 64), 1)
                np.add.at(lanes_hist, np.minimum(dist[ws], 64), lanes[ws])
                b = node[step] // 4
                g = lambda k: nodes_f[b + k]
                rx, ry, rz, ex, ey, ez = ox[step], oy[step], oz[step], dx[step], dy[step], dz[step]

                def box(lox, hix, loy, hiy, loz, hiz):
                    t0x, t0y, t0z = (lox - rx) / ex, (loy - ry) / ey, (loz - rz) / ez
                    t1x, t1y, t1z = (hix - rx) / ex, (hiy - ry) / ey, (hiz - rz) / ez
                    mn = np.maximum(np.maximum(np.minimum(t0x, t1x), np.minimum(t0y, t1y)), np.minimum(t0z, t1z))
                    mx = np.minimum(np.minimum(np.maximum(t0x, t1x), np.maximum(t0y, t1y)), np.maximum(t0z, t1z))
                    return mn, mx
                mn0, mx0 = box(g(0), g(1), g(2), g(3), g(8), g(9))
                mn1, mx1 = box(g(4), g(5), g(6), g(7), g(10), g(11))
                c0, c1 = nodes_i[b + 12].astype(np.int64), nodes_i[b + 13].astype(np.int64)
                i0 = (mn0 <= mx0) & (mx0 >= tmin[step]) & (mn0 <= tmax[step])
                i1 = (mn1 <= mx1) & (mx1 >= tmin[step]) & (mn1 <= tmax[step])
                swp = i1 & (~i0 | (mn0 > mn1))
                near, far = np.where(swp, c1, c0), np.where(swp, c0, c1)
                both = i0 & i1
                bi = step[both]
                stack[bi, sp[bi]] = far[both]
                sp[bi] += 1
                anyc = i0 | i1
                node[step[anyc]] = near[anyc]
                pop(step[~anyc])
            # ---- leaf phase for the other waves: the whole leaf, then pop
            lf = np.nonzero(leaf & ~inner_mode[wave_of])[0]
            while lf.size:
                a = -node[lf] - 1
                term = woop_u[a, 0] == 0x80000000
                pop(lf[term])
                ti, a = lf[~term], a[~term]
                if ti.size:
                    z, u4, v4 = woop_f[a], woop_f[a + 1], woop_f[a + 2]
                    rx, ry, rz, ex, ey, ez = ox[ti], oy[ti], oz[ti], dx[ti], dy[ti], dz[ti]
                    Oz = z[:, 3] - rx * z[:, 0] - ry * z[:, 1] - rz * z[:, 2]
                    t = Oz * (F(1.0) / (ex * z[:, 0] + ey * z[:, 1] + ez * z[:, 2]))
                    ok = (t > tmin[ti]) & (t < tmax[ti])
                    u = (u4[:, 0] * rx + u4[:, 1] * ry + u4[:, 2] * rz + u4[:, 3]) + t * (u4[:, 0] * ex + u4[:, 1] * ey + u4[:, 2] * ez)
                    v = (v4[:, 0] * rx + v4[:, 1] * ry + v4[:, 2] * rz + v4[:, 3]) + t * (v4[:, 0] * ex + v4[:, 1] * ey + v4[:, 2] * ez)
                    ok &= (u >= 0) & (v >= 0) & ((u + v) <= F(1.0))
                    hit = ti[ok]
                    tmax[hit] = t[ok]
                    node[ti] -= 3                              # next triangle of the leaf
                    if any_hit:
                        node[hit] = SENT
                        sp[hit] = 0
                        ti = ti[~ok]
                lf = ti
    return hist, lanes_hist


def main():
    tri, pos, cam = scenes.atrium()
    bvh = nt.sah_build(tri, pos)
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    rng = np.random.default_rng(1)
    nw = int(os.environ.get("WAVES", "400"))
    waves = np.sort(rng.choice(rays.shape[0] // 64, nw, replace=False))
    sel = (waves[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
    hist, lanes = simulate(bvh.nodes, bvh.woop, rays[sel], False)
    tot = hist.sum()
    out = dict(batch="primary", waves=nw, inner_wave_iterations=int(tot), uniform_frac=float(hist[1] / tot),
               le2_frac=float(hist[1:3].sum() / tot), le4_frac=float(hist[1:5].sum() / tot), le8_frac=float(hist[1:9].sum() / tot),
               mean_distinct=float((hist * np.arange(65)).sum() / tot), mean_active_lanes=float(lanes.sum() / tot),
               uniform_mean_lanes=float(lanes[1] / max(hist[1], 1)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
