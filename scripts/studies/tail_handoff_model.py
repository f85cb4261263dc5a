#!/usr/bin/env python3
"""Model of the tail hand-off BEFORE building it: CPU only (oracle LBVH + oracle per-ray step counts).

A batch is traced by wave-private mini-pools (K x 64 consecutive rays per wave, finished lanes refilled from the wave's own pool below
`refill` live lanes).  Policy under study: once a wave's pool is dry and fewer than T lanes are live, the wave either fills its free lanes
from a global continuation queue (when the queue holds enough) or appends its live rays to the queue and exits -- as long as more than A
waves are still active; below that a wave runs to completion (and only drains what is left in the queue).

All waves advance one iteration per tick (a lane advances one node or one triangle per iteration: the unified step).  Reported per policy:
  wave_iterations   sum over waves of their lifetime in iterations (what the SIMDs issue: the throughput term)
  makespan          last ray finished, in iterations (the critical path incl. queueing)
  est_ms            integral over ticks of max(lat_us(lanes), active_waves(t) x scale x issue_us / simds)  -- a two-regime cost of a tick

usage: tail_handoff_model.py <scene> [pools=512] [K=4]"""
import heapq
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ntrace_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402  (analysis script, not product)
import ctypes as C  # noqa: E402

SIMDS = 1024
ISSUE_US = 0.165      # one wave-iteration of the unified step on a busy SIMD (r03: 18.6 M wave-iterations in 3.0 ms on 1 024 SIMDs)
LAT_FULL_US = 0.9     # a step of a full wave alone on the chip   (profiles/r03_tail_step_latency.jsonl)
LAT_ONE_US = 0.43     # a step of one ray alone in its wave


def per_ray_steps(b, rays):
    L = oracle.lib()
    n = rays.shape[0]
    res = np.zeros(n, dtype=oracle.RESULT_DTYPE)
    inner = np.zeros(n, dtype=np.int32)
    tris = np.zeros(n, dtype=np.int32)
    vp = C.c_void_p
    L.orc_trace_compact_counts.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp]
    L.orc_trace_compact_counts.restype = C.c_int
    rays = np.ascontiguousarray(rays)
    rc = L.orc_trace_compact_counts(b["nodes"].ctypes.data, b["woop"].ctypes.data, b["tri_index"].ctypes.data, rays.ctypes.data,
                                    res.ctypes.data, n, 0, inner.ctypes.data, tris.ctypes.data)
    assert rc == 0
    return (inner + tris).astype(np.int64)


def lat_us(lanes):
    return LAT_ONE_US + (LAT_FULL_US - LAT_ONE_US) * min(max(lanes - 1, 0), 63) / 63.0


class Wave:
    __slots__ = ("fin", "pool", "pp", "start", "end", "lanes_hist")

    def __init__(self, pool):
        self.fin = []          # heap of absolute finish times of the live rays
        self.pool = pool       # step counts of the unstarted rays
        self.pp = 0
        self.start = 0
        self.end = None


def simulate(pools, T=0, A=0, M=None, refill=48, scale=1.0):
    """pools: list of arrays of per-ray steps (zero-step rays = degenerate: retire at once).  T = 0: no hand-off."""
    waves = [Wave(p) for p in pools]
    ev = []   # (time, wave index): the wave has to look at its state at that time
    for i, w in enumerate(waves):
        heapq.heappush(ev, (0, i))
    queue = []          # remaining steps of the continuations, FIFO
    qh = 0
    active = len(waves)
    handed = 0
    last_finish = 0
    timeline = []       # (t, +1 / -1, lanes) for est_ms
    lifetimes = 0
    # per-tick accounting is done afterwards from (start, end) of the waves and their live-lane step functions: too slow in python for lanes;
    # est_ms uses active-wave counts only, with the latency of a "typical" wave taken as full while > A waves are active
    while ev:
        t, i = heapq.heappop(ev)
        w = waves[i]
        while w.fin and w.fin[0] <= t:
            last_finish = max(last_finish, heapq.heappop(w.fin))
        live = len(w.fin)
        # refill from the wave's own pool
        if w.pp < len(w.pool) and live < max(refill, 1):
            while w.pp < len(w.pool) and len(w.fin) < 64:
                s = int(w.pool[w.pp]); w.pp += 1
                if s > 0:
                    heapq.heappush(w.fin, t + s)
            live = len(w.fin)
        dry = w.pp >= len(w.pool)
        if dry and T > 0 and live < T:
            q = len(queue) - qh
            need = (64 - live) if M is None else M
            if q >= need and q > 0:                      # consumer: fill the free lanes
                take = min(q, 64 - live)
                for _ in range(take):
                    heapq.heappush(w.fin, t + queue[qh]); qh += 1
                live = len(w.fin)
            elif active > max(A, 1) and live > 0:        # producer: hand the live rays off, exit (never the last wave)
                for f in w.fin:
                    queue.append(f - t)
                handed += live
                w.fin = []
                live = 0
            elif q > 0:                                  # end game: drain what is left
                take = min(q, 64 - live)
                for _ in range(take):
                    heapq.heappush(w.fin, t + queue[qh]); qh += 1
                live = len(w.fin)
        if live == 0 and dry:
            if T > 0 and len(queue) - qh > 0 and active == 1:
                heapq.heappush(ev, (t, i))   # the last wave drains the queue
                # (falls into the end-game branch above next time round)
                if len(w.fin) == 0:
                    take = min(len(queue) - qh, 64)
                    for _ in range(take):
                        heapq.heappush(w.fin, t + queue[qh]); qh += 1
                continue
            w.end = t
            active -= 1
            lifetimes += t - w.start
            timeline.append((w.start, t))
            continue
        # next time this wave has to look: when enough lanes have finished
        fins = sorted(w.fin)
        if not dry:
            k = live - (refill - 1)            # after k finishes, live < refill
            nt = fins[max(k, 1) - 1] if live >= refill else fins[0]
        elif T > 0 and live >= T:
            nt = fins[live - T]                # live drops below T when (live - T + 1) rays have finished
        else:
            nt = fins[-1]
        heapq.heappush(ev, (max(nt, t + 1), i))
    assert len(queue) - qh == 0, "continuations stranded"
    # est_ms: sweep over time; active(t) from (start, end) intervals
    pts = []
    for (s, e) in timeline:
        pts.append((s, 1)); pts.append((e, -1))
    pts.sort()
    est = 0.0
    act = 0
    prev = 0
    for (tt, d) in pts:
        if tt > prev and act > 0:
            n_full = act * scale
            per_tick = max(n_full * ISSUE_US / SIMDS, LAT_FULL_US if act > max(A, 1) else LAT_ONE_US + 0.2)
            est += (tt - prev) * per_tick
        prev = tt
        act += d
    return dict(wave_iterations=int(lifetimes), makespan=int(last_finish), handed_off=int(handed), est_ms=est * 1e-3)


def main():
    scene = sys.argv[1]
    npools = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    batch = sys.argv[3] if len(sys.argv) > 3 else "incoherent"
    tri, pos, cam = {"hairball": scenes.hairball, "courtyard": scenes.courtyard, "atrium": scenes.atrium}[scene]()
    b = oracle.lbvh_build(tri, pos, 8, 0.001)
    nr = 1 << 21
    rays = scenes.box_rays(pos, nr, seed=21)
    nrays = npools * 256
    sel = rays[:nrays]
    steps = per_ray_steps(b, sel)
    scale = nr / float(nrays)
    print(json.dumps(dict(scene=scene, batch=batch, sample=nrays, mean_steps=float(steps.mean()), max_steps=int(steps.max()),
                          lane_steps=int(steps.sum()))), flush=True)
    for K in (1, 2, 4):
        pools = [steps[i:i + 64 * K] for i in range(0, nrays, 64 * K)]
        base = simulate(pools, T=0, scale=scale)
        print(json.dumps(dict(K=K, policy="none", **base, util=float(steps.sum() / (64.0 * base["wave_iterations"])))), flush=True)
        for T in (8, 16, 24, 32):
            for A in (0, int(1024 / scale), int(3072 / scale)):
                r = simulate(pools, T=T, A=A, scale=scale)
                print(json.dumps(dict(K=K, policy="handoff", T=T, A_full=int(A * scale), **r, util=float(steps.sum() / (64.0 * r["wave_iterations"])),
                                      iter_ratio=base["wave_iterations"] / max(r["wave_iterations"], 1))), flush=True)


if __name__ == "__main__":
    main()
