#!/usr/bin/env python3
"""Model of a per-RAY dispatch order (longest ray first) on the divergent batches, BEFORE building it.  CPU only: oracle LBVH + the
oracle's per-ray step counts, and the two-term cost of a step measured in round 4 (EXPERIMENTS.md, gather roof):

    a tick (every live ray advances one step) lasts  L(n) = max(L0 + c n / R, n / R)   with n = rays in flight,
    R = 56 G requests/s, L0 = 0.45 us (a lone step), c = 0.6 (1.14 us at the knee of 64 k requests, 8.2 us with every lane live)

so a launch costs  sum over ticks of L(n(t)):  at least W / R (W = lane-steps) and at least (ticks) x L0.

Schedules compared (the same rays, the same per-ray steps):
  perray  K    the shipped per-ray launch: pools of K x 64 consecutive rays of the order, one wave each, `slots` waves resident,
               dispatched in order as slots free up
  global  G    the dynamic-fetch kernel: G persistent waves, every free lane takes the next ray of the order (refill below `thr` live lanes)
Orders: buffer order (what ships for box rays: the block order predicted from the top of the tree does nothing for random rays),
`lpt` = rays sorted by their TRUE step count, longest first (the bound of any cost feedback), `lpt_noisy` = sorted by steps x lognormal noise
(what a predictor with that error would give).

The sample is 1/16 of the 2^21-ray batch, run on 1/16 of the chip (slots, G and R scaled).

usage: lpt_order_model.py <scene> [sample=131072]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ntrace_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402  (analysis script, not product)
from tail_handoff_model import per_ray_steps  # noqa: E402

R_FULL = 56e9        # requests / s beyond the L2 (profiles/r04_gather_roof_*.json)
L0 = 0.45e-6
C_LOAD = 0.6
FULL_RAYS = 1 << 21
FULL_SLOTS = 256 * 4 * 7


def tick_s(n, R):
    return max(L0 + C_LOAD * n / R, n / R) if n > 0 else 0.0


def run_perray(steps, order, K, slots, R):
    """Pools of K x 64 positions of `order`; a lane that finishes takes the pool's next ray at once (refill threshold ignored)."""
    s = steps[order]
    npool = (len(s) + 64 * K - 1) // (64 * K)
    pad = npool * 64 * K - len(s)
    s = np.concatenate([s, np.zeros(pad, dtype=s.dtype)]).reshape(npool, 64 * K)
    rem = np.zeros((slots, 64), dtype=np.int64)      # remaining steps per lane
    pool = np.full(slots, -1, dtype=np.int64)        # pool a slot runs
    nxt = np.zeros(slots, dtype=np.int64)            # next unstarted ray of the slot's pool
    next_pool = 0
    t = 0.0
    ticks = 0
    wave_iters = 0
    while True:
        # free slots take the next pools
        free = np.flatnonzero(pool < 0)
        take = min(len(free), npool - next_pool)
        if take > 0:
            sl = free[:take]
            pool[sl] = np.arange(next_pool, next_pool + take)
            nxt[sl] = 0
            next_pool += take
        act = np.flatnonzero(pool >= 0)
        if len(act) == 0:
            break
        # refill the empty lanes of the active waves from their pools (in lane order)
        for _ in range(2):   # (a zero-step ray retires at once: second round fills its lane again)
            empty = rem[act] <= 0
            need = empty.sum(axis=1)
            avail = 64 * K - nxt[act]
            give = np.minimum(need, avail)
            if give.sum() == 0:
                break
            rank = np.cumsum(empty, axis=1) - 1
            sel = empty & (rank < give[:, None])
            rows = np.nonzero(sel)[0]
            src = nxt[act][rows] + rank[sel]
            vals = s[pool[act][rows], src]
            sub = rem[act]
            sub[sel] = vals
            rem[act] = sub
            nxt[act] += give
        live = rem[act] > 0
        n = int(live.sum())
        done = (~live.any(axis=1)) & (nxt[act] >= 64 * K)
        pool[act[done]] = -1
        if n == 0:
            continue
        # advance: jump to the next event (a lane finishing) to keep the python loop short
        sub = rem[act]
        jump = int(sub[live].min())
        t += jump * tick_s(n, R)
        ticks += jump
        wave_iters += jump * int(live.any(axis=1).sum())
        sub[live] -= jump
        rem[act] = sub
    return dict(ms=t * 1e3, ticks=ticks, wave_iterations=int(wave_iters))


def run_global(steps, order, G, R, thr=48):
    s = steps[order]
    s = s[s > 0]
    rem = np.zeros((G, 64), dtype=np.int64)
    nxt = 0
    t = 0.0
    ticks = 0
    total = len(s)
    while True:
        live = rem > 0
        if nxt < total:
            cnt = live.sum(axis=1)
            want = np.flatnonzero(cnt < thr)          # waves below the refill threshold fill ALL their empty lanes
            if len(want):
                empty = ~live[want]
                k = int(empty.sum())
                give = min(k, total - nxt)
                flat = np.flatnonzero(empty.ravel())[:give]
                sub = rem[want].ravel()
                sub[flat] = s[nxt:nxt + give]
                rem[want] = sub.reshape(len(want), 64)
                nxt += give
                live = rem > 0
        n = int(live.sum())
        if n == 0:
            break
        if nxt < total:
            # next event: some wave drops below the threshold -> step one tick at a time would be slow; jump by the smallest remaining
            jump = int(rem[live].min())
        else:
            jump = int(rem[live].min())
        t += jump * tick_s(n, R)
        ticks += jump
        rem[live] -= jump
    return dict(ms=t * 1e3, ticks=ticks)


def main():
    scene = sys.argv[1]
    sample = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 17
    cache = "/tmp/lpt_steps_%s_%d.npy" % (scene, sample)
    if os.path.exists(cache):
        steps = np.load(cache)
    else:
        tri, pos, cam = {"hairball": scenes.hairball, "courtyard": scenes.courtyard}[scene]()
        b = oracle.lbvh_build(tri, pos, 8, 0.001)
        rays = scenes.box_rays(pos, FULL_RAYS, seed=21)[:sample]
        steps = per_ray_steps(b, rays)
        np.save(cache, steps)
    frac = sample / float(FULL_RAYS)
    R = R_FULL * frac
    slots = int(FULL_SLOTS * frac)
    print(json.dumps(dict(scene=scene, sample=sample, mean_steps=float(steps.mean()), max_steps=int(steps.max()), lane_steps=int(steps.sum()),
                          floor_throughput_ms=steps.sum() / R * 1e3, floor_chain_ms=float(steps.max()) * L0 * 1e3)), flush=True)
    rng = np.random.default_rng(5)
    ident = np.arange(sample)
    lpt = np.argsort(-steps, kind="stable")
    orders = [("buffer", ident), ("lpt", lpt)]
    for sigma in (0.5, 1.0):
        orders.append(("lpt_noisy_%.1f" % sigma, np.argsort(-(steps * np.exp(rng.normal(0.0, sigma, sample))), kind="stable")))
    for name, order in orders:
        for K in (1, 4):
            for occ in (7, 4, 2, 1):
                r = run_perray(steps, order, K, max(int(256 * 4 * occ * frac), 1), R)
                print(json.dumps(dict(kernel="perray", order=name, K=K, waves_per_simd=occ, **r)), flush=True)
        for per_cu in (6, 3, 2, 1):
            r = run_global(steps, order, max(int(256 * per_cu * 4 * frac), 1), R)
            print(json.dumps(dict(kernel="global", order=name, blocks_per_cu=per_cu, **r)), flush=True)


if __name__ == "__main__":
    main()
