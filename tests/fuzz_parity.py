#!/usr/bin/env python3
"""Randomised differential test: HIP tracer / LBVH builder vs the CPU oracle, for a wall-clock budget.

Every round draws a scene (triangle soup, scale, clustering), a BVH (host SAH build or on-device
LBVH with a random leaf size), a ray mix (camera rays, random segments, axis-parallel and
zero-component directions, origins on vertices, huge / tiny / inf / NaN / degenerate extents) and
compares hit records bit for bit (id and the 32 bits of t) plus the traversal counters, for every
kernel name and for closest-hit and any-hit.  LBVH rounds also compare the tree in canonical form.

Usage: python tests/fuzz_parity.py [--seconds 240] [--seed 1] > log.json   (needs a GPU).  Test
infrastructure: the oracle is the checker here, as everywhere under tests/."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402

KERNELS = ["fermi_speculative_while_while", "kepler_dynamic_fetch", "tesla_persistent_while_while"]
DEV = "cuda:0"


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(DEV)


def make_scene(rng):
    n = int(10 ** rng.uniform(0.5, 5.2))
    scale = float(10 ** rng.uniform(-3, 4))
    if rng.random() < 0.12:  # magnitudes outside the exact-fast-path preconditions (ntr_bvh_validate clears the flags)
        scale = float(10 ** rng.choice([-30.0, -14.0, 17.0, 25.0]))
    kind = rng.integers(0, 4)
    if kind == 0:      # uniform soup
        c = rng.uniform(-1, 1, size=(n, 1, 3))
        p = c + rng.normal(0, 10 ** rng.uniform(-3, -0.5), size=(n, 3, 3))
    elif kind == 1:    # clustered (deep trees, duplicate Morton codes)
        k = max(1, n // 200)
        centers = rng.uniform(-1, 1, size=(k, 3))
        c = centers[rng.integers(0, k, size=n)][:, None, :] + rng.normal(0, 1e-3, size=(n, 1, 3))
        p = c + rng.normal(0, 10 ** rng.uniform(-6, -2), size=(n, 3, 3))
    elif kind == 2:    # axis-aligned quads on a grid (coplanar geometry, ties in t)
        m = max(1, int(np.sqrt(n / 2)))
        gx, gy = np.meshgrid(np.linspace(-1, 1, m + 1), np.linspace(-1, 1, m + 1))
        z = np.round(rng.uniform(-1, 1, size=gx.shape) * 4) / 4
        v = np.stack([gx, gy, z], -1)
        a, b, cc, d = v[:-1, :-1], v[1:, :-1], v[1:, 1:], v[:-1, 1:]
        p = np.concatenate([np.stack([a, b, cc], 2).reshape(-1, 3, 3), np.stack([a, cc, d], 2).reshape(-1, 3, 3)])
    else:              # long thin slivers
        c = rng.uniform(-1, 1, size=(n, 1, 3))
        d = rng.normal(size=(n, 1, 3))
        p = c + d * np.array([0.0, 1.0, 1.0])[None, :, None] * 0.3 + rng.normal(0, 1e-4, size=(n, 3, 3))
    pos = (p.reshape(-1, 3) * scale).astype(np.float32)
    if rng.random() < 0.3:  # snap some coordinates to zero / signed zero
        mask = rng.random(pos.shape) < 0.02
        pos[mask] = np.where(rng.random(mask.sum()) < 0.5, 0.0, -0.0)
    tri = np.arange(pos.shape[0] // 3 * 3, dtype=np.int32).reshape(-1, 3)
    if rng.random() < 0.15 and tri.shape[0] > 4:  # degenerate triangles: repeated vertices -> non-finite Woop rows
        k = rng.integers(0, tri.shape[0], size=max(1, tri.shape[0] // 50))
        tri[k, 2] = tri[k, 1]
    return tri, pos, scale


def make_rays(rng, pos, scale, n):
    lo, hi = pos.min(0).astype(np.float64), pos.max(0).astype(np.float64)
    ext = np.maximum(hi - lo, 1e-6 * scale)
    o = lo + rng.uniform(-0.3, 1.3, size=(n, 3)) * ext
    t = lo + rng.uniform(0, 1, size=(n, 3)) * ext
    d = t - o
    kind = rng.integers(0, 10, size=n)
    d[kind == 0] = np.eye(3)[rng.integers(0, 3, size=(kind == 0).sum())] * rng.choice([-1.0, 1.0], size=((kind == 0).sum(), 1))
    z = kind == 1
    d[z, rng.integers(0, 3, size=z.sum())] = rng.choice([0.0, -0.0], size=z.sum())       # one zero component
    tiny = kind == 2
    d[tiny, rng.integers(0, 3, size=tiny.sum())] *= 10 ** rng.uniform(-45, -20, size=tiny.sum())  # tiny / denormal component
    onv = kind == 3
    o[onv] = pos[rng.integers(0, pos.shape[0], size=onv.sum())]                           # origin on a vertex
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    unit = (kind % 2 == 0)[:, None] & (nrm > 0)
    d = np.where(unit, d / np.maximum(nrm, 1e-300), d)
    rays = np.zeros(n, dtype=nt.RAY_DTYPE)
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    for i, k in enumerate("xyz"):
        rays["o" + k], rays["d" + k] = o32[:, i], d32[:, i]
    diag = float(np.linalg.norm(ext))
    rays["tmin"] = np.where(rng.random(n) < 0.8, 0.0, rng.uniform(0, 0.5, size=n) * diag).astype(np.float32)
    tm = rng.uniform(0, 3, size=n) * diag
    special = rng.integers(0, 40, size=n)
    tm = np.where(special == 0, np.inf, tm)
    tm = np.where(special == 1, np.nan, tm)
    tm = np.where(special == 2, -1.0, tm)            # degenerate AO ray (RayGenKernels.cu:232)
    tm = np.where(special == 3, 3.4e38, tm)
    rays["tmax"] = tm.astype(np.float32)
    return rays


def gpu_lbvh(tri, pos, leaf, eps):
    n = tri.shape[0]
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    dn = torch.zeros(capn, dtype=torch.uint8, device=DEV)
    dw = torch.zeros(capw, dtype=torch.uint8, device=DEV)
    di = torch.zeros(capi, dtype=torch.uint8, device=DEV)
    mn, mx = oracle.scene_bbox(pos)
    res = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, leaf, eps, dn.data_ptr(), capn, dw.data_ptr(), capw,
                        di.data_ptr(), capi)
    torch.cuda.synchronize()
    nodes = dn.cpu().numpy()[:res.nodesBytes].copy()
    woop = dw.cpu().numpy()[:res.triWoopBytes].copy()
    idx = di.cpu().numpy()[:res.triIndexBytes].view(np.int32).copy()
    return nodes, woop, idx, res, (dn, dw, di)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rays", type=int, default=200000)
    ap.add_argument("--progress", default=None, help="file that receives the running totals about once a minute (survives a killed run)")
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    env_at_start = {k: v for k, v in os.environ.items() if k.startswith("NTR_")}
    cores = os.cpu_count() or 1
    t_end = time.time() + args.seconds
    tot = dict(rounds=0, sah_rounds=0, lbvh_rounds=0, lbvh_tree_mismatch=0, rays_compared=0, record_mismatches=0, counter_mismatches=0,
               fast_path_rounds=0, triangles=0)
    failures = []
    t_progress = time.time()
    while time.time() < t_end:
        if args.progress and time.time() - t_progress > 60.0:
            t_progress = time.time()
            with open(args.progress + ".tmp", "w") as pf:
                json.dump(dict(tot, failures=len(failures), seed=args.seed, elapsed=round(args.seconds - (t_end - t_progress))), pf)
            os.replace(args.progress + ".tmp", args.progress)
        tri, pos, scale = make_scene(rng)
        use_lbvh = rng.random() < 0.5
        leaf = int(rng.choice([1, 2, 4, 8, 16]))
        eps = float(rng.choice([0.0, 0.001, 0.001 * scale]))
        keep = None
        try:
            if use_lbvh:
                # build paths drawn at random: bottom-up emit (tile meetings through LDS or all through memory) or the
                # cell-table top pass, with the default or a small hand-over size
                nt.set_tunables(NTR_LBVH_SPLIT=int(rng.choice([2, 16, 100, 3000])) if rng.random() < 0.3 else None,
                                NTR_LBVH_AGG_LDS=int(rng.choice([1, 1, 0])),
                                NTR_LBVH_AGG_STAGED=int(rng.choice([-1, 0, 1])),
                                NTR_LBVH_SORT_ITEMS=int(rng.choice([0, 0, 8, 16, 24, 32])),
                                NTR_LBVH_MORTON_THREADS=int(rng.choice([0, 0, 256, 512, 1024])), NTR_LBVH_MORTON_KEYS=int(rng.choice([0, 0, 1, 2, 16])),
                                NTR_LBVH_MARK_THREADS=int(rng.choice([0, 256, 1024])))
                if os.environ.get("NTR_FUZZ_VERBOSE"):
                    print("lbvh n=%d leaf=%d eps=%g %s" % (tri.shape[0], leaf, eps, {k: v for k, v in os.environ.items() if k.startswith("NTR_LBVH")}),
                          file=sys.stderr, flush=True)
                nodes, woop, idx, res, keep = gpu_lbvh(tri, pos, leaf, eps)
                ref = oracle.lbvh_build(tri, pos, leaf, eps)
                same = (res.numNodes == ref["num_inner"] and res.numLeaves == ref["num_leaves"] and res.numLevels == ref["num_levels"] and
                        oracle.bvh_canonical_hash(nodes, woop, idx) == oracle.bvh_canonical_hash(ref["nodes"], ref["woop"], ref["tri_index"]))
                tot["lbvh_rounds"] += 1
                if not same:
                    tot["lbvh_tree_mismatch"] += 1
                    failures.append(dict(kind="lbvh", round=tot["rounds"], tris=int(tri.shape[0]), leaf=leaf, eps=eps, scale=scale,
                                         split=os.environ.get("NTR_LBVH_SPLIT"), gpu=(res.numNodes, res.numLeaves, res.numLevels),
                                         cpu=(ref["num_inner"], ref["num_leaves"], ref["num_levels"]),
                                         woop_only=oracle.bvh_canonical_hash(nodes, woop, idx, hash_woop=False) ==
                                         oracle.bvh_canonical_hash(ref["nodes"], ref["woop"], ref["tri_index"], hash_woop=False)))
                dn, dw, di = keep
                view = nt.BvhView(dn.data_ptr(), res.nodesBytes, dw.data_ptr(), res.triWoopBytes, di.data_ptr())
            else:
                bvh = nt.sah_build(tri, pos, 1, int(rng.choice([1, 4, 8])))
                nodes, woop, idx = bvh.nodes, bvh.woop, bvh.tri_index
                keep = (up(nodes), up(woop), up(idx))
                view = nt.BvhView(keep[0].data_ptr(), nodes.nbytes, keep[1].data_ptr(), woop.nbytes, keep[2].data_ptr())
                tot["sah_rounds"] += 1
            flags = view.validate()
            tot["fast_path_rounds"] += 1 if (flags & 2) else 0
            tot["generic_path_rounds"] = tot.get("generic_path_rounds", 0) + (0 if (flags & 2) else 1)
            rays = make_rays(rng, pos, scale, int(rng.integers(1, args.rays)))
            d_rays = up(rays)
            n = rays.shape[0]
            # the kernels answer degenerate rays (not tmin < tmax: Ray::degenerate, NaN extents) without
            # traversal, so the traversal counters are defined over the other rays
            live = rays[rays["tmin"] < rays["tmax"]]
            # scheduling variants only permute blocks: plain, dispatch-order prediction forced on, caller-owned hint
            sched = int(rng.integers(0, 4))   # 3: a caller-owned hint that starts from PREDICTED block costs (random ones: any order is valid)
            if sched == 1:
                nt.set_tunables(NTR_TRACE_PREDICT_MIN_RAYS=1, NTR_TRACE_PREDICT_MIN_NODES=1)
            else:
                nt.set_tunables(NTR_TRACE_PREDICT_MIN_RAYS=None, NTR_TRACE_PREDICT_MIN_NODES=None)
            # loop variants only change how the lanes of a wave interleave: while-while or unified-step loop in the per-ray kernel (by the
            # tree's leaf sizes, or forced either way) and in kepler_dynamic_fetch, any dynamic-fetch threshold, any mini-pool depth
            loop = dict(NTR_TRACE_PERRAY_UNIFIED=int(rng.choice([-1, 0, 1])), NTR_TRACE_UNIFIED=int(rng.choice([1, 1, 0])),
                        NTR_TRACE_FETCH_THRESHOLD=int(rng.choice([-1, -1, 1, 16, 33, 64])), NTR_TRACE_FLAT_FETCH=int(rng.choice([1, 1, 0])),
                        # wave-private mini-pool of the closest-hit per-ray launches: by the device's coherence estimate, off, or forced
                        NTR_TRACE_MINIPOOL=int(rng.choice([-1, -1, 0, 1, 2, 3, 4, 5, 8, 16])), NTR_TRACE_MINIPOOL_THRESHOLD=int(rng.choice([48, 48, 1, 33, 64])),
                        # ray splitting in the drain phase of the persistent kernels (trace_split.h): how often the lanes are looked at
                        NTR_TRACE_SPLIT_SLICE=int(rng.choice([8, 8, 1, 2, 5, 32, 0])),
                        # round 6: routing by coherence (both bodies launched, the device's batch word picks one) or the named body always;
                        # the persistent kernels' refill policy, dequeue-ahead, pool geometry and hint support
                        NTR_TRACE_ROUTE=int(rng.choice([1, 1, 0])), NTR_TRACE_WHOLE_WAVE=int(rng.choice([1, 1, 0])),
                        NTR_TRACE_PREFETCH_AFTER=int(rng.choice([8, 8, -1, 0, 1, 50])), NTR_TRACE_PERSISTENT_HINTS=int(rng.choice([1, 1, 0])),
                        NTR_TRACE_POOL_HEADS=int(rng.choice([128, 128, 8, 64, 1024])), NTR_TRACE_BLOCKS_PER_CU=int(rng.choice([8, 8, 1, 3, 6])),
                        NTR_TRACE_CHUNK=int(rng.choice([64, 64, 64, 32, 128, 256, 48])))
            nt.set_tunables(**loop)
            hint = nt.SchedHint() if sched >= 2 else None
            if sched == 3:
                nbk = (n + 255) // 256
                d_pc = torch.from_numpy(rng.integers(0, int(rng.choice([1, 2, 50, 1 << 20])), nbk).astype(np.int32)).to(DEV)
                hint.predict(d_pc.data_ptr(), nbk)
            tot["sched_%d_rounds" % sched] = tot.get("sched_%d_rounds" % sched, 0) + 1
            for any_hit in (False, True):
                exp, _ = oracle.trace(nodes, woop, idx, rays, any_hit=any_hit, threads=cores)
                _, est = oracle.trace(nodes, woop, idx, live, any_hit=any_hit, threads=cores)
                for kernel in KERNELS:
                    d_res = torch.full((n * 16,), 0xCD, dtype=torch.uint8, device=DEV)
                    st = view.trace_stats(kernel, n, any_hit, d_rays.data_ptr(), d_res.data_ptr())
                    got = d_res.cpu().numpy().view(nt.RESULT_DTYPE)
                    bad = int(((got["id"] != exp["id"]) | (got["t"].view(np.uint32) != exp["t"].view(np.uint32))).sum())
                    cbad = int((st.numInnerVisits, st.numTriTests, st.numLeafVisits, st.numHits) !=
                               (est.numInnerVisits, est.numTriTests, est.numLeafVisits, est.numHits))
                    d_res.fill_(0xCD)
                    for rep in range(3 if hint is not None else 1):  # a hint changes the order from its second use on
                        d_res.fill_(0xCD)
                        view.trace(kernel, n, any_hit, d_rays.data_ptr(), d_res.data_ptr(), hint=hint)
                        torch.cuda.synchronize()
                        got2 = d_res.cpu().numpy().view(nt.RESULT_DTYPE)
                        bad += int(((got2["id"] != exp["id"]) | (got2["t"].view(np.uint32) != exp["t"].view(np.uint32))).sum())
                        tot["rays_compared"] += n
                    tot["rays_compared"] += n
                    tot["record_mismatches"] += bad
                    tot["counter_mismatches"] += cbad
                    if bad or cbad:
                        failures.append(dict(kind="trace", round=tot["rounds"], kernel=kernel, any_hit=any_hit, bad=bad, counters=cbad,
                                             tris=int(tri.shape[0]), lbvh=bool(use_lbvh), scale=scale, flags=int(flags),
                                             gpu=st.as_dict(), cpu=est.as_dict()))
        except Exception as e:  # a crash is a failure too
            failures.append(dict(kind="exception", round=tot["rounds"], err=repr(e), tris=int(tri.shape[0]), lbvh=bool(use_lbvh)))
        tot["rounds"] += 1
        tot["triangles"] += int(tri.shape[0])
        del keep
        os.environ.pop("NTR_TRACE_PREDICT_MIN_RAYS", None)
    # leave no tunable of the last round behind: run in-process (tests/test_fuzz_gpu.py) the draws would otherwise steer the tests that
    # follow -- e.g. NTR_TRACE_PERRAY_UNIFIED=0 takes the per-ray launch off its pools, and the hand-off tests then hand nothing off
    nt.set_tunables(**{k: env_at_start.get(k) for k in set(list(os.environ) + list(env_at_start)) if k.startswith("NTR_TRACE_") or k.startswith("NTR_LBVH_")})
    tot["failures"] = [f for f in failures if f["kind"] != "trace"][:20] + [f for f in failures if f["kind"] == "trace"][:6]
    tot["seed"] = args.seed
    tot["seconds"] = args.seconds
    print(json.dumps(tot))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
