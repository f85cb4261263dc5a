#!/usr/bin/env python3
"""bench.py -- Mrays/s of the BVH trace hot path on MI355X (driver contract in the task brief).

A step = one pass of the hot path over one frame's ray batches on the 'atrium-262k' stand-in for
Crytek Sponza (sponza.obj is not in the reference checkout): the 1920x1080 primary batch (closest
hit) followed by the 8 x AO batches (any hit, <= 2^20 rays per batch as Renderer.cpp:45 /
RayGen.cpp:582-602), all resident in HBM before the timed region.  Ray generation is excluded from
the metric exactly as in the reference's runBenchmark (App.cpp:955-969).

Protocol = the reference's: every batch is one launch on one stream, bracketed by HIP events;
`value` = non-degenerate rays / sum of the per-batch kernel times (App.cpp:955-969), MAX over ranks.
The wall-clock rate of the same K steps and -- on one GPU -- the rate of the frame replayed as a HIP
graph with the AO batches on three streams are reported beside it (`wall_mrays`,
`extras.overlapped_frame`).

Multi-GPU (`--gpus N`; run without a launcher this script starts its own N ranks): ONE frame is
sharded by screen tile -- rank r traces the r-th contiguous 64-aligned range of the PixelTable
index space and the AO rays of its own primary hits, against a BVH built once on rank 0 and
broadcast; the only collective is the final gather of hit records to rank 0 over RCCL, timed
separately.  Rank 0 then checks the assembled frame bit for bit against its own single-GPU trace.
`--scaling weak` keeps the round-1 mode (every rank traces a full frame of its own camera)."""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
# MI355X_MICROARCH.md, 'Indexed rows: gather' table: chip-wide rate of row gathers served by the XCD L2s
# (16.8-18.8 TB/s) and by the Infinity Cache (8.6 TB/s) -- the practical ceilings of a cache-resident BVH.
L2_GATHER_PEAK_GBS = 18800.0
L2_STREAM_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md, L2 section: ~34.5 TB/s aggregate over the eight XCD L2s
MALL_GATHER_PEAK_GBS = 8600.0
NUM_SIMDS = 1024           # 256 CUs x 4 SIMD-32
NUM_TAS = 256              # one texture-address unit per CU
PEAK_CLOCK_GHZ = 2.4


# kernel symbol that WORKS for a batch (ntr_api.cpp / trace_plan.h ROUTING: the names select semantics, the body is chosen by the batch's
# coherence; template arguments <WAVES, STATS, UNIFIED, FLATF> / <WAVES, UNIFIED, FLATF>), and the grid it is launched with for n rays
def launched_symbol(kernel, wide_leaves=False, any_hit=False, incoherent=False):
    """Any-hit launches run trace_bvh_perray<1, false, true, true> (unified-step loop) under every name; coherent closest-hit launches
    trace_bvh_perray_mini (the instantiation that reads the batch word); closest-hit launches the device finds incoherent the persistent
    body: kepler_dynamic_fetch's trace_bvh_persistent<4, true, true> (also for the per-ray name), tesla's <4, false, true>."""
    if any_hit:
        return "trace_bvh_perray<1, false, true, true>"
    if not incoherent:
        return "trace_bvh_perray_mini"
    return "trace_bvh_persistent<4, %s, true>" % ("false" if kernel.startswith("tesla") else "true")


def launched_grid(kernel, n_rays, cus=256, symbol=None):
    if symbol is not None and "persistent" in symbol:
        return min(cus * 7, (n_rays + 255) // 256) * 256
    return ((n_rays + 255) // 256) * 256


def load_pmc(symbol, grid, tag=None):
    """Per-dispatch counter means of the launches of exactly `symbol` with `grid` work-items from the newest committed rocprofv3
    PMC summary (profiles/*_pmc_summary.json, written by scripts/summarize_rocprof.py from separate --pmc passes; `tag` restricts
    the search to files whose name contains it).  Returns (dict counter -> mean, file name) or ({}, None)."""
    best, src = {}, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        if tag and tag not in os.path.basename(f):
            continue
        try:
            d = json.load(open(f))
        except Exception:
            continue
        got = {}
        for k, v in d.items():
            parts = k.split("|")
            if len(parts) == 3 and parts[0].strip() == symbol and parts[1] == str(grid):
                got[parts[2]] = v
        if got:
            best, src = got, os.path.basename(f)
    return best, src


def l1_roofline(alg_bytes, sec, cus):
    """SURVEY 8(d) algorithmic bytes against the data path that actually serves a cache-resident BVH: the CUs' vector L1s, 64 bytes per
    clock each at the peak clock.  Recomputable from the line alone: achieved = algorithmic_bytes_per_launch / launch_ms."""
    peak = cus * 64.0 * PEAK_CLOCK_GHZ          # GB/s
    ach = alg_bytes / sec / 1e9
    return {"bound": "l1-data-path", "achieved": ach, "peak": peak, "unit": "GB/s", "frac": ach / peak,
            "algorithmic_bytes_per_launch": int(alg_bytes), "launch_ms": sec * 1e3,
            "peak_definition": "%d CUs x 64 B/clk (vector L1) x %.1f GHz peak clock" % (cus, PEAK_CLOCK_GHZ)}


def gather_like_for_like(lane_steps, sec, gr, pmc):
    """The HBM-resident launch's fetch rates against the gather roofs measured in the same run, like with like:
    * beyond_l2: the launch's L2 MISSES (TCC_MISS_sum of the profiled launch of the same symbol) per second of that PROFILED launch
      (its own duration, `TCC_MISS_sum@ns`) against the roof of fetches that all go beyond the L2 (768 MB table);
    * all_fetches: every lane step of this run's launch per second against the roof of fetches that all hit the L2 (2 MB table) --
      an upper bound nobody reaches with 45 % misses, kept only beside that roof."""
    out = {"all_fetches": {"records_per_s": lane_steps / sec, "roof_records_per_s": (gr.get("l2_resident_grecords_per_s") or 0) * 1e9 or None,
                           "frac": (lane_steps / sec) / (gr["l2_resident_grecords_per_s"] * 1e9) if gr.get("l2_resident_grecords_per_s") else None,
                           "roof": "dependent random 64-byte fetches, 2 MB table (L2-resident)"}}
    miss, ns = (pmc or {}).get("TCC_MISS_sum"), (pmc or {}).get("TCC_MISS_sum@ns") or (pmc or {}).get("GRBM_GUI_ACTIVE@ns")
    if miss and ns and gr.get("grecords_per_s"):
        rate = miss / (ns * 1e-9)
        out["beyond_l2"] = {"misses_per_launch": miss, "profiled_launch_ms": ns * 1e-6, "misses_per_s": rate, "roof_records_per_s": gr["grecords_per_s"] * 1e9,
                            "frac": rate / (gr["grecords_per_s"] * 1e9), "roof": "dependent random 64-byte fetches, 768 MB table (every fetch beyond the L2)"}
        out["frac"] = out["beyond_l2"]["frac"]
    else:
        out["beyond_l2"] = None
        out["frac"] = None
        out["note"] = "no PMC summary with TCC_MISS_sum and its launch duration (`@ns`, round-5 summaries) for this symbol: the like-for-like beyond-L2 fraction is not reported"
    return out


def profiled_shares(pmc, pmc_src, visits=None):
    """Busy SHARES (not bounds) of the units a trace launch keeps busy, as ONE profiled launch measured them: the per-dispatch PMC means of
    the same kernel symbol and launch shape from a committed rocprofv3 summary (named in `source`), every count divided by the cycles of
    the SAME PMC pass (`<counter>@cycles` = that pass's GRBM_GUI_ACTIVE / 8; summaries older than round 5 carry one GRBM pass, used for all
    counters and named in `cycles_from`).  Nothing here is divided by THIS run's launch time (that mixed two runs and read
    an effective clock above the chip's 2.4 GHz).  VALU issue: a wave64 instruction occupies its SIMD-32 for two cycles; TA: one
    texture-address unit per CU; wave_wait_share: SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
    if not pmc:
        return None

    def cyc(counter):   # cycles of the pass that collected `counter`
        return pmc.get(counter + "@cycles") or (pmc.get("GRBM_GUI_ACTIVE", 0.0) / 8.0) or None

    def ns(counter):
        return pmc.get(counter + "@ns") or pmc.get("GRBM_GUI_ACTIVE@ns")

    b = {"source": "profiles/" + pmc_src,
         "basis": "per-dispatch counts of the profiled launch / cycles of the same PMC pass (GRBM_GUI_ACTIVE / 8); this run's time is not used",
         "cycles_from": "per pass" if any(k.endswith("@cycles") for k in pmc) else "the summary's one GRBM_GUI_ACTIVE pass"}
    c_sq = cyc("SQ_INSTS_VALU")
    if c_sq:
        b["profiled_cycles"] = c_sq
        if ns("SQ_INSTS_VALU"):
            b["profiled_launch_us"] = ns("SQ_INSTS_VALU") * 1e-3
            b["profiled_clock_ghz"] = c_sq / ns("SQ_INSTS_VALU")
    if "SQ_INSTS_VALU" in pmc and c_sq:
        b["valu_issue_frac"] = pmc["SQ_INSTS_VALU"] * 2.0 / (NUM_SIMDS * c_sq)
    # TA_TA_BUSY counts every kind of vector-memory instruction (the unified-step loop fetches with global loads, which TA_BUFFER_TOTAL_CYCLES does not see)
    if "TA_TA_BUSY_sum" in pmc and cyc("TA_TA_BUSY_sum"):
        b["ta_frac"] = pmc["TA_TA_BUSY_sum"] / NUM_TAS / cyc("TA_TA_BUSY_sum")
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:  # KiB; gfx950: FETCH_SIZE counts 64 B per 128-B request
        b["hbm_traffic_bytes"] = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        if ns("FETCH_SIZE"):   # over the FETCH_SIZE pass's own launch duration
            b["hbm_traffic_frac"] = b["hbm_traffic_bytes"] / (ns("FETCH_SIZE") * 1e-9) / 1e9 / HBM_PEAK_GBS
    if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc and (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]) > 0:
        b["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
        if "FETCH_SIZE" in pmc and pmc["TCC_MISS_sum"] > 0:   # bytes the fabric moved per byte a 64-byte record miss consumes (128-byte line fills)
            b["line_overfetch"] = 2.0 * pmc["FETCH_SIZE"] * 1024.0 / (64.0 * pmc["TCC_MISS_sum"])
    if visits and "SQ_INSTS_VMEM_RD" in pmc:  # 4 loads per wave-iteration (node: 4 x 16 B; triangle: 3 x 16 B + 4 B)
        b["lane_util"] = visits / (pmc["SQ_INSTS_VMEM_RD"] / 4.0 * 64.0)
    if "SQ_WAIT_ANY" in pmc and "SQ_WAVE_CYCLES" in pmc and pmc["SQ_WAVE_CYCLES"] > 0:
        b["wave_wait_share"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
    return b


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250,
                    help="timed steps (frames); the default keeps the timed region above 0.25 s -- a step is ~1.2 ms on one MI355X -- so that "
                         "utilisation samplers see the GPU busy")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default): ONE frame sharded by screen tile over the ranks; weak: a full frame per rank")
    ap.add_argument("--balance", choices=("predicted", "count"), default="count",
                    help="strong scaling: ranges of equal ray count (default), or of equal predicted cost (ntr_predict_block_costs; "
                         "measured no better on the simulated ranks: profiles/r03_shard_balance_study.jsonl)")
    ap.add_argument("--flat-share", type=float, default=4.0,
                    help="balanced cuts: cost of a block that does not depend on where its rays go (ray load / store, its AO rays), "
                         "as a multiple of the mean predicted cost (scripts/studies/shard_balance_study.py)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--ao-samples", type=int, default=8)
    ap.add_argument("--ao-radius", type=float, default=5.0, help="Raygen.aoRadius of the reference's config.conf")
    ap.add_argument("--kernel", default=os.environ.get("NTR_BENCH_KERNEL", "fermi_speculative_while_while"))
    ap.add_argument("--scene-obj", default=os.environ.get("NTR_SCENE_OBJ", ""),
                    help="Wavefront OBJ to trace instead of the procedural stand-in (e.g. the real sponza.obj)")
    ap.add_argument("--camera", default=os.environ.get("NTR_CAMERA", ""),
                    help="NTrace camera signature (CameraControls::encodeSignature) for --scene-obj")
    ap.add_argument("--ao-batch-rays", type=int, default=1 << 20,
                    help="maxBatchSize of RayGen (the reference constructs it with 1 << 20, Renderer.cpp:45)")
    ap.add_argument("--ao-streams", type=int, default=3, help="HIP streams of the overlapped-frame figure (extras)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default=os.environ.get("NTR_BENCH_DIST_BACKEND", "nccl"),
                    help="process-group backend for N > 1: nccl (= RCCL over xGMI, one rank per GPU: what the driver runs), or gloo (host "
                         "staging) -- with --one-device the way to execute the N > 1 path with real kernels on a ONE-GPU box")
    ap.add_argument("--one-device", action="store_true", default=os.environ.get("NTR_BENCH_ONE_DEVICE") == "1",
                    help="every rank uses cuda:0 (needs --dist-backend gloo: RCCL refuses two ranks on one GPU).  The ranks' kernels share "
                         "the chip, so the line's times are NOT a scaling measurement; the sharded frame -> gather -> bit-compare flow is what runs")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (profiling runs)")
    ap.add_argument("--no-ao-prediction", action="store_true",
                    help="trace the AO batches without the leaf-depth dispatch hint made beside ray generation (buffer order until a launch has measured)")
    ap.add_argument("--no-hbm-point", action="store_true", help="skip the 10 M-triangle HBM-resident roofline point (extras)")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs 3-5 table (extras.configs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold-order", action="store_true", help="skip the extra K steps with the scheduling feedback off")
    ap.add_argument("--cpu-sample-rays", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process, which has not
    touched the GPU (one process per GPU over RCCL; never exec from a process that has initialised HIP)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def scene_and_camera(args, nt, scenes):
    scene_name = "atrium-262k stand-in for Crytek Sponza, seed 262267; sponza.obj is absent from the reference checkout"
    tri, pos, cam = scenes.atrium()
    if args.scene_obj:
        tri, pos, _ = nt.obj_load(args.scene_obj)
        scene_name = "OBJ %s (%d triangles)" % (os.path.basename(args.scene_obj), tri.shape[0])
        lo, hi = pos.min(0).astype(np.float64), pos.max(0).astype(np.float64)
        diag = float(np.linalg.norm(hi - lo))
        if args.camera:  # the reference's camera signature (CameraControls.cpp:342-399)
            c = nt.camera_decode(args.camera)
            eye = np.array(c["position"], dtype=np.float64)
            cam = dict(eye=tuple(eye), target=tuple(eye + np.array(c["forward"], dtype=np.float64)), up=tuple(c["up"]),
                       fov_deg=c["fov"], far=c["far"])
        else:  # inside the bounding box, looking down its longest axis
            ax = int(np.argmax(hi - lo))
            eye = 0.5 * (lo + hi)
            eye[ax] = lo[ax] + 0.15 * (hi[ax] - lo[ax])
            tgt = eye.copy()
            tgt[ax] = hi[ax]
            cam = dict(eye=tuple(eye), target=tuple(tgt), up=(0.0, 1.0, 0.0) if ax != 1 else (0.0, 0.0, 1.0), fov_deg=60.0,
                       far=3.0 * diag)
    return tri, pos, dict(cam), scene_name


class Frame:
    """The ray batches one rank traces of one frame (ntrace_amd.dist.FramePlan), resident in HBM: its slice of the primary
    batch and the AO batches generated on the device from its own primary hits."""

    def __init__(self, nt, torch, view, make_plan, cam, w, h, tri_normals, args, dev, stream, scenes):
        i32 = torch.int32
        n = w * h
        self.n_primary = n
        d_tab = self.d_tab = torch.zeros(n, dtype=i32, device=dev)
        nt.pixel_table(w, h, d_tab.data_ptr(), 0, stream)
        self.w, self.h, self.scenes, self.view, self.args, self.tri_normals = w, h, scenes, view, args, tri_normals
        self.d_rays = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
        self.d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        self.d_i2s = torch.zeros(n, dtype=i32, device=dev)
        self.d_s2i = torch.zeros(n, dtype=i32, device=dev)
        # the whole frame's primary rays are generated on every rank (untimed, App.cpp:955-969 excludes ray generation);
        # a rank traces only [lo, hi) of them
        nt.raygen_primary(self.d_rays.data_ptr(), self.d_i2s.data_ptr(), self.d_s2i.data_ptr(), d_tab.data_ptr(), cam["eye"],
                          scenes.nscreen_to_world(cam, w, h), w, h, cam["far"], 0, stream)
        plan = self.plan = make_plan(self.d_rays)   # the cut may depend on the rays (ranges of equal predicted cost)
        lo, hi = plan.lo, plan.hi
        self.batches = [dict(name="primary", n=hi - lo, any_hit=False, rays=self.d_rays.data_ptr() + lo * 32,
                             res=self.d_res.data_ptr() + lo * 16, live=hi - lo)]
        view.trace(args.kernel, hi - lo, False, self.batches[0]["rays"], self.batches[0]["res"], stream)
        self.own_hits = nt.count_hits(self.batches[0]["res"], hi - lo, stream) if hi > lo else 0
        self.keep = []
        ns = plan.samples
        ao_seed = 0xFFF2D5E4  # any fixed kernel seed; Raygen.random = false in config.conf
        for (first, cnt) in plan.ao_batches:
            b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
            b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
            b_a = torch.zeros(cnt * ns, dtype=i32, device=dev)
            nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), self.d_rays.data_ptr(), self.d_res.data_ptr(),
                         tri_normals.data_ptr(), first, cnt, ns, args.ao_radius, ao_seed, stream)
            live = nt.count_hits(self.d_res.data_ptr() + first * 16, cnt, stream) * ns
            self.keep.append((b_rays, b_res, b_a))
            self.batches.append(dict(name="ao", n=cnt * ns, any_hit=True, rays=b_rays.data_ptr(), res=b_res.data_ptr(), live=live,
                                     res_t=b_res, rays_t=b_rays, slots_t=b_a, first=first, count=cnt))
        # Dispatch hints of the secondary batches, made beside ray generation (untimed like it): a block's cost class from the depth in the
        # tree of the leaves its pixels' primary rays hit (ntr_bvh_leaf_depths once per BVH, ntr_secondary_block_costs per batch).
        self.nt, self.stream, self.ns = nt, stream, ns
        self.d_depth = None
        if not args.no_ao_prediction and args.kernel.startswith("fermi") and len(self.batches) > 1:
            self.num_tris = int(tri_normals.numel() * tri_normals.element_size() // 12)   # (three float32 per triangle)
            self.d_depth = torch.zeros(self.num_tris, dtype=i32, device=dev)
            nt.bvh_leaf_depths(view.d_nodes, view.nodes_bytes, view.d_woop, view.woop_bytes, view.d_tri_index, self.num_tris, self.d_depth.data_ptr(), stream)
            for b in self.batches[1:]:
                b["cost_t"] = torch.zeros((b["n"] + 255) // 256, dtype=i32, device=dev)
                b["hint"] = nt.SchedHint()
            self.predict_hints()
        torch.cuda.synchronize()

    def regenerate_primary(self, cam):
        """A new camera position: the frame's primary rays again, into the SAME buffers (what a renderer with a moving camera does)."""
        self.nt.raygen_primary(self.d_rays.data_ptr(), self.d_i2s.data_ptr(), self.d_s2i.data_ptr(), self.d_tab.data_ptr(), cam["eye"],
                               self.scenes.nscreen_to_world(cam, self.w, self.h), self.w, self.h, cam["far"], 0, self.stream)

    def regenerate_ao(self, count=True):
        """The AO batches again from the primary hit records as they are now, into the same buffers (asynchronous); with `count` the batches'
        live counts follow (a read-back per batch: a host synchronisation -- recount() does it after the frame's launches instead)."""
        for b in self.batches[1:]:
            a = b["slots_t"]
            self.nt.raygen_ao(b["rays"], a.data_ptr(), a.data_ptr(), self.d_rays.data_ptr(), self.d_res.data_ptr(), self.tri_normals.data_ptr(),
                              b["first"], b["count"], self.ns, self.args.ao_radius, 0xFFF2D5E4, self.stream)
        if count:
            self.recount()

    def recount(self):
        for b in self.batches[1:]:
            b["live"] = self.nt.count_hits(self.d_res.data_ptr() + b["first"] * 16, b["count"], self.stream) * self.ns
        self.batches[0]["live"] = self.plan.hi - self.plan.lo

    def predict_hints(self):
        """(Re)starts every secondary batch's hint from its predicted block costs -- what a renderer does when it generates the batch."""
        if self.d_depth is None:
            return
        for b in self.batches[1:]:
            self.nt.secondary_block_costs(self.d_res.data_ptr(), b["first"], b["count"], self.ns, self.d_depth.data_ptr(), self.num_tris,
                                          b["cost_t"].data_ptr(), self.stream)
            b["hint"].predict(b["cost_t"].data_ptr(), b["cost_t"].numel(), self.stream)

    @property
    def rays_per_step(self):  # the metric counts non-degenerate rays only (Renderer::getTotalNumRays, Renderer.cpp:676-709)
        return sum(b["live"] for b in self.batches)

    def own_primary_records(self):
        return self.d_res[self.plan.lo * 16: self.plan.hi * 16]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    import ntrace_amd as nt
    from ntrace_amd import dist as ntd
    from ntrace_amd import scenes

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the tracer has no CPU path)")
    if args.one_device:
        if world > 1 and args.dist_backend != "gloo":
            raise SystemExit("bench.py: --one-device needs --dist-backend gloo (RCCL refuses two ranks on one GPU)")
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # NTR_BENCH_FORCE_DIST=1 runs the RCCL code path (init, barrier, all-reduce, broadcast, gather) at world size 1 too
    use_dist = world > 1 or os.environ.get("NTR_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if world == 1 and "RANK" not in os.environ:   # NTR_BENCH_FORCE_DIST=1 without a launcher: a one-rank RCCL group of our own
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(sk.getsockname()[1]), "RANK": "0", "WORLD_SIZE": "1",
                               "LOCAL_RANK": "0"})
            sk.close()
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    nt.lib()
    stream = torch.cuda.current_stream().cuda_stream
    # The library's OWN multi-GPU entry points (ntr_dist_*, csrc/ntr_dist.cpp: RCCL bound by the C-ABI -- what a C++ host of the reference's
    # shape uses) carry the BVH broadcast and the frame's one collective whenever the ranks run on RCCL; torch.distributed
    # then only hands out the 128-byte group id, the barriers and the scalar reductions of the report.  gloo runs (CPU tier, --one-device)
    # keep the host-staged torch.distributed path: RCCL refuses two ranks on one GPU.
    grp, native_note = None, None
    if use_dist and args.dist_backend == "nccl" and os.environ.get("NTR_BENCH_TORCH_GATHER") != "1":
        # (a rank whose native group cannot be made -- RCCL not loadable by the C-ABI, say -- must not leave the others inside a collective:
        # every rank reports, and unless ALL succeeded the run uses the torch.distributed collectives and says so in the line)
        ok, err = 1, ""
        try:
            uid = torch.frombuffer(bytearray(nt.DistGroup.unique_id() if rank == 0 else bytes(128)), dtype=torch.uint8).to(dev)
        except Exception as e:
            uid, ok, err = torch.zeros(128, dtype=torch.uint8, device=dev), 0, repr(e)
        ntd.broadcast_(uid, 0)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        ntd.all_reduce_(flag, dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            try:
                grp = nt.DistGroup(bytes(uid.cpu().numpy().tobytes()), rank, world)
            except Exception as e:
                grp, ok, err = None, 0, repr(e)
            flag = torch.tensor([1 if grp is not None else 0], dtype=torch.int32, device=dev)
            ntd.all_reduce_(flag, dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and grp is not None:
                grp.close()
                grp = None
        if grp is None:
            native_note = "the library's RCCL group could not be made on every rank (%s): torch.distributed collectives used instead" % (err or "another rank failed")

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- scene + prebuilt BVH (host SAH build, Renderer.builder = SAHBVH, leaf prefs (1,1)): built ONCE on rank 0,
    # replicated by broadcast ------------------------------------------------------------------------------
    tri, pos, cam, scene_name = scene_and_camera(args, nt, scenes)
    bvh, sah_seconds = None, None
    if rank == 0:
        t0 = time.time()
        bvh = nt.sah_build(tri, pos, 1, 1)
        sah_seconds = time.time() - t0
    parts = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()) if bvh is not None else None
             for a in ((bvh.nodes, bvh.woop, bvh.tri_index) if bvh is not None else (None, None, None))]
    # The three Compact buffers live in ONE device allocation (256-byte aligned slices): the trace kernels' flat fetch addresses nodes and
    # triangles from one scalar base with 32-bit lane offsets, which needs both inside one 4 GiB window -- two separate allocations usually
    # are, but need not be (under rocprofv3 they were not, and the launches fell back to the two-descriptor fetch: INTEGRATION.md 3)
    sizes = torch.tensor([p_.numel() for p_ in parts] if rank == 0 else [0, 0, 0], dtype=torch.int64, device=dev)
    if use_dist:
        ntd.broadcast_(sizes, 0)   # (24 bytes over torch.distributed)
    sz = [int(x) for x in sizes.tolist()]
    offs = [0, (sz[0] + 255) // 256 * 256, (sz[0] + 255) // 256 * 256 + (sz[1] + 255) // 256 * 256]
    d_bvh = torch.zeros(offs[2] + sz[2], dtype=torch.uint8, device=dev)
    d_nodes, d_woop, d_idx = [d_bvh[offs[i]: offs[i] + sz[i]] for i in range(3)]
    if rank == 0:
        for t_, p_ in zip((d_nodes, d_woop, d_idx), parts):
            t_.copy_(p_)
    if grp is not None:   # ntr_dist_broadcast_bvh: the three buffers from the root
        grp.broadcast_bvh(d_nodes.data_ptr(), d_nodes.numel(), d_woop.data_ptr(), d_woop.numel(), d_idx.data_ptr(), d_idx.numel(), 0, stream)
        torch.cuda.synchronize()
    elif use_dist:
        ntd.broadcast_(d_bvh, 0)
    view = nt.BvhView(d_nodes.data_ptr(), d_nodes.numel(), d_woop.data_ptr(), d_woop.numel(), d_idx.data_ptr())
    view.validate(stream)
    d_nrm = up(scenes.tri_normals(tri, pos))

    w, h = args.width, args.height
    n_primary = w * h
    ns = args.ao_samples
    cut_info = {"mode": "single range"}
    if args.scaling == "weak" and world > 1:
        # round-1 mode: rank r renders its own frame of a slow camera move (2 units per rank in a 3 600-unit hall keeps the
        # frames distinct but equally costly)
        eye = np.array(cam["eye"], dtype=np.float64)
        eye[2] += 2.0 * rank
        cam["eye"] = tuple(eye)

        def make_plan(d_rays):
            return ntd.FramePlan(n_primary, 0, 1, ns, args.ao_batch_rays)
    else:
        def make_plan(d_rays):
            # strong scaling: N contiguous PixelTable ranges of equal PREDICTED cost (ntr_predict_block_costs on rank 0, one small
            # launch outside the timed region; cut points broadcast), instead of equal ray counts
            cuts = None
            if world > 1 or use_dist:
                if args.balance == "predicted":
                    if rank == 0:
                        d_cost = torch.zeros((n_primary + 255) // 256, dtype=torch.int32, device=dev)
                        nt.predict_block_costs(n_primary, d_rays.data_ptr(), d_nodes.data_ptr(), d_nodes.numel(), d_cost.data_ptr(), stream)
                        torch.cuda.synchronize()
                        cuts = ntd.balanced_cuts(d_cost.cpu().numpy(), n_primary, world, args.flat_share)
                    cuts = ntd.broadcast_cuts(cuts, world, dev)
                    cut_info.update({"mode": "equal predicted cost (ntr_predict_block_costs, flat share %g)" % args.flat_share, "cuts": cuts})
                else:
                    cut_info.update({"mode": "equal ray counts"})
            return ntd.FramePlan(n_primary, rank, world, ns, args.ao_batch_rays, cuts=cuts)
    frame = Frame(nt, torch, view, make_plan, cam, w, h, d_nrm, args, dev, stream, scenes)
    plan = frame.plan
    batches = frame.batches

    def run_batch(b, timed=False, s=stream):
        return view.trace(args.kernel, b["n"], b["any_hit"], b["rays"], b["res"], s, timed, hint=b.get("hint"))

    for _ in range(args.warmup):
        for b in batches:
            run_batch(b)
    barrier()

    # ---- timed region: exactly K steps; every batch is one launch on one stream between two HIP events ------------------
    E = torch.cuda.Event
    ev = [[(E(enable_timing=True), E(enable_timing=True)) for _ in batches] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        for bi, b in enumerate(batches):
            ev[s][bi][0].record()
            run_batch(b)
            ev[s][bi][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    nt.trace_status(stream)  # asynchronous launches report a traversal-stack overflow here (raises)
    kern_ms = np.array([[e0.elapsed_time(e1) for (e0, e1) in step] for step in ev])  # [steps, batches]
    kernel_seconds = float(kern_ms.sum()) * 1e-3                                       # this rank, all K steps
    rays_per_step = frame.rays_per_step
    total_rays_per_step, kernel_seconds_max = ntd.job_throughput(rays_per_step, kernel_seconds, dev)  # SUM rays, MAX time
    _, elapsed_max = ntd.job_throughput(0, elapsed, dev)
    prim_live_total, prim_kernel_max = ntd.job_throughput(batches[0]["live"], float(kern_ms[:, 0].sum()) * 1e-3, dev)
    ao_live_total, ao_kernel_max = ntd.job_throughput(sum(b["live"] for b in batches[1:]), float(kern_ms[:, 1:].sum()) * 1e-3, dev)

    # ---- the same K steps with the library's scheduling feedback off (NTR_TRACE_AUTO_HINT=0): every launch dispatched as if its batch
    # had never been traced before (predicted order for the primary batch, buffer order for the AO batches).  `value` is the reference's
    # protocol, which re-traces the SAME batches step after step, so its launches run in the order learned from the previous step;
    # this is the figure for batches that are new every time.  Untimed by the driver contract (reported beside `value`).
    cold = None
    if not args.no_cold_order:
        nt.set_tunables(NTR_TRACE_AUTO_HINT=0)
        for b in batches:
            run_batch(b)
        evc = [[(E(enable_timing=True), E(enable_timing=True)) for _ in batches] for _ in range(args.steps)]
        barrier()
        for s_ in range(args.steps):
            frame.predict_hints()   # every step as if the batches were new: the AO hints start over from their prediction (untimed, like ray generation)
            for bi, b in enumerate(batches):
                evc[s_][bi][0].record()
                run_batch(b)
                evc[s_][bi][1].record()
        barrier()
        cms = np.array([[e0.elapsed_time(e1) for (e0, e1) in step] for step in evc])
        _, cold_kernel_max = ntd.job_throughput(0, float(cms.sum()) * 1e-3, dev)
        cold = {"what": "the same steps with nothing learned from earlier launches of a batch (NTR_TRACE_AUTO_HINT=0; the AO batches' hints restarted "
                        "from their leaf-depth prediction before every step%s)" % ("" if frame.d_depth is not None else ": --no-ao-prediction, buffer order"),
                "mrays": total_rays_per_step * args.steps / cold_kernel_max / 1e6,
                "primary_ms": float(cms[:, 0].mean()), "ao_total_ms": float(cms[:, 1:].sum(axis=1).mean()) if len(batches) > 1 else 0.0}
        nt.set_tunables(NTR_TRACE_AUTO_HINT=None)

    # ---- final framebuffer gather (hit records of the primary batch -> rank 0) over RCCL, timed separately ---------------
    gather_ms, frame_check, native_gather = None, None, None
    sharded = use_dist and not (args.scaling == "weak" and world > 1)
    if use_dist:
        full = torch.zeros(n_primary * 16, dtype=torch.uint8, device=dev) if (sharded and grp is not None and rank == 0) else None
        barrier()
        g0 = time.perf_counter()
        if sharded and grp is not None:
            # the frame's ONE collective through the library: grouped ncclSend / ncclRecv of the ranks' 16-byte hit records to rank 0, at the
            # ranges' own offsets (equal ranges: ntr_dist_gather_records; ranges of equal predicted cost: the cut table)
            own = frame.own_primary_records()
            if plan.cuts is not None:
                grp.gather_records_cuts(own.data_ptr(), plan.cuts, full.data_ptr() if rank == 0 else 0, 0, stream)
            else:
                grp.gather_records(own.data_ptr(), n_primary, full.data_ptr() if rank == 0 else 0, 0, stream)
            native_gather = {"how": "ntr_dist_gather_records%s (csrc/ntr_dist.cpp): grouped ncclSend / ncclRecv of the ranks' 16-byte hit records to rank 0"
                                    % ("_cuts" if plan.cuts is not None else ""), "bvh_broadcast": "ntr_dist_broadcast_bvh"}
        elif sharded:
            full = ntd.gather_hit_records(frame.own_primary_records(), n_primary, cuts=plan.cuts)
        else:
            outs = [torch.empty_like(frame.d_res) for _ in range(world)] if rank == 0 else None
            ntd.gather_(frame.d_res, outs, 0)
            full = None
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if native_gather is not None:
            native_gather["ms"] = gather_ms
        if sharded:
            # checksum of checksums over every AO record of the frame (wrapping 64-bit sums)
            ao_sum = ntd.all_sum_int64(ntd.wrap_i64(sum(ntd.records_checksum(b["res_t"]) for b in batches[1:])), dev)
            if rank == 0:
                # rank 0 re-traces the WHOLE frame alone, with every rank's batch partition, and compares: primary records bit
                # for bit against the gathered frame, AO records through the checksum of checksums
                ref_res = torch.zeros_like(frame.d_res)
                view.trace(args.kernel, n_primary, False, frame.d_rays.data_ptr(), ref_res.data_ptr(), stream)
                prim_equal = bool(torch.equal(ref_res, full))
                ref_ao = 0
                for r in range(world):
                    pl = ntd.FramePlan(n_primary, r, world, ns, args.ao_batch_rays, cuts=plan.cuts)
                    for (first, cnt) in pl.ao_batches:
                        t_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
                        t_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
                        t_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
                        nt.raygen_ao(t_rays.data_ptr(), t_a.data_ptr(), t_a.data_ptr(), frame.d_rays.data_ptr(), ref_res.data_ptr(),
                                     d_nrm.data_ptr(), first, cnt, ns, args.ao_radius, 0xFFF2D5E4, stream)
                        view.trace(args.kernel, cnt * ns, True, t_rays.data_ptr(), t_res.data_ptr(), stream)
                        ref_ao += ntd.records_checksum(t_res)
                ao_equal = ntd.wrap_i64(ref_ao) == ntd.wrap_i64(ao_sum)
                frame_check = {"primary_records_equal_single_gpu_frame": prim_equal, "ao_checksum_equal_single_gpu_frame": ao_equal,
                               "records_compared": n_primary, "ao_records_checksummed": int(n_primary * ns)}
                if not (prim_equal and ao_equal):
                    raise SystemExit("bench.py: the assembled sharded frame differs from the single-GPU frame: %s" % frame_check)

    # ---- algorithmic bytes of the dominant kernel (instrumented trace, untimed) -------------------------------------------
    b0 = batches[0]
    st = view.trace_stats(args.kernel, b0["n"], False, b0["rays"], b0["res"], stream)
    alg_bytes = st.algorithmic_bytes()
    prim_ms = float(kern_ms[:, 0].mean())
    ao_ms = float(kern_ms[:, 1:].sum(axis=1).mean()) if len(batches) > 1 else 0.0
    achieved = alg_bytes / (prim_ms * 1e-3) / 1e9
    ao_alg, ao_visits = 0, 0
    for b in batches[1:]:
        st_b = view.trace_stats(args.kernel, b["n"], True, b["rays"], b["res"], stream)
        ao_alg += st_b.algorithmic_bytes()
        ao_visits += st_b.numInnerVisits + st_b.numTriTests
    ao_live = sum(b["live"] for b in batches[1:])

    # ---- N > 1: the mode in which this design scales by construction, in the SAME driver-parsed line: every rank traces a whole frame of
    # its own camera (no ray or record crosses a rank boundary), the same K steps
    frame_per_rank = None
    if (world > 1 or use_dist) and args.scaling == "strong":   # (also at world size 1 under NTR_BENCH_FORCE_DIST=1: the GPU test tier runs this code)
        cam_r = dict(cam)
        eye = np.array(cam_r["eye"], dtype=np.float64)
        eye[2] += 2.0 * rank        # a slow camera move: the frames are distinct and about equally costly
        cam_r["eye"] = tuple(eye)
        frame_w = Frame(nt, torch, view, lambda d_rays: ntd.FramePlan(n_primary, 0, 1, ns, args.ao_batch_rays), cam_r, w, h, d_nrm, args, dev, stream, scenes)
        for _ in range(args.warmup):
            for b in frame_w.batches:
                run_batch(b)
        evw = [[(E(enable_timing=True), E(enable_timing=True)) for _ in frame_w.batches] for _ in range(args.steps)]
        barrier()
        for s_ in range(args.steps):
            for bi, b in enumerate(frame_w.batches):
                evw[s_][bi][0].record()
                run_batch(b)
                evw[s_][bi][1].record()
        barrier()
        wms = np.array([[e0.elapsed_time(e1) for (e0, e1) in step] for step in evw])
        frame_per_rank = ntd.frame_per_rank_summary(frame_w.rays_per_step, float(wms.sum()) * 1e-3, args.steps, dev)
        del frame_w, evw

    overlapped = None
    if world > 1 and not args.no_extras:
        # every rank's frame share on three streams (no graph: one capture per rank is not worth its set-up here), between barriers;
        # the job's overlapped rate = rays of all ranks / MAX over ranks of the wall time per frame
        barrier()
        ov_s, how, _ = overlapped_frame(args, nt, torch, view, frame, dev, stream, try_graph=False)
        barrier()
        tot, ov_max = ntd.job_throughput(rays_per_step, ov_s, dev)
        overlapped = {"how": how + ", per rank, MAX over ranks", "ms_per_frame": ov_max * 1e3, "mrays_wall": tot / ov_max / 1e6}
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            extras = run_extras(args, nt, torch, scenes, view, frame, tri, pos, dev, stream, up, HBM_PEAK_GBS, cam)
        except Exception as e:  # extras never invalidate the headline
            extras = {"error": repr(e)}

    if frame_per_rank is not None and isinstance(extras, dict):
        extras["frame_per_rank"] = frame_per_rank
    if rank != 0:
        if grp is not None:
            grp.close()
        if use_dist:
            dist.destroy_process_group()
        return

    value = total_rays_per_step * args.steps / kernel_seconds_max / 1e6
    # binding roofs of the primary launch from the committed PMC passes of the SAME kernel symbol and launch shape (the counters
    # are fixed by the binary and the rays; the duration is this run's)
    wide = bool(view.flags & nt.BVH_WIDE_LEAVES)
    symbol = launched_symbol(args.kernel, wide)
    pmc, pmc_src = load_pmc(symbol, launched_grid(args.kernel, b0["n"]))
    visits = st.numInnerVisits + st.numTriTests   # lane steps of the unified-step loop (a terminator arrives with its triangle)
    # (lanes per wave-iteration is only reported for the persistent kernels: the per-ray kernels' uniform prologue fetches through the
    # scalar cache, so their vector-load count no longer counts iterations)
    binding = profiled_shares(pmc, pmc_src, visits if "persistent" in symbol else None)
    traffic = binding.get("hbm_traffic_bytes") if binding else None
    par = ("one frame sharded by screen tile over %d ranks (PixelTable ranges), BVH built on rank 0 and broadcast, %s gather of hit records"
           % (world, ("RCCL (the library's ntr_dist_* entry points)" if grp is not None else "RCCL (torch.distributed)") if args.dist_backend == "nccl" else "gloo (host-staged)")
           if not (args.scaling == "weak" and world > 1) else "weak scaling: one full frame per rank (camera shifted per rank), BVH replicated")
    out = {
        "metric": "Mrays/sec (primary + 8xAO) on Crytek Sponza",
        "value": value,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if (args.scaling == "weak" and world > 1) else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": ("synthetic (%s)" % scene_name) if not args.scene_obj else scene_name,
        "value_definition": "non-degenerate rays of all ranks / (sum of per-batch kernel times by HIP events, MAX over ranks): the "
                            "reference's protocol (App.cpp:955-969), which re-traces the same batches every step -- the library dispatches a "
                            "re-traced batch in the order its previous launch measured (cold_dispatch_order: the figure without that); "
                            "wall_mrays is the same rays / wall clock of the K steps",
        "wall_mrays": total_rays_per_step * args.steps / elapsed_max / 1e6,
        "config": {"workload": "Sponza-262k prebuilt SAH BVH, %dx%d primary + %dxAO (radius %g), one frame per step" % (w, h, ns, args.ao_radius),
                   "kernel": args.kernel, "bvh_flags": view.flags, "triangles": int(tri.shape[0]),
                   "rays_per_step": int(total_rays_per_step), "rays_per_step_rank0": rays_per_step,
                   "primary_rays_rank0": b0["n"], "primary_hits_rank0": frame.own_hits, "ao_rays_nondegenerate_rank0": ao_live,
                   "ao_batches_rank0": len(batches) - 1, "parallelism": par, "frame_cut": cut_info,
                   "dist_backend": (args.dist_backend if use_dist else None), "one_device": bool(args.one_device and world > 1),
                   **({"one_device_note": "all %d ranks share cuda:0 (gloo, host-staged collectives): the sharded frame -> gather -> bit-compare flow executed "
                                          "with real kernels; the ranks' launches contend for one chip, so value / ms_per_step are NOT a scaling measurement" % world}
                      if (args.one_device and world > 1) else {}),
                   "ao_dispatch_hint": ("leaf-depth prediction made beside ray generation (ntr_bvh_leaf_depths, ntr_secondary_block_costs, ntr_sched_hint_predict), "
                                        "refined by every launch's measurement (ntr_trace_bvh_hinted)") if frame.d_depth is not None else "none (library's automatic feedback)"},
        "primary_mrays": prim_live_total * args.steps / prim_kernel_max / 1e6,
        "ao_mrays": (ao_live_total * args.steps / ao_kernel_max / 1e6) if ao_kernel_max > 0 else None,
        "kernel_ms": {"primary": prim_ms, "ao_total": ao_ms, "per_step_rank0": float(kern_ms.sum(axis=1).mean()),
                      "per_batch": [round(float(x), 4) for x in kern_ms.mean(axis=0)],
                      # (a single stalled launch -- a box's management agent sampling the GPU, say -- weighs on a mean over a few hundred steps:
                      # 20 ms in one launch are 0.08 ms per step; the medians and the worst launch per batch make such a run readable)
                      "per_batch_median": [round(float(x), 4) for x in np.median(kern_ms, axis=0)],
                      "per_batch_max": [round(float(x), 4) for x in kern_ms.max(axis=0)],
                      "per_step_rank0_median": float(np.median(kern_ms.sum(axis=1)))},
        "value_median_step_rank0": (rays_per_step / (float(np.median(kern_ms.sum(axis=1))) * 1e-3) / 1e6) if world == 1 else None,
        "gather_ms": gather_ms,
        "gather_native": native_gather if native_gather is not None else ({"fallback": native_note} if native_note else None),
        "sharded_frame_check": frame_check,
        "host_sah_build_s": sah_seconds,
        "trace_stats": st.as_dict(),
        "value_cold": cold["mrays"] if cold else None,
        "value_moving_camera": (extras.get("moving_camera") or {}).get("mrays") if isinstance(extras, dict) else None,
        "value_variants_note": "`value` re-traces the same batches every step (the reference's protocol), so each launch runs in the dispatch order its "
                               "previous launch measured.  value_cold: the same steps with nothing learned from earlier launches (automatic feedback off, AO "
                               "hints restarted from their leaf-depth prediction); value_moving_camera: new rays in every launch (eye moves, rays "
                               "regenerated into the same buffers, order learned from the previous frame) -- what a renderer gets",
        "cold_dispatch_order": cold,
        "overlapped_multi_gpu": overlapped,
        "extras": extras,
        "roofline": None,
    }
    # Roofline.  Top level = the SURVEY 8(d) figure of the step's dominant kernel (the AO launches), bound "hbm": algorithmic bytes per
    # launch / mean launch duration / 8 TB/s.  The 34 MB BVH of the headline workload is cache-resident, so that fraction exceeds 1
    # (`cache_served`: true) and is not an efficiency; the same bytes against the paths that serve them (vector L1s, XCD L2s) are nested
    # under `cache_paths`, busy shares of TA / VALU from a committed PMC summary of the same symbol and grid under `utilisation` (null
    # with a note when none matches).  The other kernel of the step sits beside it (`primary`); the launches whose bytes HBM really
    # delivers are `roofline_hbm_resident` and extras.configs[4 | 5].roofline.
    cus = torch.cuda.get_device_properties(dev).multi_processor_count

    def launch_roofline(alg, sec, launches, sym, grid, label, binding_):
        """SURVEY 8(d) block of one kernel of the step: algorithmic bytes per launch / mean launch duration against HBM peak.  The 34 MB BVH of
        the headline workload never leaves the caches, so frac > 1: `cache_served`; the same bytes against the data paths that do serve
        them (vector L1s, XCD L2s) are nested under `cache_paths` -- shares of those paths, not the 8(d) figure."""
        ach = alg / sec / 1e9
        l1 = l1_roofline(alg, sec, cus)
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "cache_served": True,
                "traffic": (binding_ or {}).get("hbm_traffic_bytes"),
                "kernel": sym, "launch": label, "launches_per_step": launches,
                "algorithmic_bytes_per_launch": int(alg / launches), "launch_ms": sec * 1e3 / launches,
                "note": "SURVEY 8(d): algorithmic bytes (48 + 64 I + 48 T + 16 L + 4 H per ray, exact traversal counters) / launch time / 8 TB/s.  frac > 1: "
                        "L1 / L2 / Infinity Cache serve the 34 MB BVH (HBM-side bytes per launch in `traffic`, PMC); not an efficiency.  The launch whose "
                        "bytes HBM really delivers is `roofline_hbm_resident` / extras.configs[*].roofline",
                "cache_paths": {"l1_data_path": {"frac": l1["frac"], "peak": l1["peak"], "peak_definition": l1["peak_definition"]},
                                "l2_aggregate": {"frac": ach / L2_STREAM_PEAK_GBS, "peak": L2_STREAM_PEAK_GBS, "source": "MI355X_MICROARCH.md L2 section"},
                                "l2_gather": {"frac": ach / L2_GATHER_PEAK_GBS, "peak": L2_GATHER_PEAK_GBS, "source": "MI355X_MICROARCH.md gather table"},
                                "infinity_cache_gather": {"frac": ach / MALL_GATHER_PEAK_GBS, "peak": MALL_GATHER_PEAK_GBS}},
                "utilisation": binding_,
                "utilisation_note": None if binding_ else "no committed PMC summary under profiles/ matches the launched symbol %s with grid %d" % (sym, grid)}

    prim_roof = launch_roofline(alg_bytes, prim_ms * 1e-3, 1, "%s (%s)" % (symbol, args.kernel), launched_grid(args.kernel, b0["n"]),
                                "the primary batch of rank 0 (closest hit); the HIP events around the library call include the launch's dispatch-order "
                                "work (prediction or cost feedback, a few small kernels)", binding)
    roof = prim_roof
    if ao_ms > 0:
        ao_sym = launched_symbol(args.kernel, wide, any_hit=True)
        ao_pmc, ao_src = load_pmc(ao_sym, launched_grid(args.kernel, batches[1]["n"]))
        ao_bind = profiled_shares(ao_pmc, ao_src, (ao_visits // max(len(batches) - 1, 1)) if "persistent" in ao_sym else None) if ao_pmc else None
        ao_roof = launch_roofline(ao_alg, ao_ms * 1e-3, len(batches) - 1, "%s (%s)" % (ao_sym, args.kernel), launched_grid(args.kernel, batches[1]["n"]),
                                  "the %d AO batches of rank 0 (any hit)" % (len(batches) - 1), ao_bind)
        ao_roof["share_of_step"] = ao_ms / (ao_ms + prim_ms)
        prim_roof["share_of_step"] = prim_ms / (ao_ms + prim_ms)
        # top level = the step's DOMINANT kernel (the AO launches: three quarters of the step), the other one beside it
        if ao_ms >= prim_ms:
            roof = dict(ao_roof, primary=prim_roof)
        else:
            roof = dict(prim_roof, ao=ao_roof)
    out["roofline"] = roof
    hp = extras.get("hbm_resident_point") if isinstance(extras, dict) else None
    if hp and hp.get("incoherent"):
        out["roofline_hbm_resident"] = hp["incoherent"]["roofline"]

    # ---- CPU baseline: the oracle (restated reference CPU tracer) on a bounded sample, rank 0 at N=1 only -----------------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        cores = os.cpu_count() or 1
        rays = frame.d_rays.cpu().numpy().view(nt.RAY_DTYPE)
        oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays[:20000], threads=cores)  # warm-up
        n1 = args.cpu_sample_rays or min(n_primary, 250_000)
        c0 = time.perf_counter()
        oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays[:n1], threads=1)
        c1 = time.perf_counter()
        run_batch(batches[0])
        torch.cuda.synchronize()
        # the whole frame (every batch the GPU traced in one step), all host cores, compared record by record
        cpu_s, mism, traced = 0.0, 0, 0
        for b in batches:
            hr = (frame.d_rays if b["name"] == "primary" else b["rays_t"]).cpu().numpy().view(nt.RAY_DTYPE)
            t0c = time.perf_counter()
            ref, _ = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, hr, any_hit=b["any_hit"], threads=cores)
            cpu_s += time.perf_counter() - t0c
            got = (frame.d_res if b["name"] == "primary" else b["res_t"]).cpu().numpy().view(nt.RESULT_DTYPE)
            mism += int(((got["id"] != ref["id"]) | (got["t"].view(np.uint32) != ref["t"].view(np.uint32))).sum())
            traced += b["n"]
        out["cpu_baseline"] = {"value": rays_per_step / cpu_s / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                               "sample": "one whole step (primary + %d AO batches = %d rays, %d non-degenerate) on all %d host "
                                         "cores, %.2f s wall = %.1f CPU-s; 1 thread on the first %d primary rays: %.3f Mrays/s"
                                         % (len(batches) - 1, traced, rays_per_step, cores, cpu_s, cpu_s * cores, n1,
                                            n1 / (c1 - c0) / 1e6),
                               "single_thread_mrays": n1 / (c1 - c0) / 1e6,
                               "parity_mismatches_whole_step": mism, "rays_compared": traced}
    if grp is not None:
        grp.close()
    if use_dist:
        dist.destroy_process_group()
    # RCCL writes a version banner to the C stdout stream; push it out first so that the JSON line is the last line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


def device_lbvh(nt, torch, up, dev, stream, tri_, pos_, reps, hbm_peak):
    """On-device LBVH build (leafSize 8, epsilon 0.001) of a scene, best of `reps`; returns (view, result, info, buffers to keep alive)."""
    capn, capw, capi = nt.lbvh_capacity(tri_.shape[0])
    d_tri, d_pos = up(tri_), up(pos_)
    ln = torch.zeros(capn, dtype=torch.uint8, device=dev)
    lw = torch.zeros(capw, dtype=torch.uint8, device=dev)
    li = torch.zeros(capi, dtype=torch.uint8, device=dev)
    best = None
    for _ in range(reps):
        r = nt.lbvh_build(tri_.shape[0], d_tri.data_ptr(), pos_.shape[0], d_pos.data_ptr(), pos_.min(0), pos_.max(0), 8, 0.001,
                          ln.data_ptr(), capn, lw.data_ptr(), capw, li.data_ptr(), capi, stream)
        best = r if best is None or r.seconds < best.seconds else best
    lview = nt.BvhView(ln.data_ptr(), best.nodesBytes, lw.data_ptr(), best.triWoopBytes, li.data_ptr())
    lview.validate(stream)
    # algorithmic bytes of the build (SURVEY 8d accounting, exact from the counts): Morton 48 rd + 8 wr, 4 radix passes x
    # (8 rd + 8 wr + 4 rd histogram), Woop 48 rd + 48 wr per triangle; emit 12 rd + 16 wr per inner node, (48+4) rd +
    # (48+12) wr per triangle, 20 wr per leaf; refit (12+36) rd per triangle, 96 rd per inner child, 48 wr per node.
    nt_, ni_, nl_ = int(tri_.shape[0]), int(best.numNodes), int(best.numLeaves)
    lb = nt_ * (56 + 4 * 20 + 96 + 112 + 48) + ni_ * (28 + 48) + nl_ * 20 + max(ni_ - 1, 0) * 96
    info = {"triangles": nt_, "build_ms": best.seconds * 1e3, "mtris_per_s": nt_ / best.seconds / 1e6,
            "phases_ms": {"morton+digit_histograms": best.mortonMs, "sort_4_onesweep_passes": best.sortMs,
                          "leaf_marks+scan": best.emitMs,
                          "bottom_up_emit(nodes,boxes,woop_rows: agglomerate,runs)": best.refitMs},
            "nodes": ni_, "leaves": nl_, "algorithmic_bytes": lb,
            "roofline": {"bound": "hbm", "achieved": lb / best.seconds / 1e9, "peak": hbm_peak, "unit": "GB/s",
                         "frac": lb / best.seconds / 1e9 / hbm_peak}}
    return lview, best, info, (ln, lw, li, d_tri, d_pos)


def config_point(nt, torch, scenes, dev, stream, up, name, tri_, pos_, cam_, view_, bvh_bytes, secondary, kernels, w, h, ns, hbm_peak, pmc_tag):
    """One BASELINE.json configuration, driver-timed: the frame's batches -- 1080p primary + the 8 x AO (any hit, aoRadius scaled to the
    scene) or 8 x diffuse (closest hit, to the camera's far plane: Renderer.cpp:533-537) batches of <= 2^20 rays (Renderer.cpp:45) --
    traced by every name of `kernels` (the first one is the default selector) in the reference's protocol: per batch one untimed launch,
    then the median of three timed ones (App.cpp:955-958), Sigma of the per-batch kernel times.  Rooflines per SURVEY 8(d): algorithmic bytes
    from the exact traversal counters (ntr_trace_bvh_stats) / time / 8 TB/s; `traffic` = HBM bytes per frame from a committed PMC summary
    (profiles/*<pmc_tag>*_pmc_summary.json) of the same kernel symbol when there is one."""
    i32 = torch.int32
    rays, _ = scenes.primary_rays(cam_, w, h)
    npr = rays.shape[0]
    d_rays = up(rays)
    d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    d_nrm = up(scenes.tri_normals(tri_, pos_))
    diag = float(np.linalg.norm(pos_.max(0).astype(np.float64) - pos_.min(0)))
    any_hit = secondary == "ao"
    dist_ = 5.0 * diag / 4300.0 if any_hit else cam_["far"]   # config.conf's aoRadius 5 is in Sponza units: scaled by the scene's diagonal
    per = (1 << 20) // ns
    cache = "infinity-cache" if bvh_bytes <= (256 << 20) else "hbm"
    wide = bool(view_.flags & nt.BVH_WIDE_LEAVES)
    b_rays = torch.zeros(per * ns * 32, dtype=torch.uint8, device=dev)
    b_res = torch.zeros(per * ns * 16, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(per * ns, dtype=i32, device=dev)
    # the reference sorts every secondary batch before it traces it (Renderer.sortRays defaults to true: config.conf:27, AppEnvironment.cpp:68;
    # Renderer.cpp:562 RayBuffer::mortonSort, outside the timed trace like ray generation): the default selector's batches are timed both ways
    s_rays = torch.zeros_like(b_rays)
    s_i2s = torch.zeros(per * ns, dtype=i32, device=dev)
    s_s2i = torch.zeros(per * ns, dtype=i32, device=dev)
    s_id = torch.arange(per * ns, dtype=i32, device=dev)

    def roof(alg, sec, sym, grid_rays, launches):
        r = {"bound": "hbm", "achieved": alg / sec / 1e9, "peak": hbm_peak, "unit": "GB/s", "frac": alg / sec / 1e9 / hbm_peak,
             "algorithmic_bytes": int(alg), "ms": sec * 1e3, "bvh_resident_in": cache, "cache_served": cache != "hbm", "kernel": sym}
        pmc, src = load_pmc(sym, grid_rays, tag=pmc_tag)
        if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            tb = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 * launches
            r.update({"traffic": tb, "traffic_over_algorithmic": tb / alg if alg else None, "traffic_source": "profiles/" + src,
                      "traffic_note": "per-dispatch mean of the symbol's launches in the profiled frame x %d launches; 2 x FETCH_SIZE assumes "
                                      "128-byte requests (the guide's calibration is for streaming reads; for a 64-byte gather it is an assumption)" % launches})
        else:
            r["traffic"] = None
        return r

    out = {"config": name, "triangles": int(tri_.shape[0]), "bvh_bytes": int(bvh_bytes), "secondary": "8 x %s" % secondary,
           "secondary_distance": dist_, "by_kernel": {}}
    for ki, kn in enumerate(kernels):
        view_.trace(kn, npr, False, d_rays.data_ptr(), d_res.data_ptr(), stream)
        tp = float(np.median([view_.trace(kn, npr, False, d_rays.data_ptr(), d_res.data_ptr(), stream) for _ in range(3)]))
        row = {"primary_ms": tp * 1e3, "primary_mrays": npr / tp / 1e6}
        if ki == 0:
            st = view_.trace_stats(kn, npr, False, d_rays.data_ptr(), d_res.data_ptr(), stream)
            hits = nt.count_hits(d_res.data_ptr(), npr, stream)
            out["primary_hit_rate"] = hits / npr
            row["primary_roofline"] = roof(st.algorithmic_bytes(), tp, launched_symbol(kn, wide), launched_grid(kn, npr), 1)
        tt, live, launched, alg, nb, per_batch = 0.0, 0, 0, 0, 0, []
        tt_sorted, sort_s, per_batch_sorted = 0.0, 0.0, []
        for lo in range(0, npr, per):
            cnt = min(per, npr - lo)
            nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), lo, cnt, ns, dist_,
                         0xFFF2D5E4, stream)
            view_.trace(kn, cnt * ns, any_hit, b_rays.data_ptr(), b_res.data_ptr(), stream)
            t_b = float(np.median([view_.trace(kn, cnt * ns, any_hit, b_rays.data_ptr(), b_res.data_ptr(), stream) for _ in range(3)]))
            tt += t_b
            per_batch.append(round(t_b * 1e3, 4))
            live += nt.count_hits(d_res.data_ptr() + lo * 16, cnt, stream) * ns
            launched += cnt * ns
            nb += 1
            if ki == 0:
                alg += view_.trace_stats(kn, cnt * ns, any_hit, b_rays.data_ptr(), b_res.data_ptr(), stream).algorithmic_bytes()
                sort_s += nt.ray_morton_sort(cnt * ns, b_rays.data_ptr(), s_id.data_ptr(), s_rays.data_ptr(), s_i2s.data_ptr(), s_s2i.data_ptr(), stream)
                view_.trace(kn, cnt * ns, any_hit, s_rays.data_ptr(), b_res.data_ptr(), stream)
                t_s = float(np.median([view_.trace(kn, cnt * ns, any_hit, s_rays.data_ptr(), b_res.data_ptr(), stream) for _ in range(3)]))
                tt_sorted += t_s
                per_batch_sorted.append(round(t_s * 1e3, 4))
        row.update({"secondary_ms": tt * 1e3, "secondary_batches": nb, "secondary_mrays": live / tt / 1e6 if tt > 0 else None,
                    "secondary_mrays_all_launched": launched / tt / 1e6 if tt > 0 else None, "secondary_per_batch_ms": per_batch,
                    "frame_ms": (tp + tt) * 1e3, "frame_mrays": (npr + live) / (tp + tt) / 1e6})
        if ki == 0:
            out["rays"] = {"primary": npr, "secondary_nondegenerate": int(live), "secondary_launched": int(launched)}
            sec_sym = launched_symbol(kn, wide, any_hit=any_hit, incoherent=not any_hit)   # (diffuse batches: routed to the persistent body)
            row["secondary_roofline"] = roof(alg, tt, sec_sym, launched_grid(kn, per * ns, symbol=sec_sym), nb)
            row.update({"secondary_sorted_ms": tt_sorted * 1e3, "secondary_sorted_mrays": live / tt_sorted / 1e6 if tt_sorted > 0 else None,
                        "secondary_sorted_per_batch_ms": per_batch_sorted, "ray_sort_ms_untimed": sort_s * 1e3,
                        "frame_sorted_ms": (tp + tt_sorted) * 1e3, "frame_sorted_mrays": (npr + live) / (tp + tt_sorted) / 1e6,
                        "sorted_note": "the same batches Morton-sorted first (ntr_ray_morton_sort: the reference's RayBuffer::mortonSort, its default -- "
                                       "Renderer.sortRays true --, outside the timed trace): identical rays, hence identical algorithmic bytes"})
            row["secondary_sorted_roofline"] = roof(alg, tt_sorted, sec_sym, launched_grid(kn, per * ns, symbol=sec_sym), nb)
        out["by_kernel"][kn] = row
    k0 = out["by_kernel"][kernels[0]]
    dom = "secondary" if k0["secondary_ms"] >= k0["primary_ms"] else "primary"
    out["roofline"] = dict(k0[dom + "_roofline"], launch="%s batches, %s, rays in generation order" % (dom, kernels[0]), share_of_frame=k0[dom + "_ms"] / k0["frame_ms"])
    if dom == "secondary":
        out["roofline_sorted_rays"] = dict(k0["secondary_sorted_roofline"], launch="secondary batches, %s, rays Morton-sorted first (the reference's default, "
                                           "Renderer.sortRays = true; the sort is untimed like ray generation)" % kernels[0],
                                           share_of_frame=k0["secondary_sorted_ms"] / k0["frame_sorted_ms"])
    return out


def overlapped_frame(args, nt, torch, view, frame, dev, stream, try_graph=True):
    """The frame's independent AO batches round-robin on a few HIP streams behind the primary batch, the whole frame captured once
    into a HIP graph and replayed (every kernel runs on every replay): what an application that is not bound to the reference's
    synchronous launches gets.  Not the protocol, so not `value`.  Returns (seconds per frame, how)."""
    batches = frame.batches
    E = torch.cuda.Event
    main_stream = torch.cuda.Stream(device=dev)
    ao_streams = [torch.cuda.Stream(device=dev) for _ in range(args.ao_streams)] if (args.ao_streams > 1 and len(batches) > 2) else []

    def run_step(ms):
        view.trace(args.kernel, batches[0]["n"], False, batches[0]["rays"], batches[0]["res"], ms.cuda_stream, False)
        p1 = E()
        p1.record(ms)
        if ao_streams:
            for st_ in ao_streams:
                st_.wait_event(p1)
            for i, b in enumerate(batches[1:]):
                view.trace(args.kernel, b["n"], True, b["rays"], b["res"], ao_streams[i % len(ao_streams)].cuda_stream, False)
            for st_ in ao_streams:
                e = E()
                e.record(st_)
                ms.wait_event(e)
        else:
            for b in batches[1:]:
                view.trace(args.kernel, b["n"], True, b["rays"], b["res"], ms.cuda_stream, False)

    for _ in range(3):
        run_step(main_stream)
    torch.cuda.synchronize()
    steps = max(5, min(args.steps, 20))
    note = "launch by launch on %d streams" % max(1, len(ao_streams))
    graph = None
    if try_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=main_stream, capture_error_mode="thread_local"):
                run_step(torch.cuda.current_stream())
            graph.replay()
            torch.cuda.synchronize()
            note = "HIP graph replay, AO batches on %d streams" % max(1, len(ao_streams))
        except Exception as e:
            graph = None
            note += " (graph capture failed: %s)" % type(e).__name__
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
    t0 = time.perf_counter()
    for _ in range(steps):
        if graph is not None:
            graph.replay()
        else:
            run_step(main_stream)
    torch.cuda.synchronize()
    ov = time.perf_counter() - t0
    nt.trace_status(stream)
    del graph
    return ov / steps, note, steps


def run_extras(args, nt, torch, scenes, view, frame, tri, pos, dev, stream, up, hbm_peak, frame_cam):
    """Measured outside the timed region, one GPU only: the frame overlapped on streams and replayed as a HIP graph, the
    opt-in scheduling hints, the device-to-device copy ceiling, the on-device LBVH build of the bench scene, the ray sort
    and the HBM-resident roofline point (10 M-triangle BVH)."""
    extras = {}
    batches = frame.batches
    n_primary = batches[0]["n"]
    rays_per_step = frame.rays_per_step
    i32 = torch.int32
    E = torch.cuda.Event

    # (1) the frame overlapped on streams and replayed as a HIP graph
    ov_s, note, steps = overlapped_frame(args, nt, torch, view, frame, dev, stream)
    extras["overlapped_frame"] = {"how": note, "ms_per_frame": ov_s * 1e3, "mrays_wall": rays_per_step / ov_s / 1e6, "steps": steps}

    # (1b) the reference's persistent-threads selectors on the SAME step (BASELINE config 3 names the persistent-threads traversal;
    # the per-ray kernel is what `value` runs): one warm step, then `psteps` timed steps, per-batch HIP events as for `value`
    pk = {}
    psteps = max(3, min(args.steps, 10))
    for kn in ("tesla_persistent_while_while", "kepler_dynamic_fetch"):
        for b in batches:
            view.trace(kn, b["n"], b["any_hit"], b["rays"], b["res"], stream, False)
        pev = [[(E(enable_timing=True), E(enable_timing=True)) for _ in batches] for _ in range(psteps)]
        for s_ in range(psteps):
            for bi, b in enumerate(batches):
                pev[s_][bi][0].record()
                view.trace(kn, b["n"], b["any_hit"], b["rays"], b["res"], stream, False)
                pev[s_][bi][1].record()
        torch.cuda.synchronize()
        pms = np.array([[e0.elapsed_time(e1) for (e0, e1) in st_] for st_ in pev])
        pk[kn] = {"mrays": rays_per_step / (float(pms.sum(axis=1).mean()) * 1e-3) / 1e6,
                  "primary_mrays": n_primary / (float(pms[:, 0].mean()) * 1e-3) / 1e6,
                  "ao_mrays": (sum(b["live"] for b in batches[1:]) / (float(pms[:, 1:].sum(axis=1).mean()) * 1e-3) / 1e6) if len(batches) > 1 else None,
                  "primary_ms": float(pms[:, 0].mean()), "steps": psteps}
    pk["what"] = ("the same step traced by the persistent-threads selectors (tesla_persistent_while_while: while-while loop, whole-wave refill from "
                  "128 pool heads; kepler_dynamic_fetch: unified step, ballot/mbcnt refill of finished lanes); identical hit records")
    extras["persistent_kernels"] = pk
    for b in batches:   # restore the per-ray kernel's records (identical bits, but keep one writer for what follows)
        view.trace(args.kernel, b["n"], b["any_hit"], b["rays"], b["res"], stream, False)
    torch.cuda.synchronize()

    # (2) opt-in scheduling hints (ntr_trace_bvh_hinted): block order learned from the previous trace of the same batch
    hints = [nt.SchedHint() for _ in batches]
    for _ in range(4):
        for b, hnt in zip(batches, hints):
            view.trace(args.kernel, b["n"], b["any_hit"], b["rays"], b["res"], stream, False, hint=hnt)
    torch.cuda.synchronize()
    hsteps = max(3, min(args.steps, 10))
    hev = [[(E(enable_timing=True), E(enable_timing=True)) for _ in batches] for _ in range(hsteps)]
    for s_ in range(hsteps):
        for bi, (b, hnt) in enumerate(zip(batches, hints)):
            hev[s_][bi][0].record()
            view.trace(args.kernel, b["n"], b["any_hit"], b["rays"], b["res"], stream, False, hint=hnt)
            hev[s_][bi][1].record()
    torch.cuda.synchronize()
    hms = np.array([[e0.elapsed_time(e1) for (e0, e1) in st_] for st_ in hev])
    extras["sched_hints"] = {
        "what": "same batches re-traced with ntr_trace_bvh_hinted (block order from the previous trace of the batch); identical hit "
                "records; helps repeated / static batches, not a moving camera (DESIGN.md 4.1)",
        "mrays": rays_per_step / (float(hms.sum(axis=1).mean()) * 1e-3) / 1e6,
        "primary_mrays": n_primary / (float(hms[:, 0].mean()) * 1e-3) / 1e6,
        "ao_mrays": (sum(b["live"] for b in batches[1:]) / (float(hms[:, 1:].sum(axis=1).mean()) * 1e-3) / 1e6) if len(batches) > 1 else None,
        "primary_ms": float(hms[:, 0].mean()), "steps": hsteps}
    for hnt in hints:
        hnt.close()

    # (2b) a moving camera: every step the eye moves on (0.25 units of a 3 600-unit hall: ~1 pixel at the far wall), the primary rays are
    # generated again into the same buffers and the AO batches again from the new hits -- every batch holds NEW rays, but the library's
    # automatic feedback (keyed by buffer) dispatches it in the order the PREVIOUS frame's launch of that buffer measured.  What a renderer
    # in motion gets: between `value` (the same rays again) and cold_dispatch_order (nothing known).  Ray generation untimed, as everywhere.
    try:
        msteps = max(3, min(args.steps, 10))
        cam0 = dict(frame_cam)
        mev = [[(E(enable_timing=True), E(enable_timing=True)) for _ in batches] for _ in range(msteps)]
        lib_ms = np.zeros((msteps, len(batches)))
        mrays = 0
        for s_ in range(-2, msteps):      # two untimed frames first: the hints of the moving sequence form
            camm = dict(cam0)
            eye = np.array(camm["eye"], dtype=np.float64)
            eye[2] += float(os.environ.get("NTR_BENCH_MOVE_STEP", "0.25")) * (s_ + 3)
            camm["eye"] = tuple(eye)
            frame.regenerate_primary(camm)
            b0_ = batches[0]
            lib_timed = os.environ.get("NTR_BENCH_MOVE_TIMED") == "1"   # diagnostic: the library's own per-launch seconds (its events exclude the feedback kernels)
            if s_ >= 0:
                mev[s_][0][0].record()
            t_ = view.trace(args.kernel, b0_["n"], False, b0_["rays"], b0_["res"], stream, lib_timed)
            if s_ >= 0:
                mev[s_][0][1].record()
                if lib_timed:
                    lib_ms[s_][0] = t_ * 1e3
            frame.regenerate_ao(count=False)      # (no read-back between ray generation and the launches: the GPU stays fed, as in the timed region of `value`)
            for bi, b in enumerate(batches[1:], start=1):
                if s_ >= 0:
                    mev[s_][bi][0].record()
                t_ = view.trace(args.kernel, b["n"], True, b["rays"], b["res"], stream, lib_timed, hint=b.get("hint"))
                if s_ >= 0:
                    mev[s_][bi][1].record()
                    if lib_timed:
                        lib_ms[s_][bi] = t_ * 1e3
            frame.recount()                       # the frame's non-degenerate ray count, after its launches
            if s_ >= 0:
                mrays += frame.rays_per_step
        torch.cuda.synchronize()
        mms = np.array([[e0.elapsed_time(e1) for (e0, e1) in st_] for st_ in mev])
        if os.environ.get("NTR_BENCH_MOVE_TIMED") == "1":
            mms = lib_ms
        extras["moving_camera"] = {
            "what": "the eye moves 0.25 units per step; primary rays and AO batches regenerated into the same buffers every step (untimed), traced in the "
                    "order the previous frame's launches measured (the library's automatic feedback / the batch's hint): new rays every launch",
            "mrays": mrays / (float(mms.sum()) * 1e-3) / 1e6, "primary_ms": float(mms[:, 0].mean()),
            "ao_total_ms": float(mms[:, 1:].sum(axis=1).mean()) if len(batches) > 1 else 0.0, "steps": msteps,
            "per_batch_ms": [round(float(x), 4) for x in mms.mean(axis=0)]}
        # back to the bench camera (the extras below and the cpu_baseline leg compare against these buffers)
        frame.regenerate_primary(cam0)
        view.trace(args.kernel, batches[0]["n"], False, batches[0]["rays"], batches[0]["res"], stream, False)
        frame.regenerate_ao()
        for b in batches[1:]:
            view.trace(args.kernel, b["n"], True, b["rays"], b["res"], stream, False)
        torch.cuda.synchronize()
    except Exception as e:
        extras["moving_camera"] = {"error": repr(e)}

    # (3) practical HBM ceiling: device-to-device copy of a buffer larger than the Infinity Cache
    cp_src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    cp_dst = torch.empty_like(cp_src)
    cp_dst.copy_(cp_src)
    e0, e1 = E(enable_timing=True), E(enable_timing=True)
    e0.record()
    for _ in range(4):
        cp_dst.copy_(cp_src)
    e1.record()
    torch.cuda.synchronize()
    extras["stream_copy_GBps"] = 2.0 * 4 * cp_src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del cp_src, cp_dst

    # (3b) the practical roof of an HBM-resident BVH's fetches: dependent chains of random 64-byte records (ntr_selftest_gather_rate)
    try:
        g_waves, g_steps = 8192, 256
        gsec = nt.selftest_gather_rate(768 << 20, g_waves, 64, g_steps, stream)
        gsec2 = nt.selftest_gather_rate(2 << 20, g_waves, 64, g_steps, stream)
        extras["gather_roof"] = {"what": "%d waves x 64 lanes, each a dependent chain of %d random 64-byte records (4 x 16-byte loads per record): the memory side of a "
                                         "divergent traversal without its arithmetic -- of a 768 MB table (every fetch beyond the L2: `grecords_per_s`) and of a "
                                         "2 MB table (every fetch an L2 hit: `l2_resident_grecords_per_s`)" % (g_waves, g_steps),
                                 "grecords_per_s": g_waves * 64 * g_steps / gsec / 1e9, "GBps": g_waves * 64 * g_steps * 64 / gsec / 1e9, "ms": gsec * 1e3,
                                 "l2_resident_grecords_per_s": g_waves * 64 * g_steps / gsec2 / 1e9}
    except Exception as e:
        extras["gather_roof"] = {"error": repr(e)}

    # (4) on-device LBVH build of the bench scene + primary rays on the built tree
    def lbvh_of(tri_, pos_, reps):
        return device_lbvh(nt, torch, up, dev, stream, tri_, pos_, reps, hbm_peak)

    lview, best, info, keep = lbvh_of(tri, pos, 3)
    tmp_res = torch.zeros(n_primary * 16, dtype=torch.uint8, device=dev)
    lsec = min(lview.trace(args.kernel, n_primary, False, batches[0]["rays"], tmp_res.data_ptr(), stream) for _ in range(5))
    info["primary_mrays_on_lbvh"] = n_primary / lsec / 1e6
    extras["lbvh"] = info
    del keep, lview

    # (5) secondary-ray sort
    if len(batches) > 1:
        b1 = batches[1]
        so = torch.zeros_like(b1["rays_t"])
        sa = torch.zeros(b1["n"], dtype=i32, device=dev)
        sb = torch.zeros(b1["n"], dtype=i32, device=dev)
        ident = torch.arange(b1["n"], dtype=i32, device=dev)
        ssec = min(nt.ray_morton_sort(b1["n"], b1["rays"], ident.data_ptr(), so.data_ptr(), sa.data_ptr(), sb.data_ptr(), stream) for _ in range(3))
        sres = torch.zeros_like(b1["res_t"])
        tsec = min(view.trace(args.kernel, b1["n"], True, so.data_ptr(), sres.data_ptr(), stream) for _ in range(5))
        usec = min(view.trace(args.kernel, b1["n"], True, b1["rays"], b1["res"], stream) for _ in range(5))
        extras["ray_sort"] = {"rays": b1["n"], "sort_ms": ssec * 1e3, "trace_sorted_ms": tsec * 1e3, "trace_unsorted_ms": usec * 1e3}

    # (5b) BASELINE.json configs 3 and 4 themselves, driver-timed, each with its roofline (config 5 follows with the 10 M-triangle scene)
    if not args.no_configs:
        cfgs = extras.setdefault("configs", {})
        cfgs["what"] = ("BASELINE.json configurations 3-5 on their seeded stand-ins, one GPU, the frame's batches in the reference's protocol (per batch one "
                        "untimed launch, then the median of three timed ones; rays of primary hits / sum of kernel times); by_kernel[<first>] is the default "
                        "selector; roofline = SURVEY 8(d) algorithmic bytes / time / 8 TB/s of the frame's dominant launches")
        try:
            tri3, pos3, cam3 = scenes.conference_room()
            t0 = time.time()
            bvh3 = nt.sah_build(tri3, pos3, 1, 1)
            sah3 = time.time() - t0
            k3 = [up(bvh3.nodes), up(bvh3.woop), up(bvh3.tri_index)]
            view3 = nt.BvhView(k3[0].data_ptr(), bvh3.nodes.nbytes, k3[1].data_ptr(), bvh3.woop.nbytes, k3[2].data_ptr())
            view3.validate(stream)
            cfgs["3"] = config_point(nt, torch, scenes, dev, stream, up, "3 Conference (room-331k stand-in), host SAH, primary + 8xAO", tri3, pos3, cam3, view3,
                                     bvh3.nodes.nbytes + bvh3.woop.nbytes + bvh3.tri_index.nbytes, "ao",
                                     (args.kernel, "tesla_persistent_while_while", "kepler_dynamic_fetch"), args.width, args.height, args.ao_samples, hbm_peak, "conference")
            cfgs["3"]["host_sah_build_s"] = sah3
            del k3, view3, bvh3
        except Exception as e:
            cfgs["3"] = {"error": repr(e)}
        try:
            tri4, pos4, cam4 = scenes.hairball()
            lview4, best4, info4, keep4 = lbvh_of(tri4, pos4, 3)
            cfgs["4"] = config_point(nt, torch, scenes, dev, stream, up, "4 Hairball (2.8 M stand-in), device LBVH build + 8x diffuse", tri4, pos4, cam4, lview4,
                                     best4.nodesBytes + best4.triWoopBytes + best4.triIndexBytes, "diffuse", (args.kernel, "kepler_dynamic_fetch"),
                                     args.width, args.height, args.ao_samples, hbm_peak, "hairball_diffuse_frame")
            cfgs["4"]["lbvh_build"] = info4
            # (f-3) where the secondary-ray Morton sort could pay: one diffuse batch of this frame (long incoherent rays in a deep tree), sorted
            # by ntr_ray_morton_sort (the reference's 192-bit key, RayBuffer.cpp:103-165; outside the timed trace like ray generation) against unsorted
            try:
                r4, _ = scenes.primary_rays(cam4, args.width, args.height)
                d_r4 = up(r4)
                d_p4 = torch.zeros(r4.shape[0] * 16, dtype=torch.uint8, device=dev)
                lview4.trace(args.kernel, r4.shape[0], False, d_r4.data_ptr(), d_p4.data_ptr(), stream)
                d_n4 = up(scenes.tri_normals(tri4, pos4))
                per4 = (1 << 20) // args.ao_samples
                lo4 = (r4.shape[0] // 2) // per4 * per4
                nb4 = per4 * args.ao_samples
                b_r = torch.zeros(nb4 * 32, dtype=torch.uint8, device=dev)
                b_o = torch.zeros(nb4 * 16, dtype=torch.uint8, device=dev)
                b_a = torch.zeros(nb4, dtype=i32, device=dev)
                nt.raygen_ao(b_r.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_r4.data_ptr(), d_p4.data_ptr(), d_n4.data_ptr(), lo4, per4, args.ao_samples,
                             cam4["far"], 0xFFF2D5E4, stream)
                so4 = torch.zeros_like(b_r)
                sa4 = torch.zeros(nb4, dtype=i32, device=dev)
                sb4 = torch.zeros(nb4, dtype=i32, device=dev)
                id4 = torch.arange(nb4, dtype=i32, device=dev)
                ss4 = min(nt.ray_morton_sort(nb4, b_r.data_ptr(), id4.data_ptr(), so4.data_ptr(), sa4.data_ptr(), sb4.data_ptr(), stream) for _ in range(3))
                rs = {"rays": nb4, "sort_ms": ss4 * 1e3}
                for kn in (args.kernel, "kepler_dynamic_fetch"):
                    lview4.trace(kn, nb4, False, so4.data_ptr(), b_o.data_ptr(), stream)
                    ts_ = float(np.median([lview4.trace(kn, nb4, False, so4.data_ptr(), b_o.data_ptr(), stream) for _ in range(5)]))
                    lview4.trace(kn, nb4, False, b_r.data_ptr(), b_o.data_ptr(), stream)
                    tu_ = float(np.median([lview4.trace(kn, nb4, False, b_r.data_ptr(), b_o.data_ptr(), stream) for _ in range(5)]))
                    rs[kn] = {"trace_sorted_ms": ts_ * 1e3, "trace_unsorted_ms": tu_ * 1e3}
                cfgs["4"]["ray_sort_one_diffuse_batch"] = rs
                del d_r4, d_p4, d_n4, b_r, b_o, b_a, so4, sa4, sb4, id4
            except Exception as e:
                cfgs["4"]["ray_sort_one_diffuse_batch"] = {"error": repr(e)}
            del keep4, lview4
        except Exception as e:
            cfgs["4"] = {"error": repr(e)}

    # (6) HBM-resident roofline point: the same trace kernel on the 10 M-triangle stand-in for San Miguel.  Its LBVH is 0.75 GB
    # (nodes + Woop + index), three times the 256 MB Infinity Cache, so node and triangle fetches are served by HBM; SURVEY 8(d)
    # accounting against 8 TB/s, with the build's own roofline beside it.
    if not args.no_hbm_point:
        tri10, pos10, cam10 = scenes.courtyard()
        lview, best, info10, keep = lbvh_of(tri10, pos10, 2)
        w, h = args.width, args.height
        r10, _ = scenes.primary_rays(cam10, w, h)
        d_r10 = up(r10)
        d_o10 = torch.zeros(w * h * 16, dtype=torch.uint8, device=dev)
        lview.trace(args.kernel, w * h, False, d_r10.data_ptr(), d_o10.data_ptr(), stream)
        ts = [lview.trace(args.kernel, w * h, False, d_r10.data_ptr(), d_o10.data_ptr(), stream) for _ in range(5)]
        s10 = lview.trace_stats(args.kernel, w * h, False, d_r10.data_ptr(), d_o10.data_ptr(), stream)
        sec = float(np.mean(ts))
        # incoherent closest-hit rays (uniform origins in the bounding box, uniform directions): every ray in its own part of
        # the BVH, so node and triangle fetches miss the caches -- the launch whose bytes HBM really has to deliver
        nr = 1 << 21
        d_rr = up(scenes.box_rays(pos10, nr, seed=21))
        d_ro = torch.zeros(nr * 16, dtype=torch.uint8, device=dev)
        wide10 = bool(lview.flags & nt.BVH_WIDE_LEAVES)
        per_kernel = {}
        for kn in (args.kernel, "kepler_dynamic_fetch"):   # the selector of the frame, and the one made for divergent batches
            lview.trace(kn, nr, False, d_rr.data_ptr(), d_ro.data_ptr(), stream)
            tr = [lview.trace(kn, nr, False, d_rr.data_ptr(), d_ro.data_ptr(), stream) for _ in range(5)]
            per_kernel[kn] = float(np.mean(tr))
        # the same launch without ray splitting in the drain phase (csrc/trace_split.h), for the record
        split_env = os.environ.get("NTR_TRACE_SPLIT_SLICE")
        nt.set_tunables(NTR_TRACE_SPLIT_SLICE=0)
        lview.trace("kepler_dynamic_fetch", nr, False, d_rr.data_ptr(), d_ro.data_ptr(), stream)
        split_off = float(np.mean([lview.trace("kepler_dynamic_fetch", nr, False, d_rr.data_ptr(), d_ro.data_ptr(), stream) for _ in range(3)]))
        nt.set_tunables(NTR_TRACE_SPLIT_SLICE=split_env)   # (back to what the run was started with)
        sr = lview.trace_stats(args.kernel, nr, False, d_rr.data_ptr(), d_ro.data_ptr(), stream)
        best_kn = min(per_kernel, key=per_kernel.get)
        secr = per_kernel[best_kn]

        def roof(st_, s_):
            return {"bound": "hbm", "achieved": st_.algorithmic_bytes() / s_ / 1e9, "peak": hbm_peak, "unit": "GB/s",
                    "frac": st_.algorithmic_bytes() / s_ / 1e9 / hbm_peak, "algorithmic_bytes_per_launch": st_.algorithmic_bytes()}
        r_inc = roof(sr, secr)
        sym = launched_symbol(best_kn, wide10, incoherent=True)
        pmc_i, src_i = load_pmc(sym, launched_grid(best_kn, nr, symbol=sym), tag="courtyard10m")
        bind_i = profiled_shares(pmc_i, src_i, (sr.numInnerVisits + sr.numTriTests) if "persistent" in sym else None) if pmc_i else None
        gr = extras.get("gather_roof", {})
        steps_i = sr.numInnerVisits + sr.numTriTests
        r_inc.update({"kernel": "%s (%s)" % (sym, best_kn), "launch_ms": secr * 1e3,
                      "traffic": bind_i.get("hbm_traffic_bytes") if bind_i else None, "utilisation": bind_i,
                      "utilisation_note": None if bind_i else "no committed PMC summary under profiles/ matches %s with grid %d" % (sym, launched_grid(best_kn, nr, symbol=sym)),
                      "gather_roof": gather_like_for_like(steps_i, secr, gr, pmc_i),
                      "workload": "courtyard-10M device LBVH (0.75 GB), 2^21 incoherent closest-hit rays",
                      "note": "the launch whose bytes HBM really delivers: SURVEY 8(d) algorithmic bytes / time / 8 TB/s; `traffic` = PMC bytes of "
                              "the same kernel symbol (FETCH_SIZE x 2 + WRITE_SIZE) from profiles/"})
        extras["hbm_resident_point"] = {
            "scene": "courtyard-10M stand-in for San Miguel, device LBVH (leafSize 8)", "triangles": int(tri10.shape[0]),
            "bvh_bytes": int(best.nodesBytes + best.triWoopBytes + best.triIndexBytes),
            "primary": {"rays": w * h, "ms": sec * 1e3, "mrays": w * h / sec / 1e6, "trace_stats": s10.as_dict(), "roofline": roof(s10, sec)},
            "incoherent": {"rays": nr, "what": "2^21 closest-hit rays, origins uniform in the bounding box, directions uniform on the sphere",
                           "ms_by_kernel": {k: v * 1e3 for k, v in per_kernel.items()}, "kernel": best_kn,
                           "kepler_dynamic_fetch_without_ray_splitting_ms": split_off * 1e3,
                           "ms": secr * 1e3, "mrays": nr / secr / 1e6, "trace_stats": sr.as_dict(), "roofline": r_inc},
            "note": "HBM-side bytes (FETCH_SIZE / WRITE_SIZE / L2 hit rate) of these launches: profiles/*_trace_courtyard_* summaries",
            "lbvh_build": info10}
        if not args.no_configs:   # BASELINE config 5 itself (primary + 8 x AO on the replicated 10 M-triangle BVH; one GPU traces the whole frame here)
            try:
                c5 = config_point(nt, torch, scenes, dev, stream, up, "5 San Miguel (courtyard-10M stand-in), device LBVH, primary + 8xAO", tri10, pos10, cam10, lview,
                                  best.nodesBytes + best.triWoopBytes + best.triIndexBytes, "ao", (args.kernel, "kepler_dynamic_fetch"), w, h, args.ao_samples, hbm_peak, "courtyard_ao_frame")
                c5["lbvh_build"] = info10
                extras.setdefault("configs", {})["5"] = c5
            except Exception as e:
                extras.setdefault("configs", {})["5"] = {"error": repr(e)}
        del keep, lview, d_r10, d_o10, d_rr, d_ro
    view.trace(args.kernel, batches[0]["n"], False, batches[0]["rays"], batches[0]["res"], stream)  # restore the SAH-BVH primary results
    torch.cuda.synchronize()
    return extras


if __name__ == "__main__":
    main()
