#!/usr/bin/env python3
"""bench.py -- Mrays/s of the BVH trace hot path on MI355X (driver contract in the task brief).

A step = one pass of the hot path over one frame's ray batches on the 'atrium-262k'
stand-in for Crytek Sponza (the OBJ is not in the reference checkout): the 1920x1080
primary batch (closest hit) followed by the 8 x AO batches (any hit, <= 2^20 rays per
batch as Renderer.cpp:45 / RayGen.cpp:582-602), all resident in HBM before the timed
region.  Ray generation is excluded from the metric exactly as in the reference's
runBenchmark (App.cpp:955-969).  The AO batches of a frame are independent launches and
are issued round-robin on --ao-streams HIP streams; the reference's serial protocol (sum
of per-batch kernel times on one stream) is measured too and reported beside it.  One process per GPU; ranks trace their own screen
tile set against a replicated BVH (weak scaling), and the final framebuffer gather over
RCCL is timed separately.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# MI355X_MICROARCH.md, 'Indexed rows: gather' table: chip-wide rate of row gathers served by the XCD L2s
# (16.8-18.8 TB/s) and by the Infinity Cache (8.6 TB/s) -- the practical ceilings of a cache-resident BVH.
L2_GATHER_PEAK_GBS = 18800.0
MALL_GATHER_PEAK_GBS = 8600.0


def pmc_traffic_bytes(n_primary):
    """HBM-side bytes per primary launch from the committed rocprofv3 PMC passes (scripts/profile.sh ->
    profiles/*_pmc_summary.json; FETCH_SIZE and WRITE_SIZE collected in separate passes, in KiB).  gfx950
    correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request, so the read side is
    doubled; WRITE_SIZE is exact.  None when no profile of this launch shape is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        fs = [v for k, v in d.items() if k.endswith("|%d|FETCH_SIZE" % n_primary) and "false" in k]
        ws = [v for k, v in d.items() if k.endswith("|%d|WRITE_SIZE" % n_primary) and "false" in k]
        if fs and ws:
            best = (2.0 * fs[0] + ws[0]) * 1024.0
    return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
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
    ap.add_argument("--ao-streams", type=int, default=3,
                    help="HIP streams the independent AO batches of a frame are issued on (1 = one stream, in buffer order)")
    ap.add_argument("--no-graph", action="store_true", help="issue every frame launch by launch instead of replaying a HIP graph of it")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rays", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import ntrace_amd as nt
    from ntrace_amd import scenes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the tracer has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # NTR_BENCH_FORCE_DIST=1 runs the RCCL code path (init, barrier, all-reduce, gather) at world size 1 too
    use_dist = world > 1 or os.environ.get("NTR_BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
    nt.lib()
    stream = torch.cuda.current_stream().cuda_stream

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    # ---- scene + prebuilt BVH (host SAH build, Renderer.builder = SAHBVH, leaf prefs (1,1)) ----
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
    t0 = time.time()
    bvh = nt.sah_build(tri, pos, 1, 1)
    sah_seconds = time.time() - t0
    d_nodes, d_woop, d_idx = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
    view = nt.BvhView(d_nodes.data_ptr(), bvh.nodes.nbytes, d_woop.data_ptr(), bvh.woop.nbytes, d_idx.data_ptr())
    view.validate(stream)

    # ---- ray batches: rank r renders its own frame of a slow camera move (weak scaling: every rank traces a
    # full-resolution frame; 2 units per rank in a 3 600-unit hall keeps the frames distinct but equally costly,
    # so that the slowest rank measures the machine, not the view) ---------------------------------------
    cam = dict(cam)
    eye = np.array(cam["eye"], dtype=np.float64)
    eye[2] += 2.0 * rank
    cam["eye"] = tuple(eye)
    w, h = args.width, args.height
    n_primary = w * h
    i32 = torch.int32
    d_tab = torch.zeros(n_primary, dtype=i32, device=dev)
    nt.pixel_table(w, h, d_tab.data_ptr(), 0, stream)
    d_rays = torch.zeros(n_primary * 32, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n_primary * 16, dtype=torch.uint8, device=dev)
    d_i2s = torch.zeros(n_primary, dtype=i32, device=dev)
    d_s2i = torch.zeros(n_primary, dtype=i32, device=dev)
    nt.raygen_primary(d_rays.data_ptr(), d_i2s.data_ptr(), d_s2i.data_ptr(), d_tab.data_ptr(), cam["eye"],
                      scenes.nscreen_to_world(cam, w, h), w, h, cam["far"], 0, stream)
    batches = [dict(name="primary", n=n_primary, any_hit=False, rays=d_rays, res=d_res, live=n_primary)]

    def run_batch(b, timed=False):
        return view.trace(args.kernel, b["n"], b["any_hit"], b["rays"].data_ptr(), b["res"].data_ptr(), stream, timed)

    # AO batches (Renderer::nextBatch -> RayGen::ao, batching of RayGen.cpp:582-602: <= 2^20 output
    # rays per batch), generated on the device from the primary hits and kept resident in HBM.
    run_batch(batches[0])
    n_hits = nt.count_hits(d_res.data_ptr(), n_primary, stream)
    ns = args.ao_samples
    if ns > 0:
        d_nrm = up(scenes.tri_normals(tri, pos))
        per = max(args.ao_batch_rays // ns, 1)
        ao_seed = 0xFFF2D5E4  # any fixed kernel seed; Raygen.random = false in config.conf
        for lo in range(0, n_primary, per):
            cnt = min(per, n_primary - lo)
            b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
            b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
            b_a = torch.zeros(cnt * ns, dtype=i32, device=dev)
            nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(),
                         d_nrm.data_ptr(), lo, cnt, ns, args.ao_radius, ao_seed, stream)
            live = nt.count_hits(d_res.data_ptr() + lo * 16, cnt, stream) * ns
            batches.append(dict(name="ao", n=cnt * ns, any_hit=True, rays=b_rays, res=b_res, live=live))
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # A frame = the primary batch, then its AO batches.  The AO batches are independent of each other (each is
    # generated from the primary hits), so they are issued round-robin on a few HIP streams: the tail of one
    # launch overlaps the start of the next.  The primary launch runs alone on the main stream (its event time is
    # the roofline's launch duration) and the next frame's primary waits for every AO stream.
    main_stream = torch.cuda.Stream(device=dev)  # frames are issued, captured and replayed on this stream
    ao_streams = [torch.cuda.Stream(device=dev) for _ in range(args.ao_streams)] if (args.ao_streams > 1 and len(batches) > 2) else []

    def run_step(ms, p0=None, p1=None, a1=None):
        """One frame issued on stream `ms` (+ the AO streams).  The optional timing events bracket the primary
        launch and the AO section."""
        if p0 is not None:
            p0.record(ms)
        view.trace(args.kernel, batches[0]["n"], batches[0]["any_hit"], batches[0]["rays"].data_ptr(), batches[0]["res"].data_ptr(),
                   ms.cuda_stream, False)
        if p1 is None:
            p1 = torch.cuda.Event()
        p1.record(ms)
        if ao_streams:
            for st in ao_streams:
                st.wait_event(p1)
            for i, b in enumerate(batches[1:]):
                view.trace(args.kernel, b["n"], b["any_hit"], b["rays"].data_ptr(), b["res"].data_ptr(),
                           ao_streams[i % len(ao_streams)].cuda_stream, False)
            for st in ao_streams:
                e = torch.cuda.Event()
                e.record(st)
                ms.wait_event(e)
        else:
            for b in batches[1:]:
                view.trace(args.kernel, b["n"], b["any_hit"], b["rays"].data_ptr(), b["res"].data_ptr(), ms.cuda_stream, False)
        if a1 is not None:
            a1.record(ms)

    for _ in range(args.warmup):
        run_step(main_stream)
    barrier()

    # The frame's launches (kernels, stream fork / join) are captured ONCE into a HIP graph; a timed step is one replay of
    # it: every kernel of the frame runs on every replay, only the host's launch work and the gaps it leaves between
    # dependent launches are gone (scripts/graph_frame_experiment.py: 1.41 -> 1.29 ms per frame).  --no-graph issues the
    # frame launch by launch instead, with event timing of the primary launch and the AO section.
    graph = None
    graph_note = "launch by launch"
    if not args.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: calls of other threads (the RCCL watchdog) do not invalidate the capture; the warm-up steps
            # above allocated this stream's scratch buffers, so nothing allocates inside it
            with torch.cuda.graph(graph, stream=main_stream, capture_error_mode="thread_local"):
                run_step(torch.cuda.current_stream())
            graph.replay()  # untimed
            graph_note = "HIP graph replay"
        except Exception as e:  # never lose the measurement to a capture problem: fall back to plain launches
            graph = None
            graph_note = "launch by launch (graph capture failed: %s)" % type(e).__name__
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        barrier()

    # ---- timed region: exactly K steps -------------------------------------------------------------
    step_ms = None
    if graph is not None:
        t0 = time.perf_counter()
        for s in range(args.steps):
            graph.replay()
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for s in range(args.steps):
            run_step(main_stream, *ev[s])
        barrier()
        elapsed = time.perf_counter() - t0
        step_ms = np.array([[p0.elapsed_time(p1), p1.elapsed_time(a1)] for (p0, p1, a1) in ev])  # [steps, (primary, AO section)]

    # the reference's protocol (sum of per-batch kernel times, one stream; App.cpp:955-969), for comparison
    ser_steps = max(3, min(args.steps, 10))
    sev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in batches] for _ in range(ser_steps)]
    for s in range(ser_steps):
        for bi, b in enumerate(batches):
            sev[s][bi][0].record()
            run_batch(b)
            sev[s][bi][1].record()
    torch.cuda.synchronize()
    kern_ms = np.array([[e0.elapsed_time(e1) for (e0, e1) in step] for step in sev])  # [steps, batches], serialized
    # the metric counts non-degenerate rays only (Renderer::getTotalNumRays, Renderer.cpp:676-709)
    rays_per_step = sum(b["live"] for b in batches)

    from ntrace_amd import dist as ntd
    total_rays_per_step, elapsed_max = ntd.job_throughput(rays_per_step, elapsed, dev)  # SUM rays, MAX time

    # ---- final framebuffer gather (hit records -> rank 0) over RCCL, timed separately ----------------
    gather_ms = None
    if use_dist:
        outs = [torch.empty_like(d_res) for _ in range(world)] if rank == 0 else None
        barrier()
        g0 = time.perf_counter()
        dist.gather(d_res, outs, dst=0)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3

    # ---- algorithmic bytes of the dominant kernel (instrumented trace, untimed) ----------------------
    st = view.trace_stats(args.kernel, n_primary, False, d_rays.data_ptr(), d_res.data_ptr(), stream)
    alg_bytes = st.algorithmic_bytes()
    ao_ms_serial = float(kern_ms[:, 1:].sum(axis=1).mean()) if len(batches) > 1 else 0.0
    if step_ms is not None:
        prim_ms = float(step_ms[:, 0].mean())
        ao_ms = float(step_ms[:, 1].mean()) if len(batches) > 1 else 0.0
    else:
        # graph replays carry no timing events: the primary launch's duration is its HIP-event time in the serial pass
        # right after the timed region (same kernels, same buffers), the AO section is the rest of the frame
        prim_ms = float(kern_ms[:, 0].mean())
        ao_ms = max(elapsed / args.steps * 1e3 - prim_ms, 0.0) if len(batches) > 1 else 0.0
    achieved = alg_bytes / (prim_ms * 1e-3) / 1e9
    ao_live = sum(b["live"] for b in batches[1:])
    ao_alg = 0
    for b in batches[1:]:
        sb = view.trace_stats(args.kernel, b["n"], True, b["rays"].data_ptr(), b["res"].data_ptr(), stream)
        ao_alg += sb.algorithmic_bytes()

    # ---- extras (outside the timed region): on-device LBVH build of the same scene, secondary-ray sort ----
    extras = {}
    if rank == 0 and not args.no_extras:
        try:
            # opt-in scheduling hints (ntr_trace_bvh_hinted): block order learned from the previous trace of the
            # same batch.  NOT used for `value`; the same steps re-timed with one hint object per batch.
            hints = [nt.SchedHint() for _ in batches]
            for _ in range(4):
                for b, hnt in zip(batches, hints):
                    view.trace(args.kernel, b["n"], b["any_hit"], b["rays"].data_ptr(), b["res"].data_ptr(), stream, False, hint=hnt)
            torch.cuda.synchronize()
            hsteps = max(3, min(args.steps, 10))
            hev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in batches] for _ in range(hsteps)]
            h0 = time.perf_counter()
            for s_ in range(hsteps):
                for bi, (b, hnt) in enumerate(zip(batches, hints)):
                    hev[s_][bi][0].record()
                    view.trace(args.kernel, b["n"], b["any_hit"], b["rays"].data_ptr(), b["res"].data_ptr(), stream, False, hint=hnt)
                    hev[s_][bi][1].record()
            torch.cuda.synchronize()
            hwall = time.perf_counter() - h0
            hms = np.array([[e0.elapsed_time(e1) for (e0, e1) in st_] for st_ in hev])
            extras["sched_hints"] = {
                "what": "same batches re-traced with ntr_trace_bvh_hinted (block order from the previous trace of the batch); "
                        "identical hit records; helps repeated / static batches, not a moving camera (DESIGN.md 4.1)",
                "mrays_wall": rays_per_step * hsteps / hwall / 1e6,
                "primary_mrays": n_primary / (float(hms[:, 0].mean()) * 1e-3) / 1e6,
                "ao_mrays": (sum(b["live"] for b in batches[1:]) / (float(hms[:, 1:].sum(axis=1).mean()) * 1e-3) / 1e6) if len(batches) > 1 else None,
                "primary_ms": float(hms[:, 0].mean()), "steps": hsteps}
            for hnt in hints:
                hnt.close()
            # practical HBM ceiling: device-to-device copy of a buffer larger than the Infinity Cache
            cp_src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            cp_dst = torch.empty_like(cp_src)
            cp_dst.copy_(cp_src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                cp_dst.copy_(cp_src)
            e1.record()
            torch.cuda.synchronize()
            extras["stream_copy_GBps"] = 2.0 * 4 * cp_src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del cp_src, cp_dst
            capn, capw, capi = nt.lbvh_capacity(tri.shape[0])
            d_tri, d_pos = up(tri), up(pos)
            ln = torch.zeros(capn, dtype=torch.uint8, device=dev)
            lw = torch.zeros(capw, dtype=torch.uint8, device=dev)
            li = torch.zeros(capi, dtype=torch.uint8, device=dev)
            best = None
            for _ in range(3):
                r = nt.lbvh_build(tri.shape[0], d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), pos.min(0), pos.max(0), 8, 0.001,
                                  ln.data_ptr(), capn, lw.data_ptr(), capw, li.data_ptr(), capi, stream)
                best = r if best is None or r.seconds < best.seconds else best
            lview = nt.BvhView(ln.data_ptr(), best.nodesBytes, lw.data_ptr(), best.triWoopBytes, li.data_ptr())
            lview.validate(stream)
            lsec = min(lview.trace(args.kernel, n_primary, False, d_rays.data_ptr(), d_res.data_ptr(), stream) for _ in range(5))
            extras["lbvh"] = {"build_ms": best.seconds * 1e3, "mtris_per_s": tri.shape[0] / best.seconds / 1e6,
                              "phases_ms": {"morton": best.mortonMs, "sort": best.sortMs, "triangle_box_terms": best.woopMs,
                                            "emit_top_pass": best.emitMs, "emit_subtrees+refit+woop_placement": best.refitMs}, "nodes": best.numNodes, "leaves": best.numLeaves,
                              "primary_mrays_on_lbvh": n_primary / lsec / 1e6}
            # algorithmic bytes of the build (SURVEY 8d accounting, exact from the counts): Morton 48 rd + 8 wr,
            # 4 radix passes x (8 rd + 8 wr + 4 rd histogram), Woop 48 rd + 48 wr per triangle; emit 12 rd + 16 wr per
            # inner node, (48+4) rd + (48+12) wr per triangle, 20 wr per leaf; refit (12+36) rd per triangle,
            # 96 rd per inner child, 48 wr per node.
            nt_, ni_, nl_ = int(tri.shape[0]), int(best.numNodes), int(best.numLeaves)
            lb = nt_ * (56 + 4 * 20 + 96 + 112 + 48) + ni_ * (28 + 48) + nl_ * 20 + max(ni_ - 1, 0) * 96
            extras["lbvh"]["algorithmic_bytes"] = lb
            extras["lbvh"]["roofline"] = {"bound": "hbm", "achieved": lb / best.seconds / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": lb / best.seconds / 1e9 / HBM_PEAK_GBS}
            run_batch(batches[0])  # restore the SAH-BVH primary results
            if len(batches) > 1:
                b1 = batches[1]
                so = torch.zeros_like(b1["rays"])
                sa = torch.zeros(b1["n"], dtype=i32, device=dev)
                sb = torch.zeros(b1["n"], dtype=i32, device=dev)
                ident = torch.arange(b1["n"], dtype=i32, device=dev)
                ssec = min(nt.ray_morton_sort(b1["n"], b1["rays"].data_ptr(), ident.data_ptr(), so.data_ptr(), sa.data_ptr(),
                                              sb.data_ptr(), stream) for _ in range(3))
                sres = torch.zeros_like(b1["res"])
                tsec = min(view.trace(args.kernel, b1["n"], True, so.data_ptr(), sres.data_ptr(), stream) for _ in range(5))
                usec = min(view.trace(args.kernel, b1["n"], True, b1["rays"].data_ptr(), b1["res"].data_ptr(), stream) for _ in range(5))
                extras["ray_sort"] = {"rays": b1["n"], "sort_ms": ssec * 1e3, "trace_sorted_ms": tsec * 1e3, "trace_unsorted_ms": usec * 1e3}
        except Exception as e:  # extras never invalidate the headline
            extras["error"] = repr(e)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    value = total_rays_per_step * args.steps / elapsed_max / 1e6
    out = {
        "metric": "Mrays/sec (primary + 8xAO) on Crytek Sponza",
        "value": value,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": ("synthetic (%s)" % scene_name) if not args.scene_obj else scene_name,
        "config": {"workload": "Sponza-262k prebuilt SAH BVH, %dx%d primary + %dxAO (radius %g) per GPU" % (w, h, ns, args.ao_radius),
                   "kernel": args.kernel, "bvh_flags": view.flags, "triangles": int(tri.shape[0]), "rays_per_step_per_gpu": rays_per_step,
                   "primary_rays": n_primary, "primary_hits": n_hits, "ao_rays_nondegenerate": ao_live,
                   "ao_batches": len(batches) - 1, "ao_streams": max(1, len(ao_streams)),
                   "frame_issue": graph_note,
                   "parallelism": "screen-tile sharded rays, BVH replicated, RCCL gather of hit records"},
        "primary_mrays": n_primary / (prim_ms * 1e-3) / 1e6,
        "ao_mrays": (ao_live / (ao_ms * 1e-3) / 1e6) if ao_ms > 0 else None,
        "kernel_ms": {"primary": prim_ms, "ao_total": ao_ms},
        "reference_protocol": {"what": "sum of per-batch kernel times, all batches on one stream (App.cpp:955-969)",
                               "primary_ms": float(kern_ms[:, 0].mean()), "ao_total_ms": ao_ms_serial,
                               "primary_mrays": n_primary / (float(kern_ms[:, 0].mean()) * 1e-3) / 1e6,
                               "ao_mrays": (ao_live / (ao_ms_serial * 1e-3) / 1e6) if ao_ms_serial > 0 else None},
        "gather_ms": gather_ms,
        "host_sah_build_s": sah_seconds,
        "trace_stats": st.as_dict(),
        "extras": extras,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(n_primary),
                     "kernel": "trace_bvh (%s), primary batch" % args.kernel,
                     "launch_ms": prim_ms,
                     "launch_includes": "predict_kernel + flatten_kernel (dispatch-order prediction, about 30 us) + trace_bvh_perray; "
                                        "rocprofv3's per-kernel average for trace_bvh_perray alone is in profiles/*_rocprof_summary.txt",
                     "note": "algorithmic bytes (SURVEY 8d accounting) / HIP-event time; > 1 means the bytes are served by L1/L2/"
                             "Infinity Cache: measured HBM-side traffic is in `traffic` (bytes per launch)",
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "cache_ceilings": {"l2_gather_peak": L2_GATHER_PEAK_GBS, "frac_of_l2_gather": achieved / L2_GATHER_PEAK_GBS,
                                        "infinity_cache_gather_peak": MALL_GATHER_PEAK_GBS,
                                        "source": "MI355X_MICROARCH.md gather table; the 34 MB BVH is cache-resident"},
                     "ao": {"achieved": (ao_alg / (ao_ms * 1e-3) / 1e9) if ao_ms > 0 else None,
                            "algorithmic_bytes_all_batches": ao_alg}},
    }

    # ---- CPU baseline: the oracle (restated reference CPU tracer) on a bounded sample ---------------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        cores = os.cpu_count() or 1
        rays = d_rays.cpu().numpy().view(nt.RAY_DTYPE)
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
            hr = b["rays"].cpu().numpy().view(nt.RAY_DTYPE)
            t0c = time.perf_counter()
            ref, _ = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, hr, any_hit=b["any_hit"], threads=cores)
            cpu_s += time.perf_counter() - t0c
            got = b["res"].cpu().numpy().view(nt.RESULT_DTYPE)
            mism += int(((got["id"] != ref["id"]) | (got["t"].view(np.uint32) != ref["t"].view(np.uint32))).sum())
            traced += b["n"]
        out["cpu_baseline"] = {"value": rays_per_step / cpu_s / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                               "sample": "one whole step (primary + %d AO batches = %d rays, %d non-degenerate) on all %d host "
                                         "cores, %.2f s wall = %.1f CPU-s; 1 thread on the first %d primary rays: %.3f Mrays/s"
                                         % (len(batches) - 1, traced, rays_per_step, cores, cpu_s, cpu_s * cores, n1,
                                            n1 / (c1 - c0) / 1e6),
                               "single_thread_mrays": n1 / (c1 - c0) / 1e6,
                               "parity_mismatches_whole_step": mism, "rays_compared": traced}
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
