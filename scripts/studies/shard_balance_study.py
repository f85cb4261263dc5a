#!/usr/bin/env python3
"""Strong-scaling load balance of one 1080p frame, simulated on ONE GPU: for N = 2, 4, 8 the work of every rank (its slice of
the primary batch, the AO batches from its own hits) is traced in turn; the job's rate is total rays / MAX over ranks.

Plans (all contiguous in the PixelTable index space, so a rank owns compact screen tiles):
  equal-count        FramePlan's round-2 cut: N ranges of equal ray count
  balanced-<f>       ntrace_amd.dist.balanced_cuts: equal PREDICTED cost, block weight = predictor + f * mean(predictor)
Per rank two figures: protocol_ms = sum of the per-batch kernel times (the reference's metric, one launch at a time) and
overlap_ms = the same launches issued asynchronously, the AO batches round-robin on three streams behind the primary batch,
timed by one pair of events (what an application not bound to synchronous launches gets).
Also fits T_rank ~ a * sum(predictor) + b * blocks over 16 equal-count slices: b / (a * mean predictor) is the flat share.

usage: shard_balance_study.py [scene ...]      scenes: atrium (host SAH) courtyard hairball (device LBVH)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import dist as ntd  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"
E = torch.cuda.Event


def study(scene):
    tri, pos, cam = scene_of(scene)
    keep = []
    if scene in ("atrium", "conference"):
        bvh = nt.sah_build(tri, pos)
        d_n, d_w, d_i = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
        keep += [d_n, d_w, d_i]
        view = nt.BvhView(d_n.data_ptr(), bvh.nodes.nbytes, d_w.data_ptr(), bvh.woop.nbytes, d_i.data_ptr())
        nodes_ptr, nodes_bytes = d_n.data_ptr(), bvh.nodes.nbytes
        radius = 5.0
    else:
        best, bufs = lbvh(tri, pos, 2)
        keep += list(bufs)
        view = nt.BvhView(bufs[0].data_ptr(), best.nodesBytes, bufs[1].data_ptr(), best.triWoopBytes, bufs[2].data_ptr())
        nodes_ptr, nodes_bytes = bufs[0].data_ptr(), best.nodesBytes
        radius = 5.0 * float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0))) / 4300.0
    view.validate()
    w, h, ns = 1920, 1080, 8
    rays, _ = scenes.primary_rays(cam, w, h)
    n = rays.shape[0]
    full = up(rays)
    d_nrm = up(scenes.tri_normals(tri, pos))
    nb = (n + 255) // 256
    d_cost = torch.zeros(nb, dtype=torch.int32, device=dev)
    nt.predict_block_costs(n, full.data_ptr(), nodes_ptr, nodes_bytes, d_cost.data_ptr())
    torch.cuda.synchronize()
    cost = d_cost.cpu().numpy().astype(np.float64)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    main = torch.cuda.Stream(device=dev)

    def rank_time(lo, hi, slots=None):
        """(protocol_ms, overlap_ms) of the rank that owns primary slots [lo, hi) -- or the slots `slots` (a striped plan), gathered
        into a buffer of their own."""
        m = hi - lo if slots is None else int(slots.numel())
        if m <= 0:
            return 0.0, 0.0
        own = None
        if slots is not None:
            own = full.view(-1, 32).index_select(0, slots).contiguous().view(-1)
        r_ptr = full.data_ptr() + lo * 32 if own is None else own.data_ptr()
        res = torch.zeros(m * 16, dtype=torch.uint8, device=dev)
        per = (1 << 20) // ns
        view.trace(K, m, False, r_ptr, res.data_ptr())
        ao = []
        for first in range(0, m, per):
            cnt = min(per, m - first)
            b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
            b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
            b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
            nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), r_ptr, res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, radius, 0xFFF2D5E4)
            ao.append((b_rays, b_res, cnt * ns))
        torch.cuda.synchronize()
        best = None
        for _ in range(4):
            t = view.trace(K, m, False, r_ptr, res.data_ptr())
            for (br, bs, k) in ao:
                t += view.trace(K, k, True, br.data_ptr(), bs.data_ptr())
            best = t if best is None or t < best else best
        ov = None
        for _ in range(4):
            e0, e1 = E(enable_timing=True), E(enable_timing=True)
            e0.record(main)
            view.trace(K, m, False, r_ptr, res.data_ptr(), main.cuda_stream, False)
            p1 = E()
            p1.record(main)
            for st in streams:
                st.wait_event(p1)
            for i, (br, bs, k) in enumerate(ao):
                view.trace(K, k, True, br.data_ptr(), bs.data_ptr(), streams[i % 3].cuda_stream, False)
            for st in streams:
                ev = E()
                ev.record(st)
                main.wait_event(ev)
            e1.record(main)
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            ov = t if ov is None or t < ov else ov
        return best * 1e3, ov

    one_p, one_o = rank_time(0, n)
    print(json.dumps(dict(scene=scene, ranks=1, plan="whole frame", protocol_ms=one_p, overlap_ms=one_o, predictor_mean=float(cost.mean()))), flush=True)
    # fit on 16 equal-count slices
    sl = [ntd.shard_range(n, r, 16) for r in range(16)]
    ts = np.array([rank_time(lo, hi)[0] for (lo, hi) in sl])
    A = np.array([[cost[lo // 256:(hi + 255) // 256].sum(), (hi - lo) / 256.0] for (lo, hi) in sl])
    coef, *_ = np.linalg.lstsq(A, ts, rcond=None)
    a, b = float(coef[0]), float(coef[1])
    flat_fit = b / (a * cost.mean()) if a > 0 and cost.mean() > 0 else None
    print(json.dumps(dict(scene=scene, fit="T_rank ~ a * sum(predictor) + b * blocks over 16 equal-count slices", a=a, b=b, flat_share_fit=flat_fit,
                          slice_ms=[round(float(x), 4) for x in ts], rel_residual=float(np.abs(A @ coef - ts).mean() / ts.mean()))), flush=True)
    for N in (2, 4, 8):
        plans = {"equal-count": [ntd.shard_range(n, r, N)[0] for r in range(N)] + [n]}
        for f in (0.5, 1.0, 2.0, 4.0):
            plans["balanced-%g" % f] = ntd.balanced_cuts(cost, n, N, f)
        if flat_fit is not None and flat_fit > 0:
            plans["balanced-fit(%.2f)" % flat_fit] = ntd.balanced_cuts(cost, n, N, flat_fit)
        for stripe in (1024, 4096, 16384, 65536):   # rank r owns the stripes s with s % N == r (stripe = that many consecutive PixelTable slots)
            idx = torch.arange(n, device=dev)
            tt = [rank_time(0, 0, idx[((idx // stripe) % N) == r]) for r in range(N)]
            tp = [x[0] for x in tt]
            to = [x[1] for x in tt]
            print(json.dumps(dict(scene=scene, ranks=N, plan="striped-%d" % stripe, max_ms=max(tp), mean_ms=float(np.mean(tp)), min_ms=min(tp),
                                  efficiency=one_p / max(tp) / N, overlap_max_ms=max(to), overlap_efficiency=one_o / max(to) / N)), flush=True)
        for name, cuts in plans.items():
            tt = [rank_time(cuts[r], cuts[r + 1]) for r in range(N)]
            tp = [x[0] for x in tt]
            to = [x[1] for x in tt]
            print(json.dumps(dict(scene=scene, ranks=N, plan=name, max_ms=max(tp), mean_ms=float(np.mean(tp)), min_ms=min(tp),
                                  efficiency=one_p / max(tp) / N, overlap_max_ms=max(to), overlap_efficiency=one_o / max(to) / N,
                                  overlap_speedup_vs_protocol_one=one_p / max(to), cuts=cuts)), flush=True)
    del keep


for sc in (sys.argv[1:] or ["atrium", "courtyard"]):
    study(sc)
