#!/usr/bin/env python3
"""Why are divergent batches slow?  Per scene and batch kind: the distribution of per-ray traversal steps (inner visits +
triangle tests, from the oracle's per-ray counters on the downloaded buffers; a strided sample), the same per 64-ray wave
(max over the wave = what a per-ray-kernel wave runs), and the launch times of the per-ray and dynamic-fetch kernels.
critical_path_us = longest ray's steps x the measured mean time of a wave iteration: no schedule of whole rays beats it.

usage: divergence_study.py <scene>[,<scene>...] [sample_rays]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402  (analysis script, not product)
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")


def per_ray_counts(nodes, woop, idx, rays, any_hit):
    L = oracle.lib()
    n = rays.shape[0]
    res = np.zeros(n, dtype=oracle.RESULT_DTYPE)
    inner = np.zeros(n, dtype=np.int32)
    tris = np.zeros(n, dtype=np.int32)
    vp = C.c_void_p
    L.orc_trace_compact_counts.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp]
    L.orc_trace_compact_counts.restype = C.c_int
    rays = np.ascontiguousarray(rays)
    rc = L.orc_trace_compact_counts(nodes.ctypes.data, woop.ctypes.data, idx.ctypes.data, rays.ctypes.data, res.ctypes.data, n, int(any_hit),
                                    inner.ctypes.data, tris.ctypes.data)
    assert rc == 0
    return inner, tris


def pct(a):
    a = np.asarray(a, dtype=np.float64)
    return dict(mean=float(a.mean()), p50=float(np.percentile(a, 50)), p90=float(np.percentile(a, 90)), p99=float(np.percentile(a, 99)),
                p999=float(np.percentile(a, 99.9)), max=float(a.max()))


def main():
    names = sys.argv[1].split(",")
    sample = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 17
    for scene in names:
        tri, pos, cam = scene_of(scene)
        if scene in ("atrium", "conference"):
            bvh = nt.sah_build(tri, pos, 1, 1)
            h_n, h_w, h_i = bvh.nodes, bvh.woop, bvh.tri_index
            d_n, d_w, d_i = up(h_n), up(h_w), up(h_i)
            view = nt.BvhView(d_n.data_ptr(), h_n.nbytes, d_w.data_ptr(), h_w.nbytes, d_i.data_ptr())
        else:
            best, bufs = lbvh(tri, pos, 2)
            view = nt.BvhView(bufs[0].data_ptr(), best.nodesBytes, bufs[1].data_ptr(), best.triWoopBytes, bufs[2].data_ptr())
            h_n = bufs[0].cpu().numpy()[:best.nodesBytes]
            h_w = bufs[1].cpu().numpy()[:best.triWoopBytes]
            h_i = bufs[2].cpu().numpy()[:best.triIndexBytes].view(np.int32)
        view.validate()
        w, h = 1920, 1080
        rays, _ = scenes.primary_rays(cam, w, h)
        npr = rays.shape[0]
        d_rays = up(rays)
        d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
        view.trace("fermi_speculative_while_while", npr, False, d_rays.data_ptr(), d_res.data_ptr())
        d_nrm = up(scenes.tri_normals(tri, pos))
        ns, cnt = 8, (1 << 20) // 8
        first = min(900000, npr - cnt)
        b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
        b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
        nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns,
                     cam["far"], 0xFFF2D5E4)
        torch.cuda.synchronize()
        nr = 1 << 21
        inc = scenes.box_rays(pos, nr, seed=21)
        batches = [("primary", rays, d_rays), ("diffuse", b_rays.cpu().numpy().view(nt.RAY_DTYPE), b_rays), ("incoherent", inc, up(inc))]
        for (bname, hr, dr) in batches:
            n = hr.shape[0]
            d_o = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
            times = {}
            for kernel, env in (("fermi_speculative_while_while", {}), ("kepler_dynamic_fetch", {"NTR_TRACE_FETCH_THRESHOLD": "48"})):
                nt.set_tunables(**env)
                view.trace(kernel, n, False, dr.data_ptr(), d_o.data_ptr())
                times[kernel] = min(view.trace(kernel, n, False, dr.data_ptr(), d_o.data_ptr()) for _ in range(3)) * 1e3
                nt.set_tunables(**{k: None for k in env})
            # per-ray steps on whole waves of the batch: `sample` rays as 64-ray groups spread over the batch
            nw = max(sample // 64, 1)
            wave_ids = np.linspace(0, n // 64 - 1, nw).astype(np.int64)
            sel = (wave_ids[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
            inner, tris = per_ray_counts(h_n, h_w, h_i, hr[sel], False)
            steps = (inner + tris).astype(np.int64)
            live = steps > 0
            wave_max = steps.reshape(-1, 64).max(axis=1)
            wave_sum = steps.reshape(-1, 64).sum(axis=1)
            # what a wave-private mini-pool would buy: a wave owns K x 64 consecutive rays and a lane that finishes takes the wave's next
            # unstarted ray (greedy list scheduling on 64 lanes); iterations of the wave = its makespan
            import heapq
            pool = {}
            for K in (1, 2, 4, 8):
                tot_it = 0
                longest = 0
                groups = steps[: (steps.size // (64 * K)) * 64 * K].reshape(-1, 64 * K)
                for g in groups:
                    lanes = list(g[:64].astype(int))
                    heapq.heapify(lanes)
                    for x in g[64:]:
                        t = heapq.heappop(lanes)
                        heapq.heappush(lanes, t + int(x))
                    mk = max(lanes)
                    tot_it += mk
                    longest = max(longest, mk)
                pool[K] = dict(util=float(groups.sum() / max(tot_it * 64.0, 1)), wave_iterations=int(tot_it), longest_wave=int(longest))
            print(json.dumps(dict(scene=scene, batch=bname, rays=n, sample=int(sel.size), ms=times, wave_private_pool=pool,
                                  steps_per_ray=pct(steps[live]) if live.any() else None,
                                  wave_max_steps=pct(wave_max), lane_util_perray_model=float(wave_sum.sum() / (wave_max.sum() * 64.0)),
                                  top_rays_share=dict(top_1pct=float(np.sort(steps)[-max(sel.size // 100, 1):].sum() / max(steps.sum(), 1)),
                                                      top_01pct=float(np.sort(steps)[-max(sel.size // 1000, 1):].sum() / max(steps.sum(), 1))))),
                  flush=True)


if __name__ == "__main__":
    main()
