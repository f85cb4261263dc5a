#!/usr/bin/env python3
"""Tail hand-off on the device: launch times of the divergent batches with the continuation queue off / on over
its thresholds, records compared with the hand-off-free launch of the same batch.

usage: tail_handoff_gpu.py <scene>[,<scene>...] [quick|full]
Batches: 2^21 box rays (incoherent), one 2^20-ray diffuse batch (closest hit), the 1080p primary batch (coherent: must not move)."""
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
KERNEL = "fermi_speculative_while_while"
ENV = ("NTR_TRACE_HANDOFF", "NTR_TRACE_HANDOFF_BELOW", "NTR_TRACE_HANDOFF_MIN_QUEUE", "NTR_TRACE_HANDOFF_KEEP_WAVES", "NTR_TRACE_HANDOFF_FLAGS",
       "NTR_TRACE_MINIPOOL")


def run(view, rays, env, reps=5, warm=3):
    nt.set_tunables(**{k: None for k in ENV})
    nt.set_tunables(**env)
    n = rays.shape[0]
    d_rays = up(rays)   # a new buffer: a new automatic hint (first launch registers, second is predicted, from the third the measured order)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    for _ in range(warm):
        view.trace(KERNEL, n, False, d_rays.data_ptr(), d_res.data_ptr())
    ts = [view.trace(KERNEL, n, False, d_rays.data_ptr(), d_res.data_ptr()) * 1e3 for _ in range(reps)]
    pushed, popped, cap = nt.trace_handoff_counts(0)
    got = d_res.cpu().numpy().view(nt.RESULT_DTYPE).copy()
    return dict(ms_min=round(min(ts), 4), ms_mean=round(float(np.mean(ts)), 4), handed_off=pushed, taken=popped), got


def main():
    names = sys.argv[1].split(",")
    mode = sys.argv[2] if len(sys.argv) > 2 else "quick"
    for scene in names:
        tri, pos, cam = scene_of(scene)
        if scene in ("atrium", "conference"):
            bvh = nt.sah_build(tri, pos, 1, 1)
            keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
            view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
        else:
            best, keep = lbvh(tri, pos, 2)
            view = nt.BvhView(keep[0].data_ptr(), best.nodesBytes, keep[1].data_ptr(), best.triWoopBytes, keep[2].data_ptr())
        view.validate()
        prim = scenes.primary_rays(cam, 1920, 1080)[0]
        npr = prim.shape[0]
        d_prim = up(prim)
        d_pres = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
        view.trace(KERNEL, npr, False, d_prim.data_ptr(), d_pres.data_ptr())
        d_nrm = up(scenes.tri_normals(tri, pos))
        ns, cnt = 8, (1 << 20) // 8
        first = min(900000, npr - cnt)
        b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
        b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
        nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_prim.data_ptr(), d_pres.data_ptr(), d_nrm.data_ptr(), first, cnt, ns,
                     cam["far"], 0xFFF2D5E4)
        torch.cuda.synchronize()
        diffuse = b_rays.cpu().numpy().view(nt.RAY_DTYPE).copy()
        batches = [("incoherent", scenes.box_rays(pos, 1 << 21, seed=21)), ("diffuse", diffuse), ("primary", prim)]
        for bname, rays in batches:
            base, ref = run(view, rays, {"NTR_TRACE_HANDOFF": "0"})
            print(json.dumps(dict(scene=scene, batch=bname, rays=int(rays.shape[0]), config="handoff off (K by the device)", **base)), flush=True)
            cfgs = []
            if bname == "primary":
                cfgs = [{}]   # defaults: the coherent batch must not move
            else:
                for k in (("1", "2", "4") if mode == "full" else ("1", "4")):
                    base_k, _ = run(view, rays, {"NTR_TRACE_HANDOFF": "0", "NTR_TRACE_MINIPOOL": k})
                    print(json.dumps(dict(scene=scene, batch=bname, config="handoff off, K=%s" % k, **base_k)), flush=True)
                    ts = ("8", "16", "24", "32") if mode == "full" else ("16", "32")
                    ks = ("0", "1024", "3072") if mode == "full" else ("0", "1024")
                    for t, a in itertools.product(ts, ks):
                        cfgs.append({"NTR_TRACE_HANDOFF": "1", "NTR_TRACE_MINIPOOL": k, "NTR_TRACE_HANDOFF_BELOW": t, "NTR_TRACE_HANDOFF_KEEP_WAVES": a,
                                     "NTR_TRACE_HANDOFF_FLAGS": "2" if k == "1" else "0"})
                    cfgs.append({"NTR_TRACE_HANDOFF": "1", "NTR_TRACE_MINIPOOL": k, "NTR_TRACE_HANDOFF_BELOW": "24", "NTR_TRACE_HANDOFF_KEEP_WAVES": "1024", "NTR_TRACE_HANDOFF_MIN_QUEUE": "16",
                                 "NTR_TRACE_HANDOFF_FLAGS": "2" if k == "1" else "0"})
                    cfgs.append({"NTR_TRACE_HANDOFF": "1", "NTR_TRACE_MINIPOOL": k, "NTR_TRACE_HANDOFF_BELOW": "24", "NTR_TRACE_HANDOFF_KEEP_WAVES": "1024",
                                 "NTR_TRACE_HANDOFF_FLAGS": "3" if k == "1" else "1"})
                cfgs.append({})   # the defaults
            for env in cfgs:
                r, got = run(view, rays, env)
                eq = bool((got["id"] == ref["id"]).all() and (got["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
                print(json.dumps(dict(scene=scene, batch=bname, config=env or "defaults", records_equal=eq, **r)), flush=True)
        nt.set_tunables(**{k: None for k in ENV})
        assert nt.trace_status() == 0


if __name__ == "__main__":
    main()
