#!/usr/bin/env python3
"""What does one traversal step cost when nothing hides its latency?  The longest rays of an incoherent batch (per-ray step counts from
the oracle's counters on a sample of whole waves) traced alone: 1 ray, 8, 64 (one wave), 64 x 64 (one wave per SIMD quarter...),
with the per-ray kernel.  time / steps of the longest ray = the latency of a dependent step (instructions + one memory round trip) --
the slope of every divergent launch's tail.
usage: tail_step_latency.py <scene>[,<scene>...] [sample_rays]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))                    # scripts/ (workloads.py)
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402
from divergence_study import per_ray_counts  # noqa: E402

dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"


def main():
    sample = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
    for scene in sys.argv[1].split(","):
        tri, pos, cam = scene_of(scene)
        if scene in ("atrium", "conference"):
            bvh = nt.sah_build(tri, pos, 1, 1)
            h_n, h_w, h_i = bvh.nodes, bvh.woop, bvh.tri_index
            keep = [up(h_n), up(h_w), up(h_i)]
            view = nt.BvhView(keep[0].data_ptr(), h_n.nbytes, keep[1].data_ptr(), h_w.nbytes, keep[2].data_ptr())
        else:
            best, keep = lbvh(tri, pos, 2)
            view = nt.BvhView(keep[0].data_ptr(), best.nodesBytes, keep[1].data_ptr(), best.triWoopBytes, keep[2].data_ptr())
            h_n = keep[0].cpu().numpy()[:best.nodesBytes]
            h_w = keep[1].cpu().numpy()[:best.triWoopBytes]
            h_i = keep[2].cpu().numpy()[:best.triIndexBytes].view(np.int32)
        view.validate()
        rays = scenes.box_rays(pos, sample, seed=21)
        inner, tris = per_ray_counts(h_n, h_w, h_i, rays, False)
        steps = (inner + tris).astype(np.int64)
        order = np.argsort(-steps)
        for count in (1, 8, 64, 64 * 16, 64 * 256):
            sel = order[:count]
            sub = np.ascontiguousarray(rays[sel])
            d_r = up(sub)
            d_o = torch.zeros(count * 16, dtype=torch.uint8, device=dev)
            nt.set_tunables(NTR_TRACE_MINIPOOL="0", NTR_TRACE_AUTO_HINT="0")
            view.trace(K, count, False, d_r.data_ptr(), d_o.data_ptr())
            ts = [view.trace(K, count, False, d_r.data_ptr(), d_o.data_ptr()) for _ in range(5)]
            t = min(ts)
            mx = int(steps[sel].max())
            print(json.dumps(dict(scene=scene, rays=count, longest_ray_steps=mx, mean_steps=float(steps[sel].mean()), inner_share=float(inner[sel].sum() / max(steps[sel].sum(), 1)),
                                  ms=t * 1e3, us_per_step_of_longest=t * 1e6 / mx)), flush=True)


if __name__ == "__main__":
    main()
