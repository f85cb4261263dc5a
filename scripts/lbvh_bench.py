#!/usr/bin/env python3
"""LBVH build timing + trace throughput on the built tree (configs 2 and 4 of BASELINE.json).
Usage: python scripts/lbvh_bench.py [--tris N ...]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="atrium")
    ap.add_argument("--tris", type=int, default=2800000)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    tri, pos, cam = scenes.atrium() if args.scene == "atrium" else scenes.hairball(args.tris)
    n = tri.shape[0]

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    d_tri, d_pos = up(tri), up(pos)
    capn, capw, capi = nt.lbvh_capacity(n)
    d_nodes = torch.zeros(capn, dtype=torch.uint8, device=dev)
    d_woop = torch.zeros(capw, dtype=torch.uint8, device=dev)
    d_idx = torch.zeros(capi, dtype=torch.uint8, device=dev)
    mn, mx = pos.min(0), pos.max(0)
    best = None
    for _ in range(args.reps):
        r = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, d_nodes.data_ptr(), capn,
                          d_woop.data_ptr(), capw, d_idx.data_ptr(), capi, stream)
        if best is None or r.seconds < best.seconds:
            best = r
    view = nt.BvhView(d_nodes.data_ptr(), best.nodesBytes, d_woop.data_ptr(), best.triWoopBytes, d_idx.data_ptr())
    view.validate(stream)
    w, h = 1920, 1080
    rays, _ = scenes.primary_rays(cam, w, h)
    d_rays = up(rays)
    d_res = torch.zeros(w * h * 16, dtype=torch.uint8, device=dev)
    times = [view.trace("fermi_speculative_while_while", w * h, False, d_rays.data_ptr(), d_res.data_ptr(), stream) for _ in range(6)]
    st = view.trace_stats("fermi_speculative_while_while", w * h, False, d_rays.data_ptr(), d_res.data_ptr(), stream)
    out = dict(scene=args.scene, triangles=n, build=best.as_dict(), bvh_flags=view.flags,
               primary_ms=min(times) * 1e3, primary_mrays=w * h / min(times) / 1e6, trace_stats=st.as_dict(),
               build_mtris_per_s=n / best.seconds / 1e6)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
