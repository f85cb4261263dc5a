#!/usr/bin/env python3
"""Which kernel for which batch?  Per scene (device LBVH, or host SAH for atrium) and per batch kind -- 1080p primary, one
2^20-ray AO batch (any hit), one 2^20-ray diffuse batch (closest hit, far), 2^21 incoherent rays -- the launch time of the
per-ray kernel and of the persistent kernels (whole-wave refill / dynamic fetch at several thresholds), plus the lane
utilisation of the per-ray launch (lane visits / (wave-iterations x 64), from the exact counters).  One JSON line per
(scene, batch, kernel).

usage: kernel_matrix.py <scene>[,<scene>...] [reps]     scenes: atrium hairball courtyard conference"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import lbvh, scene_of, up  # noqa: E402

dev = torch.device("cuda:0")
VARIANTS = [("fermi_speculative_while_while", {"NTR_TRACE_MINIPOOL": "0"}),   # the plain per-ray kernel
            ("fermi_speculative_while_while", {}),                            # default: mini-pool depth decided per batch on the device
            ("kepler_dynamic_fetch", {}),
            ("tesla_persistent_while_while", {})]
EXTRA = [e for e in os.environ.get("KM_EXTRA_ENV", "").split(";") if e]   # e.g. "NTR_TRACE_CHUNK=128;NTR_TRACE_POOL_HEADS=256"


def main():
    names = sys.argv[1].split(",")
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for scene in names:
        tri, pos, cam = scene_of(scene)
        keep = []
        if scene in ("atrium", "conference"):
            bvh = nt.sah_build(tri, pos, 1, 1)
            d_n, d_w, d_i = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
            view = nt.BvhView(d_n.data_ptr(), bvh.nodes.nbytes, d_w.data_ptr(), bvh.woop.nbytes, d_i.data_ptr())
            keep += [d_n, d_w, d_i]
        else:
            best, bufs = lbvh(tri, pos, 2)
            keep += list(bufs)
            view = nt.BvhView(bufs[0].data_ptr(), best.nodesBytes, bufs[1].data_ptr(), best.triWoopBytes, bufs[2].data_ptr())
        view.validate()
        w, h = 1920, 1080
        rays, _ = scenes.primary_rays(cam, w, h)
        npr = rays.shape[0]
        d_rays = up(rays)
        d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
        view.trace(VARIANTS[0][0], npr, False, d_rays.data_ptr(), d_res.data_ptr())
        d_nrm = up(scenes.tri_normals(tri, pos))
        ns, cnt = 8, (1 << 20) // 8
        diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
        radius = 5.0 if scene == "atrium" else 5.0 * diag / 4300.0
        first = min(900000, npr - cnt)
        batches = [("primary", npr, False, d_rays, d_res)]
        for nm, dist_, anyh in (("ao", radius, True), ("diffuse", cam["far"], False)):
            b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
            b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
            b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
            nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt,
                         ns, dist_, 0xFFF2D5E4)
            batches.append((nm, cnt * ns, anyh, b_rays, b_res))
        nr = 1 << 21
        d_rr = up(scenes.box_rays(pos, nr, seed=21))
        batches.append(("incoherent", nr, False, d_rr, torch.zeros(nr * 16, dtype=torch.uint8, device=dev)))
        torch.cuda.synchronize()
        for (bname, n, anyh, br, bo) in batches:
            ref = None
            st = view.trace_stats(VARIANTS[0][0], n, anyh, br.data_ptr(), bo.data_ptr())
            for (kernel, env) in VARIANTS:
                for k, v in env.items():
                    os.environ[k] = v
                for e in EXTRA:
                    k, v = e.split("=")
                    os.environ[k] = v
                nt.set_tunables()
                bo.zero_()
                view.trace(kernel, n, anyh, br.data_ptr(), bo.data_ptr())
                ts = [view.trace(kernel, n, anyh, br.data_ptr(), bo.data_ptr()) for _ in range(reps)]
                got = bo.clone()
                same = None
                if ref is None:
                    ref = got
                else:
                    same = bool(torch.equal(ref.view(torch.int32).view(-1, 4)[:, :2], got.view(torch.int32).view(-1, 4)[:, :2]))
                for k in env:
                    os.environ.pop(k, None)
                print(json.dumps(dict(scene=scene, batch=bname, rays=n, any_hit=anyh, kernel=kernel, env=env, extra=EXTRA,
                                      ms_min=float(np.min(ts)) * 1e3, ms_mean=float(np.mean(ts)) * 1e3,
                                      mrays=n / float(np.min(ts)) / 1e6, records_equal_perray=same,
                                      per_ray=dict(inner=st.numInnerVisits / n, tris=st.numTriTests / n, leaves=st.numLeafVisits / n),
                                      alg_bytes=st.algorithmic_bytes(), hbm_frac=st.algorithmic_bytes() / float(np.min(ts)) / 8e12)),
                      flush=True)
        del keep, batches, view


if __name__ == "__main__":
    main()
