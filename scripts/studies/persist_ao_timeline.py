#!/usr/bin/env python3
"""Where does a persistent wave's time go on an AO batch?  The per-ray kernel traces a 2^20-ray AO batch of the bench frame in 55 us, the
persistent selectors in 147-155 us.  Per-wave stamps of the persistent kernel (experiment build: start, end, cycles spent in the refill
section, refills, rays taken) for `tesla_persistent_while_while` and `kepler_dynamic_fetch` at a few grids, next to the per-ray kernel's
launch time.  usage (exp library): NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_exp.so python3 scripts/studies/persist_ao_timeline.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ntrace_amd as nt  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from workloads import up  # noqa: E402

dev = torch.device("cuda:0")
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
d_nodes, d_woop, d_idx = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)
view = nt.BvhView(d_nodes.data_ptr(), bvh.nodes.nbytes, d_woop.data_ptr(), bvh.woop.nbytes, d_idx.data_ptr())
view.validate()
rays, _ = scenes.primary_rays(cam, 1920, 1080)
npr = rays.shape[0]
d_rays = up(rays)
d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
view.trace("fermi_speculative_while_while", npr, False, d_rays.data_ptr(), d_res.data_ptr())
d_nrm = up(scenes.tri_normals(tri, pos))
ns, cnt = 8, (1 << 20) // 8
first = 900000
b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, 5.0, 0xFFF2D5E4)
n = cnt * ns
res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
per_ray = min(view.trace("fermi_speculative_while_while", n, True, b_rays.data_ptr(), res.data_ptr()) for _ in range(6))
print(json.dumps(dict(kernel="fermi_speculative_while_while (per-ray)", ms=round(per_ray * 1e3, 4))), flush=True)
for kernel in ("tesla_persistent_while_while", "kepler_dynamic_fetch"):
    for bpc in (6, 8):
        os.environ["NTR_TRACE_BLOCKS_PER_CU"] = str(bpc)
        nt.set_tunables()
        nw = 256 * bpc * 4
        for _ in range(3):
            view.trace(kernel, n, True, b_rays.data_ptr(), res.data_ptr())
        plain = min(view.trace(kernel, n, True, b_rays.data_ptr(), res.data_ptr()) for _ in range(5))
        tl = torch.zeros(nw * 6, dtype=torch.int64, device=dev)
        nt.experiment_hooks(timeline=tl.data_ptr())
        sec = view.trace(kernel, n, True, b_rays.data_ptr(), res.data_ptr())
        nt.experiment_hooks()
        t_ = tl.cpu().numpy().reshape(-1, 6)
        t_ = t_[t_[:, 0] > 0]
        s, e = t_[:, 0].astype(np.float64), t_[:, 1].astype(np.float64)       # 100 MHz realtime stamps
        t0 = s.min()
        life = (e - s) / 100.0
        refill_cyc, refills, rays_taken, end_cyc = t_[:, 2].astype(np.float64), t_[:, 3], t_[:, 4], t_[:, 5]
        print(json.dumps(dict(kernel=kernel, blocks_per_cu=bpc, waves=int(t_.shape[0]), ms_plain=round(plain * 1e3, 4), ms_with_stamps=round(sec * 1e3, 4),
                              launch_span_us=round(float((e.max() - t0) / 100.0), 1), first_wave_start_spread_us=round(float((s.max() - t0) / 100.0), 1),
                              wave_life_us=dict(mean=round(float(life.mean()), 1), p10=round(float(np.percentile(life, 10)), 1), p90=round(float(np.percentile(life, 90)), 1),
                                                max=round(float(life.max()), 1)),
                              refill_cycles_per_wave=round(float(refill_cyc.mean()), 0), refill_share_of_life=round(float((refill_cyc / 2400.0).mean() / max(life.mean(), 1e-9)), 3),
                              refills_per_wave=round(float(refills.mean()), 2), rays_per_wave=round(float(rays_taken.mean()), 1))), flush=True)
os.environ.pop("NTR_TRACE_BLOCKS_PER_CU", None)
