#!/usr/bin/env python3
"""Anatomy of one AO launch of the per-ray kernel (experiment build: per-wave start / end stamps): how many waves are resident over time,
how long the launch runs below full occupancy at its start and at its end, how long a wave lives.  A full 2^20-ray AO batch of the
bench frame, in buffer order and in the order the library learns.
usage (exp library): NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_exp.so python3 scripts/studies/ao_launch_timeline.py"""
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
for first in (131072 * 4, 131072 * 9):
    b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev)
    b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, cnt, ns, 5.0, 0xFFF2D5E4)
    n = cnt * ns
    res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    nw = n // 64
    for mode in ("buffer order", "learned order"):
        nt.set_tunables(NTR_TRACE_AUTO_HINT=0 if mode == "buffer order" else None)
        for _ in range(6):
            view.trace("fermi_speculative_while_while", n, True, b_rays.data_ptr(), res.data_ptr())
        plain = min(view.trace("fermi_speculative_while_while", n, True, b_rays.data_ptr(), res.data_ptr()) for _ in range(5))
        tl = torch.zeros(nw * 3, dtype=torch.int64, device=dev)
        nt.experiment_hooks(timeline=tl.data_ptr())
        sec = view.trace("fermi_speculative_while_while", n, True, b_rays.data_ptr(), res.data_ptr())
        nt.experiment_hooks()
        t = tl.cpu().numpy().reshape(-1, 3)
        t = t[t[:, 0] > 0]
        s, e = t[:, 0].astype(np.float64) / 100.0, t[:, 1].astype(np.float64) / 100.0     # us (100 MHz stamps)
        t0 = s.min()
        s -= t0
        e -= t0
        span = e.max()
        grid = np.linspace(0.0, span, 221)
        resident = np.array([((s <= x) & (e > x)).sum() for x in grid])
        full = 0.9 * resident.max()
        above = np.nonzero(resident >= full)[0]
        life = e - s
        # the waves that end last: when did they start, how long did they live, where in the dispatch order were they (wave index = its
        # place in the launch: the kernel maps workgroup i to block order[i / 4])
        late = np.argsort(e)[-64:]
        late_info = dict(start_us=[round(float(x), 1) for x in np.percentile(s[late], [0, 25, 50, 75, 100])],
                         life_us=[round(float(x), 1) for x in np.percentile(life[late], [0, 25, 50, 75, 100])])
        # lives of the second-round waves by start time
        bins = [(15, 25), (25, 35), (35, 50)]
        second = {("%d-%d" % b): (round(float(np.percentile(life[(s >= b[0]) & (s < b[1])], 50)), 1), round(float(np.percentile(life[(s >= b[0]) & (s < b[1])], 95)), 1),
                                  int(((s >= b[0]) & (s < b[1])).sum())) for b in bins if ((s >= b[0]) & (s < b[1])).any()}
        first_round = life[s < 5.0]
        print(json.dumps(dict(first_slot=first, mode=mode, ms_plain=round(plain * 1e3, 4), ms_stamped=round(sec * 1e3, 4), waves=int(t.shape[0]), span_us=round(float(span), 1),
                              peak_resident=int(resident.max()), mean_resident=round(float(resident.mean()), 0),
                              us_until_90pct_of_peak=round(float(grid[above[0]]), 1), us_from_last_90pct_to_end=round(float(span - grid[above[-1]]), 1),
                              last_wave_start_us=round(float(s.max()), 1),
                              wave_life_us=dict(mean=round(float(life.mean()), 1), p50=round(float(np.median(life)), 1), p90=round(float(np.percentile(life, 90)), 1),
                                                max=round(float(life.max()), 1)),
                              life_of_waves_started_in_last_10us=round(float(life[s > s.max() - 10].mean()), 1),
                              last_64_waves_to_end=late_info, second_round_life_p50_p95_count_by_start_us=second,
                              first_round_life_p50_p95_max=[round(float(np.percentile(first_round, 50)), 1), round(float(np.percentile(first_round, 95)), 1), round(float(first_round.max()), 1)],
                              resident_every_5pct=[int(resident[i]) for i in range(0, 221, 11)])), flush=True)
nt.set_tunables(NTR_TRACE_AUTO_HINT=None)
