#!/usr/bin/env python3
"""Round 6 diagnostic (needs a library built with the per-wave counters of scripts/studies/rejected_patches/persist_wave_counters.patch:
NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_diag.so): where a persistent launch spends its waves' time on the headline frame's batches.
Per launch: waves, sum of lives, time in the refill section, time in traversal, chunks taken, first start -> last end, last start."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ntrace_amd as nt
from ntrace_amd import scenes, dist as ntd
import bench

dev = torch.device("cuda:0")
L = nt.lib()
L.ntr_debug_stats.restype = C.c_int
L.ntr_debug_stats.argtypes = [C.c_int, C.c_void_p]


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


stream = torch.cuda.current_stream().cuda_stream
args = bench.parse(["--no-extras"])
tri, pos, cam = scenes.atrium()
bvh = nt.sah_build(tri, pos, 1, 1)
keep = [up(bvh.nodes), up(bvh.woop), up(bvh.tri_index)]
view = nt.BvhView(keep[0].data_ptr(), bvh.nodes.nbytes, keep[1].data_ptr(), bvh.woop.nbytes, keep[2].data_ptr())
view.validate(stream)
d_nrm = up(scenes.tri_normals(tri, pos))
frame = bench.Frame(nt, torch, view, lambda d: ntd.FramePlan(1920 * 1080, 0, 1, 8, 1 << 20), cam, 1920, 1080, d_nrm, args, dev, stream, scenes)
settings = sys.argv[1:] or ["-"]
for st in settings:
    env = {} if st == "-" else dict(kv.split("=") for kv in st.split(","))
    nt.set_tunables(**env)
    for kn in ("tesla_persistent_while_while", "kepler_dynamic_fetch"):
        for bi in (0, 3, 9):
            b = frame.batches[bi]
            view.trace(kn, b["n"], b["any_hit"], b["rays"], b["res"], stream, True)
            for _ in range(4):      # (the batch's automatic hint forms: registered, measured, ordered)
                view.trace(kn, b["n"], b["any_hit"], b["rays"], b["res"], stream, True)
            buf = np.zeros((16384, 8), dtype=np.uint64)
            L.ntr_debug_stats(1, None)
            sec = view.trace(kn, b["n"], b["any_hit"], b["rays"], b["res"], stream, True)
            L.ntr_debug_stats(0, buf.ctypes.data)
            v = buf[buf[:, 5] == 1].astype(np.float64)
            t0 = v[:, 0].min()
            life = (v[:, 1] - v[:, 0]) / 100.0
            end = (v[:, 1] - t0) / 100.0
            hw = buf[buf[:, 5] == 1][:, 7]
            xcc = ((hw >> np.uint64(16)) & np.uint64(0xF)).astype(int)     # HW_ID: XCC_ID is not in HW_ID on gfx9; SE_ID bits 13-15 -- kept raw
            se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(int)
            by_se = {int(k): round(float(end[se == k].mean()), 1) for k in np.unique(se)}
            pc = lambda a, q: float(np.percentile(a, q))
            print(json.dumps(dict(setting=st, kernel=kn, batch=bi, rays=b["n"], launch_us=sec * 1e6, waves=int(v.shape[0]), life_us=float(life.mean()),
                                  refill_us=float(v[:, 2].mean() / 100.0), trav_us=float(v[:, 3].mean() / 100.0), first_chunk_us=float(v[:, 6].mean() / 100.0),
                                  chunks_per_wave=float(v[:, 4].mean()), chunks_max=float(v[:, 4].max()), span_us=float(end.max()),
                                  last_start_us=float((v[:, 0].max() - t0) / 100.0), end_pct={q: round(pc(end, q), 1) for q in (1, 10, 50, 90, 99, 99.9)},
                                  life_pct={q: round(pc(life, q), 1) for q in (1, 10, 50, 90, 99, 99.9)}, mean_end_by_se=by_se,
                                  chunks_hist=np.bincount(v[:, 4].astype(int)).tolist())), flush=True)
    nt.set_tunables(**{k: None for k in env})
