#!/usr/bin/env python3
"""Measured numbers for the five BASELINE.json configurations (seeded stand-ins for the named scenes):
BVH build time, primary, 8xAO (radius 5, any hit) and 8x diffuse (closest hit, to the camera's far plane: Renderer.cpp:533-537) trace rates at
1920x1080 on one GPU, rays / sum of per-batch kernel time as the reference counts them.  Config 3 names the persistent-threads traversal and
config 4 diffuse rays: config 3 is also measured with the two persistent kernel names.  Prints a markdown table."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ntrace_amd as nt
from ntrace_amd import scenes
dev = torch.device("cuda:0")
K = "fermi_speculative_while_while"
def up(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

def measure(name, tri, pos, cam, builder, K=K):
    n = tri.shape[0]
    keep = []
    if builder == "sah":
        t0 = time.time(); bvh = nt.sah_build(tri, pos, 1, 1); build = "%.1f s (host SAH)" % (time.time() - t0)
        d_n, d_w, d_i = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index); keep += [d_n, d_w, d_i]
        view = nt.BvhView(d_n.data_ptr(), bvh.nodes.nbytes, d_w.data_ptr(), bvh.woop.nbytes, d_i.data_ptr())
    else:
        capn, capw, capi = nt.lbvh_capacity(n)
        d_tri, d_pos = up(tri), up(pos)
        d_n = torch.zeros(capn, dtype=torch.uint8, device=dev); d_w = torch.zeros(capw, dtype=torch.uint8, device=dev); d_i = torch.zeros(capi, dtype=torch.uint8, device=dev)
        keep += [d_tri, d_pos, d_n, d_w, d_i]
        best = None
        for _ in range(4):
            r = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), pos.min(0), pos.max(0), 8, 0.001, d_n.data_ptr(), capn, d_w.data_ptr(), capw, d_i.data_ptr(), capi)
            best = r if best is None or r.seconds < best.seconds else best
        build = "%.2f ms (device LBVH, %.2f Gtris/s)" % (best.seconds * 1e3, n / best.seconds / 1e9)
        view = nt.BvhView(d_n.data_ptr(), best.nodesBytes, d_w.data_ptr(), best.triWoopBytes, d_i.data_ptr())
    view.validate()
    w, h = 1920, 1080
    rays, _ = scenes.primary_rays(cam, w, h)
    npr = rays.shape[0]; d_rays = up(rays); d_res = torch.zeros(npr * 16, dtype=torch.uint8, device=dev)
    for _ in range(2): view.trace(K, npr, False, d_rays.data_ptr(), d_res.data_ptr())
    tp = np.median([view.trace(K, npr, False, d_rays.data_ptr(), d_res.data_ptr()) for _ in range(7)])
    hits = nt.count_hits(d_res.data_ptr(), npr)
    d_nrm = up(scenes.tri_normals(tri, pos))
    ns, per = 8, (1 << 20) // 8
    diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
    radius = 5.0 * diag / 4300.0  # config.conf's aoRadius 5 is in Sponza units; scaled to the scene's diagonal
    rate = {}
    for kind, dist_, any_hit in (("ao", radius, True), ("diffuse", cam["far"], False)):
        tt, live, launched = 0.0, 0, 0
        for lo in range(0, npr, per):
            cnt = min(per, npr - lo)
            b_rays = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device=dev); b_res = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device=dev)
            b_a = torch.zeros(cnt * ns, dtype=torch.int32, device=dev)
            nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), lo, cnt, ns, dist_, 0xFFF2D5E4)
            view.trace(K, cnt * ns, any_hit, b_rays.data_ptr(), b_res.data_ptr())
            tt += float(np.median([view.trace(K, cnt * ns, any_hit, b_rays.data_ptr(), b_res.data_ptr()) for _ in range(3)]))
            live += nt.count_hits(d_res.data_ptr() + lo * 16, cnt) * ns
            launched += cnt * ns
        rate[kind] = "%.0f (%.0f)" % (live / tt / 1e6, launched / tt / 1e6) if tt > 0 else "0"
    print("| %s | %d | %s | %s | %.0f | %s | %s | %.1f %% |" % (name, n, build, K.split("_")[0] + ("" if K.startswith("fermi") else " (persistent)"), npr / tp / 1e6, rate["ao"],
                                                                    rate["diffuse"], 100.0 * hits / npr), flush=True)

print("Secondary rates: the reference's count -- rays of primary HITS (Renderer::getTotalNumRays) / sum of per-batch kernel time --, and in brackets all rays of the\nbatches (the rays of missed pixels are degenerate and end at once) / the same time.\n")
print("| config (stand-in) | triangles | BVH build | kernel name | primary Mrays/s | 8xAO Mrays/s | 8x diffuse Mrays/s | primary hit rate |")
print("|---|---|---|---|---|---|---|---|")
ONLY = os.environ.get("CFG_ONLY", "")   # e.g. "4": only that configuration (sweeps)
if ONLY:
    tri, pos, cam = {"2": scenes.atrium, "3": scenes.conference_room, "4": scenes.hairball, "5": scenes.courtyard}[ONLY]()
    measure("%s (CFG_ONLY)" % ONLY, tri, pos, cam, "sah" if ONLY in ("2", "3") else "lbvh")
    sys.exit(0)
tri, pos, cam = scenes.cornell_box(); measure("1 Cornell box", tri, pos, cam, "sah")
tri, pos, cam = scenes.atrium(); measure("2 Sponza (atrium-262k), SAH", tri, pos, cam, "sah"); measure("2 Sponza (atrium-262k), LBVH", tri, pos, cam, "lbvh")
tri, pos, cam = scenes.conference_room()
for kn in (K, "tesla_persistent_while_while", "kepler_dynamic_fetch"):
    measure("3 Conference (room-331k), SAH", tri, pos, cam, "sah", kn)
tri, pos, cam = scenes.hairball(); measure("4 Hairball (2.8 M), LBVH", tri, pos, cam, "lbvh")
tri, pos, cam = scenes.courtyard(); measure("5 San Miguel (courtyard-10M), LBVH", tri, pos, cam, "lbvh")
if "--sah-10m" in sys.argv:
    measure("5 San Miguel (courtyard-10M), SAH", tri, pos, cam, "sah")
