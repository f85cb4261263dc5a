"""BASELINE.json configs 2-5 as parity cases at their full sizes (seeded stand-ins for the named scenes):
every hit record of every full-size batch checked bit for bit against the CPU oracle (all host cores)."""
import os

import numpy as np
import pytest

import ntrace_amd as nt
from ntrace_amd import scenes
from oracle import oracle

pytestmark = pytest.mark.gpu
K = "fermi_speculative_while_while"
THREADS = os.cpu_count() or 8


def device_lbvh(tri, pos):
    import torch
    from gpu_util import up
    n = tri.shape[0]
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    d_nodes = torch.zeros(capn, dtype=torch.uint8, device="cuda:0")
    d_woop = torch.zeros(capw, dtype=torch.uint8, device="cuda:0")
    d_idx = torch.zeros(capi, dtype=torch.uint8, device="cuda:0")
    mn, mx = oracle.scene_bbox(pos)
    res = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, d_nodes.data_ptr(), capn,
                        d_woop.data_ptr(), capw, d_idx.data_ptr(), capi)
    view = nt.BvhView(d_nodes.data_ptr(), res.nodesBytes, d_woop.data_ptr(), res.triWoopBytes, d_idx.data_ptr())
    view.validate()
    return view, res, (d_nodes, d_woop, d_idx, d_tri, d_pos)


def trace(view, rays, any_hit):
    import torch
    from gpu_util import up
    d_rays = up(rays)
    d_res = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device="cuda:0")
    view.trace(K, rays.shape[0], any_hit, d_rays.data_ptr(), d_res.data_ptr())
    return d_res.cpu().numpy().view(nt.RESULT_DTYPE), d_rays, d_res


def check_sample(view, res, keep, rays, got, any_hit, stride=1):
    """Oracle trace of every ray (stride 1) on the downloaded GPU-built buffers."""
    from gpu_util import assert_parity
    nodes = keep[0].cpu().numpy()[:res.nodesBytes]
    woop = keep[1].cpu().numpy()[:res.triWoopBytes]
    idx = keep[2].cpu().numpy()[:res.triIndexBytes].view(np.int32)
    sel = np.arange(0, rays.shape[0], stride)
    exp, _ = oracle.trace(nodes, woop, idx, rays[sel], any_hit=any_hit, threads=THREADS)
    assert_parity(got[sel], exp, "every record" if stride == 1 else "sample stride %d" % stride)


def test_config3_conference_primary_plus_ao():
    """Conference-class (331 k tris): prebuilt host SAH BVH, 1080p primary + 8 AO per hit, any-hit."""
    import torch
    from gpu_util import DeviceBvh, assert_parity, up
    tri, pos, cam = scenes.conference_room()
    assert tri.shape[0] == 331000
    dbvh = DeviceBvh(nt.sah_build(tri, pos))
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    got, d_rays, d_res = trace(dbvh.view, rays, False)
    exp, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, threads=THREADS)
    assert_parity(got, exp, "conference primary, every record")
    # one AO batch (131072 primaries x 8) generated on the device, traced any-hit, checked in full
    ns, cnt = 8, 131072
    d_nrm = up(scenes.tri_normals(tri, pos))
    d_ao = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device="cuda:0")
    d_a = torch.zeros(cnt * ns, dtype=torch.int32, device="cuda:0")
    nt.raygen_ao(d_ao.data_ptr(), d_a.data_ptr(), d_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(),
                 400000, cnt, ns, 5.0, 0xFFF2D5E4)
    torch.cuda.synchronize()
    ao = d_ao.cpu().numpy().view(nt.RAY_DTYPE)
    d_aores = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device="cuda:0")
    dbvh.view.trace("kepler_dynamic_fetch", cnt * ns, True, d_ao.data_ptr(), d_aores.data_ptr())
    gao = d_aores.cpu().numpy().view(nt.RESULT_DTYPE)
    eao, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, ao, any_hit=True, threads=THREADS)
    assert_parity(gao, eao, "conference AO")
    degenerate = ao["tmax"] < ao["tmin"]
    assert (gao["id"][degenerate] == -1).all()


def test_config4_hairball_gpu_lbvh_refit_then_diffuse():
    """Hairball-class (2.8 M tris): on-device LBVH build + refit, then diffuse secondary rays (closest hit)."""
    import torch
    from gpu_util import assert_parity, up
    tri, pos, cam = scenes.hairball()
    view, res, keep = device_lbvh(tri, pos)
    ref = oracle.lbvh_build(tri, pos, 8, 0.001)
    nodes = keep[0].cpu().numpy()[:res.nodesBytes]
    woop = keep[1].cpu().numpy()[:res.triWoopBytes]
    idx = keep[2].cpu().numpy()[:res.triIndexBytes].view(np.int32)
    assert oracle.bvh_canonical_hash(nodes, woop, idx) == oracle.bvh_canonical_hash(ref["nodes"], ref["woop"], ref["tri_index"])
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    got, d_rays, d_res = trace(view, rays, False)
    check_sample(view, res, keep, rays, got, False)
    # diffuse = AO generator with maxDist = camera far, closest hit (Renderer.cpp:533-537): EVERY batch of the frame, as
    # RayGen::ao cuts them (at most 2^20 output rays per batch, Renderer.cpp:45, RayGen.cpp:582-602), both kernel families
    ns, per = 8, (1 << 20) // 8
    n = rays.shape[0]
    d_nrm = up(scenes.tri_normals(tri, pos))
    d_df = torch.zeros(per * ns * 32, dtype=torch.uint8, device="cuda:0")
    d_a = torch.zeros(per * ns, dtype=torch.int32, device="cuda:0")
    d_dres = torch.zeros(per * ns * 16, dtype=torch.uint8, device="cuda:0")
    batches = live = 0
    for bi, first in enumerate(range(0, n, per)):
        cnt = min(per, n - first)
        nt.raygen_ao(d_df.data_ptr(), d_a.data_ptr(), d_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(),
                     first, cnt, ns, cam["far"], 0xFFF2D5E4)
        torch.cuda.synchronize()
        df = d_df.cpu().numpy().view(nt.RAY_DTYPE)[:cnt * ns].copy()
        exp, _ = oracle.trace(nodes, woop, idx, df, any_hit=False, threads=THREADS)
        for kernel in ((K, "kepler_dynamic_fetch") if bi % 4 == 0 else (K,)):
            d_dres.fill_(0xCD)
            view.trace(kernel, cnt * ns, False, d_df.data_ptr(), d_dres.data_ptr())
            gd = d_dres.cpu().numpy().view(nt.RESULT_DTYPE)[:cnt * ns]
            assert_parity(gd, exp, "hairball diffuse batch %d (%s), every record" % (bi, kernel))
        batches += 1
        live += int((df["tmax"] > df["tmin"]).sum())
    assert batches == (n + per - 1) // per and live > 1000000


def test_lbvh_sort_tile_sizes_and_ticket_path_3_5m_triangles():
    """The one-sweep sort's tile sizes on 3.5 M triangles: 8 / 16 / 24 / 32 keys per thread = 1 709 / 855 / 570 / 428 tiles, of which the first
    and the third are more than the device holds at once (their passes take tiles by ticket) and the others are not (workgroup b sorts
    tile b) -- radix_sort.h onesweep_launch.  Every setting must give the default's buffers byte for byte, and those the oracle's tree."""
    import torch
    tri, pos, cam = scenes.hairball(3_500_000, seed=35)
    ref = None
    try:
        for items in (0, 8, 16, 24, 32):
            nt.set_tunables(NTR_LBVH_SORT_ITEMS=items if items else None)
            view, res, keep = device_lbvh(tri, pos)
            torch.cuda.synchronize()
            got = (keep[0][:res.nodesBytes].clone(), keep[1][:res.triWoopBytes].clone(), keep[2][:res.triIndexBytes].clone())
            if ref is None:
                ref = got
                exp = oracle.lbvh_build(tri, pos, 8, 0.001)
                assert oracle.bvh_canonical_hash(got[0].cpu().numpy(), got[1].cpu().numpy(), got[2].cpu().numpy().view(np.int32)) == \
                    oracle.bvh_canonical_hash(exp["nodes"], exp["woop"], exp["tri_index"])
            for a, b in zip(got, ref):
                assert torch.equal(a, b), items
            del view, keep
    finally:
        nt.set_tunables(NTR_LBVH_SORT_ITEMS=None)


@pytest.mark.parametrize("n", [4095, 4096, 4097, 6143, 6144, 6145, 8191, 8192, 8193, 16385, 40000])
def test_lbvh_sort_tile_sizes_ragged_counts(n):
    """Triangle counts one below, at and above the one-sweep tile sizes (2 048 / 4 096 / 6 144 / 8 192 keys): a ragged last tile, a last tile
    of one key, waves of a tile that hold no key at all -- for every tile size, against the default's buffers and the oracle's tree."""
    import torch
    tri, pos, cam = scenes.random_soup(n, seed=n)
    ref = None
    try:
        for items in (0, 8, 16, 24, 32):
            nt.set_tunables(NTR_LBVH_SORT_ITEMS=items if items else None)
            view, res, keep = device_lbvh(tri, pos)
            torch.cuda.synchronize()
            got = (keep[0][:res.nodesBytes].clone(), keep[1][:res.triWoopBytes].clone(), keep[2][:res.triIndexBytes].clone())
            if ref is None:
                ref = got
                exp = oracle.lbvh_build(tri, pos, 8, 0.001)
                assert oracle.bvh_canonical_hash(got[0].cpu().numpy(), got[1].cpu().numpy(), got[2].cpu().numpy().view(np.int32)) == \
                    oracle.bvh_canonical_hash(exp["nodes"], exp["woop"], exp["tri_index"])
            for a, b in zip(got, ref):
                assert torch.equal(a, b), items
    finally:
        nt.set_tunables(NTR_LBVH_SORT_ITEMS=None)


def test_config5_san_miguel_class_10m_triangles():
    """San-Miguel-class (10 M tris): BVH built once on the device (the host SAH builder is O(n log^2 n)),
    1080p primary; per-rank ray shards traced separately equal the whole-frame trace (what the 8-GPU
    run does with a replicated BVH), plus a sample against the oracle."""
    from ntrace_amd import dist as ntd
    tri, pos, cam = scenes.courtyard()
    assert tri.shape[0] == 10000000
    view, res, keep = device_lbvh(tri, pos)
    assert res.numNodes > 1000000 and res.triWoopBytes < 2 ** 32
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    got, _, _ = trace(view, rays, False)
    n = rays.shape[0]
    parts = []
    for rank in range(8):
        lo, hi = ntd.shard_range(n, rank, 8)
        g, _, _ = trace(view, rays[lo:hi], False)
        parts.append(g)
    whole = np.concatenate(parts)
    assert np.array_equal(whole["id"], got["id"]) and np.array_equal(whole["t"].view(np.uint32), got["t"].view(np.uint32))
    check_sample(view, res, keep, rays, got, False)


def test_config5_san_miguel_class_ao_leg():
    """The 8 x AO leg of configuration 5 on the 10 M-triangle device LBVH: AO batches of 2^20 rays generated on the device from
    the primary hits (rayGenAOKernel semantics, radius scaled to the scene), traced any-hit by both kernel families, every
    record compared with the oracle on the downloaded buffers (Renderer.cpp:501-564, CudaBVH.cpp:1183-1225)."""
    import torch
    from gpu_util import assert_parity, up
    tri, pos, cam = scenes.courtyard()
    view, res, keep = device_lbvh(tri, pos)
    nodes = keep[0].cpu().numpy()[:res.nodesBytes]
    woop = keep[1].cpu().numpy()[:res.triWoopBytes]
    idx = keep[2].cpu().numpy()[:res.triIndexBytes].view(np.int32)
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    got, d_rays, d_res = trace(view, rays, False)
    ns, cnt = 8, (1 << 20) // 8
    diag = float(np.linalg.norm(pos.max(0).astype(np.float64) - pos.min(0)))
    radius = 5.0 * diag / 4300.0           # config.conf's aoRadius 5 is in Sponza units
    d_nrm = up(scenes.tri_normals(tri, pos))
    d_ao = torch.zeros(cnt * ns * 32, dtype=torch.uint8, device="cuda:0")
    d_a = torch.zeros(cnt * ns, dtype=torch.int32, device="cuda:0")
    d_aores = torch.zeros(cnt * ns * 16, dtype=torch.uint8, device="cuda:0")
    hits = 0
    for first in (0, 917504, rays.shape[0] - cnt):   # the first, a middle and the last batch of the frame
        nt.raygen_ao(d_ao.data_ptr(), d_a.data_ptr(), d_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(),
                     first, cnt, ns, radius, 0xFFF2D5E4)
        torch.cuda.synchronize()
        ao = d_ao.cpu().numpy().view(nt.RAY_DTYPE).copy()
        exp, _ = oracle.trace(nodes, woop, idx, ao, any_hit=True, threads=THREADS)
        for kernel in (K, "kepler_dynamic_fetch", "tesla_persistent_while_while"):
            d_aores.fill_(0xCD)
            view.trace(kernel, cnt * ns, True, d_ao.data_ptr(), d_aores.data_ptr())
            gao = d_aores.cpu().numpy().view(nt.RESULT_DTYPE)
            assert_parity(gao, exp, "courtyard-10M AO batch at %d (%s), every record" % (first, kernel))
        degenerate = ao["tmax"] < ao["tmin"]
        assert (exp["id"][degenerate] == -1).all()
        hits += int((exp["id"] >= 0).sum())
    assert hits > 10000   # the radius is large enough for occlusion to occur
