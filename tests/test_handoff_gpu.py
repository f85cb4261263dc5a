"""GPU parity of the tail hand-off (round 4; measured 5-10 % slower than leaving the tails alone, so it lives in the A/B build of the
library only -- libntrace_amd_ab.so, `make -C ntrace_amd/csrc ab` -- and these tests run against that library): a pool wave whose own rays are all started and of which only a few are still live
appends those rays' complete traversal state to a continuation queue and exits, or fills its free lanes from that queue.  A ray goes
on exactly where it stood -- its visiting order cannot change -- so every record must stay the oracle's, whatever the thresholds."""
import numpy as np
import pytest

import ntrace_amd as nt
from ntrace_amd import scenes
from oracle import oracle
from ray_sets import edge_rays

pytestmark = pytest.mark.gpu
K = "fermi_speculative_while_while"


@pytest.fixture(autouse=True)
def ab_library(request):
    """Every test of this module calls the A/B build; the product library is restored afterwards."""
    import os
    if not os.path.exists(nt.ab_lib_path()):
        pytest.fail("libntrace_amd_ab.so missing: __graft_entry__.build() makes it (make -C ntrace_amd/csrc ab)")
    nt.use_library(nt.ab_lib_path())
    request.addfinalizer(lambda: (nt.use_library(None), nt.set_tunables()))

HANDOFF_ENV = ("NTR_TRACE_HANDOFF", "NTR_TRACE_HANDOFF_BELOW", "NTR_TRACE_HANDOFF_MIN_QUEUE", "NTR_TRACE_HANDOFF_KEEP_WAVES", "NTR_TRACE_HANDOFF_FLAGS",
               "NTR_TRACE_MINIPOOL", "NTR_TRACE_MINIPOOL_THRESHOLD")


def _lbvh_device_bvh(tri, pos):
    import torch
    from gpu_util import DeviceBvh, up
    n = tri.shape[0]
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    bufs = [torch.zeros(c, dtype=torch.uint8, device="cuda:0") for c in (capn, capw, capi)]
    mn, mx = oracle.scene_bbox(pos)
    res = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, bufs[0].data_ptr(), capn, bufs[1].data_ptr(), capw,
                        bufs[2].data_ptr(), capi)
    torch.cuda.synchronize()
    return DeviceBvh(nt.HostBvh(bufs[0].cpu().numpy()[:res.nodesBytes].copy(), bufs[1].cpu().numpy()[:res.triWoopBytes].copy(),
                                bufs[2].cpu().numpy()[:res.triIndexBytes].view(np.int32).copy()))


@pytest.mark.parametrize("tree", ["sah leaves of 1", "device lbvh"])
def test_tail_handoff_changes_no_record(monkeypatch, tree):
    """Pool K 1 / 2 / 4 / 7 (K = 1 through NTR_TRACE_HANDOFF_FLAGS bit 1), hand-off thresholds from 'a wave with one free lane' to 'only
    the last ray', queue thresholds from 'take whatever waits' to 'a full refill', with and without the end-game rule and the raised
    priority, ragged counts, scattered + coherent + edge-case rays (an SAH tree with one-triangle leaves is 25 levels deep: stacks beyond
    the slot's 24 entries stay where they are).  Every ray handed off is taken up again (appended == taken), and some are."""
    from gpu_util import DeviceBvh, assert_parity, gpu_trace
    tri, pos, cam = scenes.random_soup(30000, seed=31)
    dbvh = DeviceBvh(nt.sah_build(tri, pos, 1, 1)) if tree == "sah leaves of 1" else _lbvh_device_bvh(tri, pos)
    rays = np.concatenate([scenes.random_rays(150000, seed=5), edge_rays(), scenes.primary_rays(cam, 160, 120)[0]])
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=False, threads=8)
    handed = 0
    configs = []
    for k, flags in (("1", 2), ("2", 0), ("4", 1), ("7", 3)):
        for below, minq, keep in (("16", "64", "0"), ("63", "1", "0"), ("2", "64", "0"), ("24", "8", "256"), ("33", "16", "100000")):
            configs.append({"NTR_TRACE_HANDOFF": "1", "NTR_TRACE_MINIPOOL": k, "NTR_TRACE_HANDOFF_FLAGS": str(flags), "NTR_TRACE_HANDOFF_BELOW": below,
                            "NTR_TRACE_HANDOFF_MIN_QUEUE": minq, "NTR_TRACE_HANDOFF_KEEP_WAVES": keep})
    configs.append({"NTR_TRACE_MINIPOOL": "4", "NTR_TRACE_HANDOFF": "0"})
    configs.append({"NTR_TRACE_HANDOFF": "1", "NTR_TRACE_MINIPOOL": "2", "NTR_TRACE_MINIPOOL_THRESHOLD": "64", "NTR_TRACE_HANDOFF_BELOW": "64", "NTR_TRACE_HANDOFF_KEEP_WAVES": "0"})
    for env in configs:
        for k in HANDOFF_ENV:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        nt.set_tunables()
        for n in (rays.shape[0], 100000, 4097, 65, 1):
            got, _ = gpu_trace(K, dbvh, rays[:n], False)
            assert_parity(got, ref[:n], "%s hand-off %s n=%d" % (tree, env, n))
            import torch
            pushed, popped, cap = nt.trace_handoff_counts(torch.cuda.current_stream().cuda_stream)
            assert pushed == popped, (env, n, pushed, popped)
            if env.get("NTR_TRACE_HANDOFF") == "0":
                continue
            if n == rays.shape[0]:
                handed += pushed
    assert handed > 0, "no configuration handed a single ray off: the test does not reach the queue"
    assert nt.trace_status() == 0


def test_tail_handoff_on_two_streams_and_repeated_launches(monkeypatch):
    """Each stream owns its queue; a batch re-traced under its learned order keeps handing off and stays exact."""
    import torch
    from gpu_util import DeviceBvh, assert_parity, up
    tri, pos, cam = scenes.random_soup(20000, seed=37)
    dbvh = _lbvh_device_bvh(tri, pos)
    for k in HANDOFF_ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("NTR_TRACE_HANDOFF", "1")
    monkeypatch.setenv("NTR_TRACE_MINIPOOL", "2")
    monkeypatch.setenv("NTR_TRACE_HANDOFF_KEEP_WAVES", "0")
    monkeypatch.setenv("NTR_TRACE_AUTO_HINT_MIN_RAYS", "1")
    nt.set_tunables()
    sets = [scenes.random_rays(120000, seed=s) for s in (1, 2)]
    refs = [oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, r, any_hit=False, threads=8)[0] for r in sets]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    d_rays = [up(r) for r in sets]
    d_res = [torch.zeros(r.shape[0] * 16, dtype=torch.uint8, device="cuda:0") for r in sets]
    torch.cuda.synchronize()
    for rep in range(6):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                d_res[i].zero_()
                dbvh.view.trace(K, sets[i].shape[0], False, d_rays[i].data_ptr(), d_res[i].data_ptr(), streams[i].cuda_stream, False)
        torch.cuda.synchronize()
        for i in (0, 1):
            assert_parity(d_res[i].cpu().numpy().view(nt.RESULT_DTYPE), refs[i], "stream %d launch %d" % (i, rep))
            pushed, popped, _ = nt.trace_handoff_counts(streams[i].cuda_stream)
            assert pushed == popped and pushed > 0, (i, rep, pushed, popped)
    assert nt.trace_status() == 0


@pytest.mark.parametrize("tree", ["sah leaves of 1", "device lbvh"])
def test_per_ray_launch_with_ray_splitting_changes_no_record(monkeypatch, tree):
    """The A/B build's per-ray / mini-pool launch with ray splitting compiled in (NTR_TRACE_SPLIT_PERRAY=1; trace_split.h -- measured, it
    only pays in the drain phase of the persistent kernels, which is where the product has it): every lane of a wave that is done takes
    over stack entries of the lanes still on their way, with and without the wave-private pool.  Records stay the oracle's."""
    from gpu_util import DeviceBvh, assert_parity, gpu_trace
    tri, pos, cam = scenes.random_soup(30000, seed=37)
    dbvh = DeviceBvh(nt.sah_build(tri, pos, 1, 1)) if tree == "sah leaves of 1" else _lbvh_device_bvh(tri, pos)
    rays = np.concatenate([scenes.box_rays(pos, 200000, seed=6), edge_rays(), scenes.primary_rays(cam, 320, 200)[0]])
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=False, threads=8)
    try:
        monkeypatch.setenv("NTR_TRACE_SPLIT_PERRAY", "1")
        for pool in ("1", "4", "-1"):
            for slice_ in ("1", "8", "32"):
                monkeypatch.setenv("NTR_TRACE_MINIPOOL", pool)
                monkeypatch.setenv("NTR_TRACE_SPLIT_SLICE", slice_)
                nt.set_tunables()
                for n in (rays.shape[0], 63, 4097):
                    got, _ = gpu_trace(K, dbvh, rays[:n], False)
                    assert_parity(got, ref[:n], "%s per-ray split, pool %s slice %s n=%d" % (tree, pool, slice_, n))
    finally:
        for k in ("NTR_TRACE_SPLIT_PERRAY", "NTR_TRACE_MINIPOOL", "NTR_TRACE_SPLIT_SLICE"):
            monkeypatch.delenv(k, raising=False)
        nt.set_tunables()
