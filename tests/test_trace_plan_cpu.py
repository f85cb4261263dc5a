"""The PLAN half of ntr_trace_bvh (csrc/trace_plan.h through ntr_trace_plan): which kernel variant, grid and scheduling aids a batch
gets, as a pure function of the tunables and the batch -- checked without a device.  The launch half (tests -m gpu) binds run-time
state to exactly this plan.  Reference behaviour it must keep: one selector per kernel file name (CudaBVHTracer.cpp:252-258), grid
sized from the device for persistent kernels (:152-160)."""
import pytest

import ntrace_amd as nt

MB = 1 << 20
# csrc/trace_kernels.h
PERRAY, PERSISTENT, STATS, W2, W1, PERSISTENT_UNIFIED, UNIFIED_W1, UNIFIED_MINI = range(8)


@pytest.fixture(autouse=True)
def default_tunables(monkeypatch):
    import os
    for k in list(os.environ):
        if k.startswith("NTR_"):
            monkeypatch.delenv(k, raising=False)
    nt.set_tunables()
    yield
    nt.set_tunables()


def test_headline_batches_default_selector():
    # 1080p primary batch: per-ray kernel, one-wave workgroups, unified-step loop, may run as ray pools (K decided on the device),
    # predicted on first sight, the library's own hint afterwards
    p = nt.trace_plan("fermi_speculative_while_while", 1920 * 1080, False, 17 * MB, 17 * MB)
    assert (p.variant, p.launchVariant) == (PERRAY, UNIFIED_MINI)
    assert p.numBlocks == 8100 and p.launchBlocks == 4 * 8100 and p.orderBlocks == 8100
    assert p.minipool and p.poolKFromDevice and p.poolKConst == 1 and p.fetchThreshold == 48
    assert p.hintable and p.useAutoHint and p.predictable and p.probeOnRefresh and not p.persistentOrder
    # an AO batch: any hit -> plain unified per-ray launch, no prediction, hint yes
    a = nt.trace_plan("fermi_speculative_while_while", MB, True, 17 * MB, 17 * MB)
    assert a.launchVariant == UNIFIED_W1 and a.launchBlocks == 4 * 4096 and not a.minipool
    assert a.useAutoHint and not a.predictable and not a.probeOnRefresh and a.leafSwitchBelow == 24


def test_small_batches_and_trees_are_left_alone():
    p = nt.trace_plan("fermi_speculative_while_while", 1000, False, 17 * MB, 17 * MB)
    assert not p.useAutoHint and not p.predictable and p.numBlocks == 4 and p.launchBlocks == 16
    p = nt.trace_plan("fermi_speculative_while_while", 2 * MB, False, 64 * 100, 4800)      # Cornell-box class tree
    assert not p.predictable and p.useAutoHint


def test_flags_stats_capture_and_caller_hint():
    s = nt.trace_plan("kepler_dynamic_fetch", MB, False, 17 * MB, 17 * MB, flags=nt._capi.PLAN_FLAG_STATS)
    assert s.variant == STATS and s.launchVariant == STATS and not s.hintable and not s.useAutoHint and not s.predictable
    c = nt.trace_plan("fermi_speculative_while_while", 2 * MB, False, 17 * MB, 17 * MB, flags=nt._capi.PLAN_FLAG_CAPTURING)
    assert not c.useAutoHint and c.predictable       # (a captured launch may still be predicted, from spare scratch)
    h = nt.trace_plan("fermi_speculative_while_while", 2 * MB, False, 17 * MB, 17 * MB, flags=nt._capi.PLAN_FLAG_CALLER_HINT)
    assert h.hintable and not h.useAutoHint


def test_persistent_selectors_grid_and_pool():
    for route in ("1", "0"):
        nt.set_tunables(NTR_TRACE_ROUTE=route)
        for name, unified, thr in (("tesla_persistent_while_while", 0, 0), ("tesla_persistent_speculative_while_while", 0, 0),
                                   ("kepler_dynamic_fetch", 1, 48)):
            p = nt.trace_plan(name, 2 * MB, False, 600 * MB, 700 * MB, num_cus=256)
            if route == "1":    # a large closest-hit launch is routed: kepler_dynamic_fetch's body is the persistent side under every name
                unified, thr = 1, 48
            assert p.variant == PERSISTENT and p.unified == unified and p.fetchThreshold == thr and p.persistentFetchThreshold == thr
            assert p.launchVariant == p.persistentVariant == (PERSISTENT_UNIFIED if unified else PERSISTENT)
            assert p.numBlocks == p.launchBlocks == p.persistentBlocks == 256 * 7 and p.numHeads == 128 and p.chunk == 64
            assert p.shardRays * p.numHeads >= 2 * MB and p.shardRays % 64 == 0
            assert p.numBlocksIncoherent == (256 * 3 if unified else 0) and p.numBlocksDivergent == (256 * 4 if unified else 0)   # only the dynamic-fetch body
            assert p.persistentOrder and p.predictable and p.hintable and p.useAutoHint     # (round 6: the pool is handed out in a hint's order too)
            # large closest-hit launch: BOTH bodies, the device's batch word decides (the per-ray side has a fermi launch's shape)
            assert p.coherentRoute == (1 if route == "1" else 0)
            if route == "1":
                assert p.perrayBlocks == 4 * p.orderBlocks and p.perrayFetchThreshold == 48
        # a batch smaller than the grid: one workgroup per 256 rays; too small for the estimate: the named body alone
        p = nt.trace_plan("kepler_dynamic_fetch", 1000, False, 17 * MB, 17 * MB, num_cus=256)
        assert p.variant == PERSISTENT and p.numBlocks == 4 and p.numBlocksIncoherent == 4 and not p.predictable and p.coherentRoute == 0
        t = nt.trace_plan("tesla_persistent_while_while", 1000, False, 17 * MB, 17 * MB, num_cus=256)
        assert t.unified == 0 and t.fetchThreshold == 0 and t.launchVariant == PERSISTENT
    nt.set_tunables(NTR_TRACE_ROUTE=None)


def test_routing_by_coherence():
    # any-hit launches run the per-ray body under every name
    a = nt.trace_plan("fermi_speculative_while_while", MB, True, 17 * MB, 17 * MB)
    for name in ("tesla_persistent_while_while", "kepler_dynamic_fetch"):
        p = nt.trace_plan(name, MB, True, 17 * MB, 17 * MB)
        assert p.coherentRoute == 2 and (p.variant, p.launchVariant, p.launchBlocks) == (a.variant, a.launchVariant, a.launchBlocks)
        assert p.hintable and p.useAutoHint
    # the per-ray name: large closest-hit launches carry kepler_dynamic_fetch's body beside their own
    f = nt.trace_plan("fermi_speculative_while_while", 2 * MB, False, 600 * MB, 700 * MB, num_cus=256)
    assert f.coherentRoute == 1 and f.launchVariant == UNIFIED_MINI and f.persistentVariant == PERSISTENT_UNIFIED
    assert f.persistentBlocks == 256 * 7 and f.numBlocksIncoherent == 256 * 3 and f.persistentFetchThreshold == 48 and f.numHeads == 128
    small = nt.trace_plan("fermi_speculative_while_while", 1000, False, 17 * MB, 17 * MB)
    assert small.coherentRoute == 0
    # NTR_TRACE_ROUTE=0: the named body, always
    nt.set_tunables(NTR_TRACE_ROUTE=0)
    p = nt.trace_plan("kepler_dynamic_fetch", MB, True, 17 * MB, 17 * MB)
    assert p.coherentRoute == 0 and p.variant == PERSISTENT and p.launchVariant == PERSISTENT_UNIFIED
    assert nt.trace_plan("fermi_speculative_while_while", 2 * MB, False, 600 * MB, 700 * MB).coherentRoute == 0
    assert nt.trace_plan("tesla_persistent_while_while", 2 * MB, False, 600 * MB, 700 * MB).coherentRoute == 0
    nt.set_tunables(NTR_TRACE_ROUTE=None)
    # the stats variant is never routed
    assert nt.trace_plan("kepler_dynamic_fetch", MB, True, 17 * MB, 17 * MB, flags=nt._capi.PLAN_FLAG_STATS).coherentRoute == 0


def test_flat_fetch_needs_one_4gib_window():
    near = nt.trace_plan("kepler_dynamic_fetch", MB, False, 17 * MB, 17 * MB, nodes_addr=1 << 33, woop_addr=(1 << 33) + 3 * (1 << 30))
    far = nt.trace_plan("kepler_dynamic_fetch", MB, False, 17 * MB, 17 * MB, nodes_addr=1 << 33, woop_addr=(1 << 33) + 5 * (1 << 30))
    tiny = nt.trace_plan("kepler_dynamic_fetch", MB, False, 64, 48)
    assert near.flatFetch == 1 and far.flatFetch == 0 and tiny.flatFetch == 0


def test_tunables_steer_the_plan(monkeypatch):
    nt.set_tunables(NTR_TRACE_MINIPOOL=0, NTR_TRACE_BLOCKS_PER_CU=4, NTR_TRACE_POOL_HEADS=4000, NTR_TRACE_PREDICT=0)
    p = nt.trace_plan("fermi_speculative_while_while", 2 * MB, False, 17 * MB, 17 * MB)
    assert p.launchVariant == UNIFIED_W1 and not p.minipool and not p.predictable
    q = nt.trace_plan("tesla_persistent_while_while", 2 * MB, False, 17 * MB, 17 * MB, num_cus=256)
    assert q.numBlocks == 1024 and q.numHeads == 1024 and not q.predictable and q.coherentRoute == 0
    nt.set_tunables(NTR_TRACE_MINIPOOL=None, NTR_TRACE_BLOCKS_PER_CU=None, NTR_TRACE_POOL_HEADS=None, NTR_TRACE_PREDICT=None)
    nt.set_tunables(NTR_TRACE_PERSISTENT_HINTS=0)
    q = nt.trace_plan("kepler_dynamic_fetch", 2 * MB, False, 17 * MB, 17 * MB)
    assert not q.hintable and not q.useAutoHint and q.predictable
    nt.set_tunables(NTR_TRACE_PERSISTENT_HINTS=None)
    nt.set_tunables(NTR_TRACE_MINIPOOL=4, NTR_TRACE_PERRAY_UNIFIED=-1)
    p = nt.trace_plan("fermi_speculative_while_while", 2 * MB, False, 17 * MB, 17 * MB)
    assert p.minipool and p.poolKConst == 4 and not p.poolKFromDevice
    a = nt.trace_plan("fermi_speculative_while_while", MB, True, 17 * MB, 17 * MB)                 # round 3's rule: any hit on a
    assert a.launchVariant == W1                                                                    # one-triangle-leaf tree: while-while
    a = nt.trace_plan("fermi_speculative_while_while", MB, True, 17 * MB, 17 * MB, bvh_flags=16)   # NTR_BVH_WIDE_LEAVES
    assert a.launchVariant == UNIFIED_W1


def test_hint_life_cycle():
    st = nt.trace_plan_hint_step
    assert st(False, False, 0) == dict(zeroK=True, refresh=True, useOrder=False)      # a new hint: measures under buffer order
    assert st(True, False, 1) == dict(zeroK=False, refresh=True, useOrder=True)       # first launches all refresh
    assert st(True, False, 2)["refresh"] and not st(True, False, 3)["refresh"]
    assert st(True, False, 16)["refresh"] and not st(True, False, 17)["refresh"]      # then the 16th, the 32nd,
    assert st(True, False, 32)["refresh"] and not st(True, False, 48)["refresh"]      # ... and every 64th from there on: a schedule that
    assert [u for u in range(3, 300) if st(True, False, u)["refresh"]] == [16, 32, 64, 128, 192, 256]   # has held is re-measured less often
    assert st(True, True, 0) == dict(zeroK=False, refresh=False, useOrder=True)       # a predicted order just runs once


def test_unknown_kernel_and_bad_arguments():
    with pytest.raises(nt.NtrError) as e:
        nt.trace_plan("no_such_kernel", 10, False, 64, 64)
    assert e.value.code == -5
    with pytest.raises(nt.NtrError):
        nt.trace_plan("kepler_dynamic_fetch", -1, False, 64, 64)
