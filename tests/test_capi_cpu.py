"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, and the argument checks that precede any device work behave like the
reference's host-side checks.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ntrace_amd as nt
from ntrace_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ntrace_amd.h")).read()
    return sorted(set(re.findall(r"NTR_API\s+[\w\s\*]+?\b(ntr_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = nt.lib()
    names = header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libntrace_amd.so does not export %s" % n
    bound = {s[0] for s in _capi.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)


def test_query_config_known_and_unknown_kernels():
    for k in nt.KERNELS:
        cfg = nt.query_config(k)
        assert cfg.bvhLayout == nt.BVHLayout_Compact
        assert cfg.blockWidth == 64 and cfg.blockHeight >= 1
    assert nt.query_config("fermi_speculative_while_while").usePersistentThreads == 0
    assert nt.query_config("kepler_dynamic_fetch").usePersistentThreads == 1
    with pytest.raises(nt.NtrError) as e:
        nt.query_config("no_such_kernel")
    assert e.value.code == -5


def test_trace_argument_checks_precede_device_work():
    # empty batch -> 0 seconds, no error (CudaBVHTracer.cpp:92-94)
    assert nt.trace_bvh("kepler_dynamic_fetch", 0, False, 0, 0, 0, 0, 0, 0, 0) == 0.0
    # missing BVH (CudaBVHTracer.cpp:97-98)
    with pytest.raises(nt.NtrError) as e:
        nt.trace_bvh("kepler_dynamic_fetch", 10, False, 1, 1, 0, 0, 0, 0, 0)
    assert "No BVH" in str(e.value)
    # wrong layout (CudaBVHTracer.cpp:99-100)
    with pytest.raises(nt.NtrError) as e:
        nt.trace_bvh("kepler_dynamic_fetch", 10, False, 1, 1, 1, 64, 1, 16, 1, layout=0)
    assert e.value.code == -4 and "Incorrect BVH layout" in str(e.value)
    with pytest.raises(nt.NtrError):
        nt.trace_bvh("bogus", 10, False, 1, 1, 1, 64, 1, 16, 1)


def test_no_cpu_fallback_without_device():
    """On a box without a GPU a compute call must fail loudly, never silently compute."""
    cnt = C.c_int(-1)
    rc = nt.lib().ntr_device_count(C.byref(cnt))
    if rc == 0 and cnt.value > 0:
        pytest.skip("a GPU is present")
    rays = np.zeros(64, dtype=nt.RAY_DTYPE)
    buf = np.zeros(4096, dtype=np.uint8)
    with pytest.raises(nt.NtrError) as e:
        nt.trace_bvh("fermi_speculative_while_while", 64, False, rays.ctypes.data, buf.ctypes.data,
                     buf.ctypes.data, 4096, buf.ctypes.data, 4096, buf.ctypes.data)
    # bad extents are rejected before any device work
    with pytest.raises(nt.NtrError) as e2:
        nt.trace_bvh("fermi_speculative_while_while", 64, False, rays.ctypes.data, buf.ctypes.data,
                     buf.ctypes.data, 100, buf.ctypes.data, 4096, buf.ctypes.data)
    assert e2.value.code == -1
    assert e.value.code in (-2, -3)


def test_sched_hint_object_lifecycle_without_device():
    """NtrSchedHint is a host object: create / reset / destroy work without a GPU; tracing with it does not."""
    import ctypes as C
    L = nt.lib()
    h = C.c_void_p()
    assert L.ntr_sched_hint_create(C.byref(h)) == 0 and h.value
    assert L.ntr_sched_hint_reset(h) == 0
    assert L.ntr_sched_hint_reset(None) != 0
    assert L.ntr_sched_hint_create(None) != 0
    assert L.ntr_sched_hint_destroy(h) == 0
    assert L.ntr_sched_hint_destroy(None) == 0


def test_multi_gpu_partition_and_diagnostics_argument_checks():
    """The round-4 entry points check their arguments before any device or RCCL work: the partition arithmetic is pure host code, the
    diagnostics reject null / out-of-range arguments, and a group call without a group fails with a message instead of crashing."""
    L = nt.lib()
    lo, hi = C.c_int32(-1), C.c_int32(-1)
    assert L.ntr_frame_shard(1000, 0, 4, 64, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (0, 256)
    assert L.ntr_frame_shard(1000, 3, 4, 64, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (768, 1000)
    for bad in ((1000, 4, 4, 64), (1000, -1, 4, 64), (1000, 0, 0, 64), (-1, 0, 1, 64), (1000, 0, 1, 0)):
        assert L.ntr_frame_shard(*bad, C.byref(lo), C.byref(hi)) == -1
    assert L.ntr_frame_shard(1000, 0, 1, 64, None, C.byref(hi)) == -1
    n = C.c_int32(-1)
    assert L.ntr_frame_ao_batches(0, 1000, 8, 1 << 20, None, None, 0, C.byref(n)) == 0 and n.value == 1
    assert L.ntr_frame_ao_batches(0, 1000, 0, 1 << 20, None, None, 0, C.byref(n)) == 0 and n.value == 0     # no samples: no batches
    assert L.ntr_frame_ao_batches(10, 5, 8, 1 << 20, None, None, 0, C.byref(n)) == -1
    assert L.ntr_frame_ao_batches(0, 1000, 8, 1 << 20, None, None, 4, C.byref(n)) == -1                     # capacity without arrays
    assert nt.frame_ao_batches(64, 1064, 8, 4096) == [(64, 512), (576, 488)]
    # diagnostics
    assert not hasattr(L, "ntr_trace_handoff_counts") and not hasattr(L, "ntr_experiment_hooks")   # rejected experiments / diagnostic hooks are patches (scripts/studies/rejected_patches/), never in the library
    sec = C.c_float(0.0)
    assert L.ntr_selftest_gather_rate(16, 1, 64, 1, None, C.byref(sec)) == -1          # table too small
    assert L.ntr_selftest_gather_rate(1 << 20, 1, 65, 1, None, C.byref(sec)) == -1     # more lanes than a wave has
    assert L.ntr_selftest_gather_rate(1 << 20, 1, 64, 1, None, None) == -1
    pairs, bad = C.c_uint64(0), C.c_uint64(0)
    assert L.ntr_selftest_division_hard(0, 0, None, C.byref(bad), None) == -1
    assert L.ntr_selftest_division_hard(52, 0, C.byref(pairs), C.byref(bad), None) == -1    # operands would leave the FASTDIV range
    assert L.ntr_selftest_division_hard(0, -41, C.byref(pairs), C.byref(bad), None) == -1
    # dispatch hints from a prediction: argument checks come before any device work
    assert L.ntr_sched_hint_predict(None, None, 16, None) == -1
    assert L.ntr_secondary_block_costs(None, 0, 0, 8, None, 0, None, None) == 0          # an empty batch: nothing to do
    assert L.ntr_secondary_block_costs(None, 0, 16, 8, None, 0, None, None) == -1
    assert L.ntr_secondary_block_costs(None, 0, 0, 0, None, 0, None, None) == -1          # no samples per input ray
    assert L.ntr_bvh_leaf_depths(None, 64, None, 16, None, 1, None, None, None) == -1
    # group calls without a group
    assert L.ntr_dist_info(None, None, None) == -1 and "null group" in nt.lib().ntr_last_error().decode()
    assert L.ntr_dist_broadcast(None, None, 0, 0, None) == -1
    assert L.ntr_dist_gather_records(None, None, 0, 64, None, 0, None) == -1
    assert L.ntr_dist_destroy(None) == 0


def test_automatic_hint_table_is_per_device_and_starves_no_one():
    """the scheduling tables were process-global (96 automatic hints for ALL devices), so eight host threads with 17
    batches each (a 1080p frame: 1 primary + 16 AO) asked for 136 entries and the devices that came last ran without the learned
    dispatch order.  The tables are per device now; the table logic itself (no HIP) is driven here for 8 simulated devices x 17 keys:
    from the second round on every device finds all 17 of its batches.  One device alone holds 96 keys; with more batches than entries
    cycling through, least-recently-used replacement hints few or none of them -- never an error."""
    assert nt.selftest_auto_hint_table(8, 17, 3) == [17] * 8
    assert nt.selftest_auto_hint_table(8, 17, 1) == [0] * 8          # a batch seen once only registers
    assert nt.selftest_auto_hint_table(64, 96, 2) == [96] * 64
    got = nt.selftest_auto_hint_table(2, 120, 4)                      # more batches than entries
    assert len(got) == 2 and all(0 <= g <= 96 for g in got)
    L = nt.lib()
    assert L.ntr_selftest_auto_hint_table(0, 1, 1, None) == -1 and L.ntr_selftest_auto_hint_table(1, 1, 1, None) == -1
