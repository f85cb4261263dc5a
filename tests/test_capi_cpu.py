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
