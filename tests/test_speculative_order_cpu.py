"""The committed counter-example to speculative traversal (tests/golden/speculative_counterexample.npz, made by
tests/golden/make_spec_counterexample.py): on a three-triangle tree with two coplanar overlapping triangles, the reference's
speculative kernel ORDER (postponed leaves, fermi_speculative_while_while.cu:170-186; emulated with the CPU tracer's own
arithmetic in tests/spec_emulation.py) accepts triangle 1 at a t one ulp below the CPU tracer's record (triangle 0), because it
enters a box the CPU order culls against the shrunken ray.tmax.  That is why `fermi_speculative_while_while` and
`tesla_persistent_speculative_while_while` select non-speculative kernels here; tests/test_speculative_order_gpu.py asserts
that every shipped kernel returns the CPU record."""
import os

import numpy as np

import np_tracer
import ntrace_amd as nt
import spec_emulation
from oracle import oracle

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "speculative_counterexample.npz"))
RAYS = np.ascontiguousarray(G["rays"]).view(nt.RAY_DTYPE).reshape(-1)


def test_cpu_order_record_is_triangle_0():
    ref, st = oracle.trace(G["nodes"], G["woop"], G["tri_index"], RAYS)
    assert list(ref["id"]) == list(G["cpu_id"]) == [0, 2]
    assert np.array_equal(ref["t"].view(np.uint32), G["cpu_t_bits"])
    ids, ts = np_tracer.trace(G["nodes"], G["woop"], G["tri_index"], RAYS)
    assert np.array_equal(ids, G["cpu_id"]) and np.array_equal(ts.view(np.uint32), G["cpu_t_bits"])
    bvh = nt.host_bvh_wrap(G["nodes"], G["woop"], G["tri_index"])
    got, _, _ = bvh.host_trace(RAYS, False)
    assert np.array_equal(got["id"], G["cpu_id"]) and np.array_equal(got["t"].view(np.uint32), G["cpu_t_bits"])
    bvh.close()
    # the culled box: B's entry distance lies beyond the hit on A, although B's own Woop t lies before it
    entry = G["box_entry_B_bits"].view(np.float32)[0]
    assert G["spec_t_bits"].view(np.float32)[0] < G["cpu_t_bits"].view(np.float32)[0] < entry


def test_speculative_order_returns_another_record():
    sid, st = spec_emulation.trace_warp(G["nodes"], G["woop"], G["tri_index"], RAYS)
    assert list(sid) == list(G["spec_id"]) == [1, 2] and np.array_equal(st.view(np.uint32), G["spec_t_bits"])
    assert sid[0] != G["cpu_id"][0] and st.view(np.uint32)[0] != G["cpu_t_bits"][0]
    # with no second lane still searching the vote ends the speculation at once and the same kernel agrees with the CPU:
    # the speculative kernel's record depends on what the OTHER lanes of the warp are doing
    aid, at = spec_emulation.trace_warp(G["nodes"], G["woop"], G["tri_index"], RAYS[:1])
    assert aid[0] == G["cpu_id"][0] and at.view(np.uint32)[0] == G["cpu_t_bits"][0]
