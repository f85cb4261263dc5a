"""Hand-derived known-answer vectors (tests/kat_vectors.py) against the three CPU implementations: the C oracle,
the numpy restatement, and the mirror's host tracer (CudaAS::trace).  The expected values are literals worked out
from the reference's source expressions, not the output of any of them."""
import numpy as np
import pytest

import kat_vectors as kat
import np_tracer
import ntrace_amd as nt
from oracle import oracle

CASES = kat.cases()


def _check(name, ids, ts, exp, key):
    for i, e in enumerate(exp):
        want_id, want_bits = e[key]
        assert int(ids[i]) == want_id and int(np.asarray(ts[i], dtype=np.float32).view(np.uint32)) == want_bits, \
            "%s ray %d (%s): got (%d, 0x%08X), derived (%d, 0x%08X)" % (name, i, key, int(ids[i]), int(np.asarray(ts[i], dtype=np.float32).view(np.uint32)), want_id, want_bits)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_known_answers_oracle(case):
    name, nodes, woop, idx, rays, exp = case
    for any_hit, key in ((False, "closest"), (True, "any")):
        ref, _ = oracle.trace(nodes, woop, idx, rays, any_hit=any_hit)
        _check(name + "/oracle", ref["id"], ref["t"], exp, key)
    for i, e in enumerate(exp):  # counters decide the accept-rule cases (same record either way)
        if e["inner"] is None:
            continue
        _, st = oracle.trace(nodes, woop, idx, rays[i:i + 1], any_hit=False)
        assert (st.numInnerVisits, st.numTriTests) == (e["inner"], e["tris"]), (name, i, st.as_dict())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_known_answers_numpy_restatement(case):
    name, nodes, woop, idx, rays, exp = case
    for any_hit, key in ((False, "closest"), (True, "any")):
        ids, ts = np_tracer.trace(nodes, woop, idx, rays, any_hit=any_hit)
        _check(name + "/numpy", ids, ts, exp, key)
    for i, e in enumerate(exp):
        if e["inner"] is None:
            continue
        _, _, st = np_tracer.trace(nodes, woop, idx, rays[i:i + 1], any_hit=False, return_stats=True)
        assert (st["numInnerVisits"], st["numTriTests"]) == (e["inner"], e["tris"]), (name, i, st)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_known_answers_host_tracer(case):
    name, nodes, woop, idx, rays, exp = case
    bvh = nt.host_bvh_wrap(nodes, woop, idx)
    for any_hit, key in ((False, "closest"), (True, "any")):
        got, _, _ = bvh.host_trace(rays, any_hit)
        _check(name + "/host", got["id"], got["t"], exp, key)
    for i, e in enumerate(exp):
        if e["inner"] is None:
            continue
        _, _, st = bvh.host_trace(rays[i:i + 1], False, want_stats=True)
        assert (st.numInnerVisits, st.numTriTests) == (e["inner"], e["tris"]), (name, i, st.as_dict())
    bvh.close()


def test_hand_written_woop_rows_are_what_the_builder_produces():
    """The closed-form Woop rows the vectors use are the values of CudaBVH::woopifyTri (CudaBVH.cpp:668-687) on the same triangles."""
    pos = np.array([[2, 0, 4], [0, 2, 4], [0, 0, 4], [2.5, 0, 8], [0.5, 2, 8], [0.5, 0, 8]], dtype=np.float32)
    tri = np.array([[0, 1, 2], [3, 4, 5]], dtype=np.int32)
    bvh = nt.sah_build(tri, pos, 1, 1)
    woop = bvh.woop.view(np.float32).reshape(-1, 4)
    for t, corner in ((0, (0, 0, 4)), (1, (0.5, 0, 8))):
        # locate the triangle's rows through triIndex (one triangle per leaf: rows at float4 index of its id)
        where = [a for a in range(0, woop.shape[0], 4) if bvh.tri_index[a] == t and woop.view(np.uint32)[a, 0] != 0x80000000]
        assert where
        rows = np.array(kat.tri_rows(*corner), dtype=np.float32)
        # equal as values: the cofactor inverse produces some zeros with a minus sign, which no product with a finite
        # ray component can turn into a different t, u or v
        assert np.array_equal(woop[where[0]:where[0] + 3], rows), (t, woop[where[0]:where[0] + 3], rows)


def test_derived_constants():
    assert kat.bits(1.0 / 3.0) == 0x3EAAAAAB
    assert kat.bits(float(np.float32(5.0) * np.float32(kat.from_bits(0x3EAAAAAB)))) == 0x3FD55556
    assert kat.bits(float(np.float32(5.0) / np.float32(3.0))) == 0x3FD55555
    assert kat.bits(3.4028234663852886e38) == kat.FLT_MAX_BITS


def test_lbvh_known_answer_tree_oracle():
    """The hand-derived LBVH vector (tests/kat_lbvh.py: Morton codes with a clamped cell, the stable sort of a scrambled input, a 30-level
    chain down to the level-0 node whose children are leaves whatever their size, two median splits, epsilon boxes, Woop rows with their
    signed zeros) against the C oracle: codes, sorted order, counts, extents, and the whole tree bit for bit."""
    import kat_lbvh as kl
    tri, pos = kl.scene()
    mn, mx = oracle.scene_bbox(pos)
    assert tuple(float(x) for x in mn) == kl.SCENE_MIN and tuple(float(x) for x in mx) == kl.SCENE_MAX   # the vertex box IS the derived scene box
    ref = oracle.lbvh_build(tri, pos, kl.LEAF_SIZE, kl.EPSILON)
    assert [int(x) for x in ref["morton_sorted"]] == kl.CODES_SORTED
    assert [int(x) for x in ref["tri_sorted"]] == kl.ORIG_OF_SORTED
    assert (ref["num_inner"], ref["num_leaves"]) == (kl.NUM_INNER, kl.NUM_LEAVES)
    assert ref["nodes"].nbytes == 64 * kl.NUM_INNER and ref["woop"].nbytes == 16 * (3 * 39 + kl.NUM_LEAVES) and ref["tri_index"].nbytes == 4 * (3 * 39 + kl.NUM_LEAVES)
    assert kl.compare(ref["nodes"], ref["woop"], ref["tri_index"], "oracle") == (kl.NUM_INNER, kl.NUM_LEAVES)
