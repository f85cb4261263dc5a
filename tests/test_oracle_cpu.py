"""Pins for the CPU oracle (the reference ships no golden vectors for this path and is not
buildable in this image -> 'parity unpinned'; these are the strongest pins available):
  * an independent numpy binary32 restatement (tests/np_tracer.py) agrees bit for bit,
  * closest-hit t equals brute force over all triangles bit for bit,
  * committed golden fixtures (tests/golden, made by tests/golden/make_golden.py).
Also checks the host SAH builder's structural invariants (config 1 plumbing)."""
import os

import numpy as np
import pytest

import ntrace_amd as nt
import np_tracer
from ntrace_amd import scenes
from oracle import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def soup():
    tri, pos, cam = scenes.random_soup(2500, seed=21)
    return nt.sah_build(tri, pos), cam, tri, pos


def edge_rays():
    from test_trace_gpu import edge_rays as er
    return er()


@pytest.mark.parametrize("any_hit", [False, True])
def test_oracle_vs_numpy_restatement(soup, any_hit):
    bvh, cam, _, _ = soup
    rays = np.concatenate([scenes.primary_rays(cam, 40, 40)[0], scenes.random_rays(1500, seed=2), edge_rays()])
    ref, st = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays, any_hit=any_hit)
    nid, nt_ = np_tracer.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays, any_hit=any_hit)
    assert np.array_equal(nid, ref["id"])
    assert np.array_equal(u32(nt_), u32(ref["t"]))
    assert st.numRays == rays.shape[0]


def test_oracle_closest_equals_bruteforce(soup):
    bvh, cam, _, _ = soup
    rays = np.concatenate([scenes.primary_rays(cam, 48, 48)[0], scenes.random_rays(1000, seed=4)])
    ref, _ = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays)
    bf = oracle.bruteforce_closest(bvh.woop, bvh.tri_index, rays)
    assert np.array_equal(u32(bf["t"]), u32(ref["t"]))
    assert np.array_equal(bf["id"] >= 0, ref["id"] >= 0)
    # ids may differ only on exact-t ties between triangles
    diff = bf["id"] != ref["id"]
    assert diff.mean() < 0.01


def test_multithreaded_oracle_matches_single(soup):
    bvh, cam, _, _ = soup
    rays = scenes.random_rays(5000, seed=9)
    a, sa = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays, threads=1)
    b, sb = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays, threads=5)
    assert np.array_equal(a, b)
    assert sa.as_dict() == sb.as_dict()


def test_miss_record_and_degenerate_rays(soup):
    bvh, _, _, _ = soup
    rays = np.zeros(3, dtype=nt.RAY_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = 1000.0, 1000.0, 1000.0
    rays["dx"] = 1.0
    rays["tmax"] = [7.5, -1.0, np.inf]
    ref, _ = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays)
    assert (ref["id"] == -1).all()
    assert np.array_equal(u32(ref["t"]), u32(rays["tmax"]))  # miss = (-1, ray.tmax), CudaBVH.cpp:273-274


def test_infinite_tmax_records_missed_test_at_flt_max():
    """updateHit re-tests the FW_F32_MAX 'miss' value (CudaBVH.cpp:1121-1124, 1200): with
    tmax = +inf the first *tested* triangle is recorded at t = FLT_MAX even if missed."""
    tri = np.array([[0, 1, 2]], dtype=np.int32)
    pos = np.array([[0, 0, 5], [1, 0, 5], [0, 1, 5], [10, 10, 5], [11, 10, 5], [10, 11, 5]], dtype=np.float32)
    tri = np.array([[0, 1, 2], [3, 4, 5]], dtype=np.int32)
    bvh = nt.sah_build(tri, pos)
    rays = np.zeros(1, dtype=nt.RAY_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = 0.9, 0.9, 0.0  # inside the bbox of tri 0, outside the triangle
    rays["dz"] = 1.0
    rays["tmax"] = np.inf
    ref, st = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays)
    assert st.numTriTests >= 1
    assert ref["id"][0] == 0 and ref["t"][0] == np.float32(3.4028234663852886e38)
    nid, ntt = np_tracer.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays)
    assert nid[0] == 0 and u32(ntt)[0] == u32(ref["t"])[0]


# ---- host SAH builder + Compact flatten (config 1: Cornell box plumbing) --------------------------
def walk_compact(bvh):
    nodes_f = bvh.nodes.view(np.float32)
    nodes_i = bvh.nodes.view(np.int32)
    woop_u = bvh.woop.view(np.uint32).reshape(-1, 4)
    leaves, inner, stack = [], 0, [0]
    while stack:
        ofs = stack.pop()
        inner += 1
        b = ofs // 4
        for c in (nodes_i[b + 12], nodes_i[b + 13]):
            if c >= 0:
                stack.append(int(c))
            else:
                a, ids = ~int(c), []
                while woop_u[a, 0] != 0x80000000:
                    ids.append(int(bvh.tri_index[a]))
                    a += 3
                assert (woop_u[a] == 0x80000000).all()
                leaves.append(ids)
    return inner, leaves


@pytest.mark.parametrize("scene", ["cornell", "soup"])
def test_sah_compact_invariants(scene):
    tri, pos, cam = scenes.cornell_box() if scene == "cornell" else scenes.random_soup(3000, seed=5)
    bvh = nt.sah_build(tri, pos, 1, 1)
    inner, leaves = walk_compact(bvh)
    assert inner * 64 == bvh.nodes.nbytes
    assert all(len(l) == 1 for l in leaves)              # leaf prefs (1,1), Renderer.cpp:89
    ids = sorted(i for l in leaves for i in l)
    assert ids == list(range(tri.shape[0]))              # every triangle in exactly one leaf
    assert inner == len(leaves) - 1
    assert bvh.woop.nbytes == (3 * tri.shape[0] + len(leaves)) * 16
    assert bvh.tri_index.shape[0] * 16 == bvh.woop.nbytes
    # child boxes contain their triangles: every primary ray's closest hit equals brute force
    rays, _ = scenes.primary_rays(cam, 64, 48)
    ref, _ = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays)
    bf = oracle.bruteforce_closest(bvh.woop, bvh.tri_index, rays)
    assert np.array_equal(u32(bf["t"]), u32(ref["t"]))


def test_cornell_config_cpu_plumbing():
    """BASELINE config 1: Cornell box, host SAH build + CPU primary trace, no GPU."""
    tri, pos, cam = scenes.cornell_box()
    bvh = nt.sah_build(tri, pos, 1, 1)
    rays, slot_to_pixel = scenes.primary_rays(cam, 192, 108)
    assert sorted(slot_to_pixel.tolist()) == list(range(192 * 108))
    ref, st = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays)
    assert 0.5 < (ref["id"] >= 0).mean() <= 1.0
    assert st.numInnerVisits > 0 and st.maxStackDepth < 100


def test_golden_fixtures():
    files = sorted(f for f in os.listdir(GOLD) if f.endswith(".npz"))
    assert files, "no golden fixtures committed"
    for f in files:
        g = np.load(os.path.join(GOLD, f))
        rays = g["rays"].view(nt.RAY_DTYPE).reshape(-1)
        for any_hit, key in ((False, "closest"), (True, "any")):
            ref, st = oracle.trace(g["nodes"], g["woop"], g["tri_index"], rays, any_hit=any_hit)
            exp = g["res_" + key].view(nt.RESULT_DTYPE).reshape(-1)
            assert np.array_equal(ref["id"], exp["id"]), f
            assert np.array_equal(u32(ref["t"]), u32(exp["t"])), f
            assert [st.numInnerVisits, st.numTriTests, st.numLeafVisits, st.numHits] == g["stats_" + key].tolist(), f
