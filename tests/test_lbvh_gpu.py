"""On-device LBVH build vs the CPU restatement of HLBVHBuilder::buildLBVH.

Node numbering / leaf placement depend on atomic order (in the reference too), so trees are
compared in canonical form: same topology, same split-axis words, same child boxes bit for
bit, same triangles per leaf in sorted order, same Woop rows bit for bit."""
import numpy as np
import pytest

import ntrace_amd as nt
from ntrace_amd import scenes
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "split16-64t", "split40-256t", "bottom-up-no-lds", "bottom-up-no-lds-split16", "bottom-up-staged", "bottom-up-staged-split16", "bottom-up-one-launch", "bottom-up-one-launch-split16"])
def build_path(request, monkeypatch):
    """Every test runs on the default path (one-sweep sort, bottom-up emit with scanned indices; scenes of at most `split`
    triangles are one subtree workgroup), on the same path with a tiny `split` (so that small scenes take the bottom-up emit, and
    runs of equal keys its slow path with hand-over roots and the oversize fallback), with every meeting through memory, and with
    the border chains in one launch / two launches.  (The superseded round-1 / round-2 paths are a patch under
    scripts/studies/rejected_patches/ and no longer part of any build.)"""
    request.addfinalizer(lambda: nt.set_tunables())
    for k in ("NTR_LBVH_SPLIT", "NTR_LBVH_SUB_THREADS", "NTR_LBVH_AGG_LDS", "NTR_LBVH_AGG_STAGED"):
        monkeypatch.delenv(k, raising=False)
    if request.param == "bottom-up-no-lds":  # every meeting of the bottom-up emit through memory
        monkeypatch.setenv("NTR_LBVH_AGG_LDS", "0")
    elif request.param == "bottom-up-no-lds-split16":
        monkeypatch.setenv("NTR_LBVH_AGG_LDS", "0")
        monkeypatch.setenv("NTR_LBVH_SPLIT", "16")
    elif request.param == "bottom-up-one-launch":  # border meetings through memory inside the tile kernel (default below 2^20 triangles)
        monkeypatch.setenv("NTR_LBVH_AGG_STAGED", "0")
    elif request.param == "bottom-up-one-launch-split16":
        monkeypatch.setenv("NTR_LBVH_AGG_STAGED", "0")
        monkeypatch.setenv("NTR_LBVH_SPLIT", "16")
    elif request.param == "bottom-up-staged":  # clusters that outgrow their tile go to ONE second launch, chains through memory
        monkeypatch.setenv("NTR_LBVH_AGG_STAGED", "1")
    elif request.param == "bottom-up-staged-split16":
        monkeypatch.setenv("NTR_LBVH_AGG_STAGED", "1")
        monkeypatch.setenv("NTR_LBVH_SPLIT", "16")
    elif request.param == "split16-64t":
        monkeypatch.setenv("NTR_LBVH_SPLIT", "16")
        monkeypatch.setenv("NTR_LBVH_SUB_THREADS", "64")
    elif request.param == "split40-256t":
        monkeypatch.setenv("NTR_LBVH_SPLIT", "40")
        monkeypatch.setenv("NTR_LBVH_SUB_THREADS", "256")
    nt.set_tunables()  # the library reads the environment once
    return request.param


def gpu_lbvh(tri, pos, leaf_size=8, epsilon=0.001):
    import torch
    from gpu_util import up
    tri = np.ascontiguousarray(tri, dtype=np.int32)
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    n = tri.shape[0]
    capn, capw, capi = nt.lbvh_capacity(n)
    d_tri, d_pos = up(tri), up(pos)
    d_nodes = torch.zeros(capn, dtype=torch.uint8, device="cuda:0")
    d_woop = torch.zeros(capw, dtype=torch.uint8, device="cuda:0")
    d_idx = torch.zeros(capi, dtype=torch.uint8, device="cuda:0")
    mn, mx = oracle.scene_bbox(pos)
    res = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, leaf_size, epsilon,
                        d_nodes.data_ptr(), capn, d_woop.data_ptr(), capw, d_idx.data_ptr(), capi)
    torch.cuda.synchronize()
    nodes = d_nodes.cpu().numpy()[:res.nodesBytes].copy()
    woop = d_woop.cpu().numpy()[:res.triWoopBytes].copy()
    idx = d_idx.cpu().numpy()[:res.triIndexBytes].view(np.int32).copy()
    return nodes, woop, idx, res, (d_nodes, d_woop, d_idx)


def check_against_oracle(tri, pos, leaf_size=8, epsilon=0.001):
    """Counts, exact extents (the reference's sizes, HLBVHBuilder.cpp:382-386: the leaf marks follow the depth rule too, so no slot is
    ever set aside and left unused) and the canonical tree."""
    nodes, woop, idx, res, keep = gpu_lbvh(tri, pos, leaf_size, epsilon)
    ref = oracle.lbvh_build(tri, pos, leaf_size, epsilon)
    assert res.numNodes == ref["num_inner"] and res.numLeaves == ref["num_leaves"] and res.numLevels == ref["num_levels"]
    assert nodes.nbytes == ref["nodes"].nbytes and woop.nbytes == ref["woop"].nbytes and idx.nbytes == ref["tri_index"].nbytes
    assert oracle.bvh_canonical_hash(nodes, woop, idx) == oracle.bvh_canonical_hash(ref["nodes"], ref["woop"], ref["tri_index"])
    return nodes, woop, idx, res, ref, keep


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (7, 3), (9, 4), (64, 5), (1000, 6), (5000, 7), (70000, 8)])
def test_lbvh_matches_oracle_random_soups(n, seed):
    tri, pos, _ = scenes.random_soup(n, seed=seed, walls=(n > 100))
    check_against_oracle(tri, pos)


@pytest.mark.parametrize("leaf_size", [1, 2, 8, 16, 32, 33, 64])   # more than 32 triangles per leaf: the top-down path
def test_lbvh_leaf_sizes(leaf_size):
    tri, pos, _ = scenes.random_soup(3000, seed=21)
    check_against_oracle(tri, pos, leaf_size=leaf_size)
    tri, pos, _ = scenes.random_soup(9000, seed=22, walls=True)   # more triangles than one subtree workgroup takes
    check_against_oracle(tri, pos, leaf_size=leaf_size)


def test_lbvh_duplicate_morton_codes_median_split_and_depth_limit():
    """Many triangles in one Morton cell: median splits (emitTreeKernel.cu:282) and the forced
    leaves of the last level (oldLevel == 0, :289-292) -> leaves larger than leafSize."""
    rng = np.random.default_rng(5)
    n = 6000
    c = np.array([0.3, 0.4, 0.5]) + rng.normal(0, 1e-6, size=(n, 1, 3))
    p = (c + rng.normal(0, 1e-7, size=(n, 3, 3))).reshape(-1, 3)
    far = np.array([[-10, -10, -10], [10, -10, -10], [-10, 10, 10], [10, 10, 10], [9, 10, 10], [10, 9, 10]], dtype=np.float64)
    pos = np.concatenate([p, far]).astype(np.float32)
    tri = np.concatenate([np.arange(n * 3).reshape(-1, 3), np.array([[n * 3, n * 3 + 1, n * 3 + 2], [n * 3 + 3, n * 3 + 4, n * 3 + 5]])]).astype(np.int32)
    nodes, woop, idx, res, ref, _ = check_against_oracle(tri, pos, leaf_size=2)
    assert res.numLevels >= 12  # ~log2(6000 / 2) median levels below the split that isolates the cluster


@pytest.mark.parametrize("n", [9, 700, 5000])
def test_lbvh_all_codes_equal(n):
    """Every triangle in the same Morton cell (n copies of one triangle): the whole tree is the reference's median subtree, rooted at
    node 0 (emitTreeKernel.cu:282; the bottom-up path's root run)."""
    one = np.array([[0.25, 0.5, 0.75], [0.5, 0.5, 0.75], [0.25, 0.75, 0.8]], dtype=np.float32)
    pos = np.tile(one, (n, 1))
    tri = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    nodes, woop, idx, res, ref, _ = check_against_oracle(tri, pos, leaf_size=4)
    assert res.numLeaves >= n // 4


def test_lbvh_depth_limit_forces_oversized_leaves():
    """A chain of Morton codes 0 (x10), 2^0, 2^1, ..., 2^29: one level per bit, and at the last
    level (kernel bit 0, `oldLevel == 0`, emitTreeKernel.cu:289-292) the 10 duplicates become one
    leaf although leafSize is 2."""
    cells = [(0, 0, 0)] * 10
    for k in range(30):
        c = [0, 0, 0]
        c[k % 3] = 1 << (k // 3)
        cells.append(tuple(c))
    ctr = (np.array(cells, dtype=np.float64) + 0.5)
    d = 0.05
    p = np.stack([ctr + [-d, -d, 0], ctr + [d, -d, 0], ctr + [0, d, 0]], axis=1).reshape(-1, 3)
    corner = np.array([[0, 0, 0], [0.01, 0, 0], [0, 0.01, 0], [1024, 1024, 1024], [1023.99, 1024, 1024], [1024, 1023.99, 1024]])
    pos = np.concatenate([p, corner]).astype(np.float32)
    nt_ = len(cells)
    tri = np.concatenate([np.arange(nt_ * 3).reshape(-1, 3), [[nt_ * 3, nt_ * 3 + 1, nt_ * 3 + 2], [nt_ * 3 + 3, nt_ * 3 + 4, nt_ * 3 + 5]]]).astype(np.int32)
    nodes, woop, idx, res, ref, _ = check_against_oracle(tri, pos, leaf_size=2)
    assert res.numLevels == 30
    # the bvhcache file written from this tree has the reference's size: S32 layout + 3 x (S64 size + bytes), exact extents
    assert 4 + 3 * 8 + nodes.nbytes + woop.nbytes + idx.nbytes == 4 + 3 * 8 + ref["nodes"].nbytes + ref["woop"].nbytes + ref["tri_index"].nbytes
    w = woop.view(np.uint32).reshape(-1, 4)
    sizes, a, cur = [], 0, 0
    while a < w.shape[0]:
        if w[a, 0] == 0x80000000:
            sizes.append(cur); cur = 0; a += 1
        else:
            cur += 1; a += 3
    assert max(sizes) >= 10


def _cell_triangles(cells):
    ctr = (np.array(cells, dtype=np.float64) + 0.5)
    d = 0.05
    p = np.stack([ctr + [-d, -d, 0], ctr + [d, -d, 0], ctr + [0, d, 0]], axis=1).reshape(-1, 3)
    corner = np.array([[0, 0, 0], [0.01, 0, 0], [0, 0.01, 0], [1024, 1024, 1024], [1023.99, 1024, 1024], [1024, 1023.99, 1024]])
    pos = np.concatenate([p, corner]).astype(np.float32)
    nt_ = len(cells)
    tri = np.concatenate([np.arange(nt_ * 3).reshape(-1, 3), [[nt_ * 3, nt_ * 3 + 1, nt_ * 3 + 2], [nt_ * 3 + 3, nt_ * 3 + 4, nt_ * 3 + 5]]]).astype(np.int32)
    return tri, pos


@pytest.mark.parametrize("depth,dups,leaf", [(30, 10, 2), (30, 700, 8), (29, 37, 2), (29, 1500, 8), (28, 200, 2), (27, 70, 4), (26, 3000, 8), (12, 5000, 2)])
def test_lbvh_depth_rule_at_every_depth_with_exact_extents(depth, dups, leaf):
    """The leaf marks follow the depth rule themselves (round 5: no relocation pass).  `dups` triangles in Morton cell 0 below a chain of
    `depth` ancestors (one code 2^b per ancestor, the highest bits): the run's median subtree may use the levels depth ... 29 only
    (emitTreeKernel.cu:289-292) -- a run at depth 30 is ONE leaf, at depth 29 two leaves, and so on, whatever leafSize says.  Runs
    longer than a wave, than a 512-key tile and than a 1024-key mark block; counts, exact extents and the canonical tree equal the
    oracle's on every build path."""
    cells = [(0, 0, 0)] * dups
    for k in range(30 - depth, 30):
        c = [0, 0, 0]
        c[k % 3] = 1 << (k // 3)
        cells.append(tuple(c))
    # a second, shallower run elsewhere (its subtree is not cut short), and a few loose triangles
    far = [0, 0, 0]
    far[29 % 3] = (1 << (29 // 3)) | 3
    cells += [tuple(far)] * (3 * leaf + 1)
    tri, pos = _cell_triangles(cells)
    nodes, woop, idx, res, ref, _ = check_against_oracle(tri, pos, leaf_size=leaf)
    w = woop.view(np.uint32).reshape(-1, 4)
    sizes, a, cur = [], 0, 0
    while a < w.shape[0]:
        if w[a, 0] == 0x80000000:
            sizes.append(cur); cur = 0; a += 1
        else:
            cur += 1; a += 3
    levels_left = 30 - depth
    want_big = dups > leaf * (1 << levels_left)          # the run cannot be cut down to leafSize in the levels it has left
    assert (max(sizes) > leaf) == want_big


def test_ray_splitting_on_coincident_triangles_in_a_deep_tree(monkeypatch):
    """The hardest case for the persistent kernels' ray splitting (csrc/trace_split.h): 3 000 IDENTICAL triangles in one Morton cell
    below a chain of 18 ancestors.  Every ray that hits one of them hits all of them at the same t -- the record is the first one in
    visiting order, nothing else distinguishes them --, every box of the run's median subtree is the same box, so both children are
    entered at every level and the stacks run 12 entries deep over the chain.  Helpers' hits can only ever be valid under the exact
    bound of the lone ray here; records must be the oracle's with the lanes looked at after every step, every 3 / 8 steps and never."""
    import torch
    from gpu_util import assert_parity, up
    depth, dups, leaf = 18, 3000, 2
    cells = [(0, 0, 0)] * dups
    for k in range(30 - depth, 30):
        c = [0, 0, 0]
        c[k % 3] = 1 << (k // 3)
        cells.append(tuple(c))
    tri, pos = _cell_triangles(cells)
    nodes, woop, idx, res, ref, (d_nodes, d_woop, d_idx) = check_against_oracle(tri, pos, leaf_size=leaf)
    view = nt.BvhView(d_nodes.data_ptr(), res.nodesBytes, d_woop.data_ptr(), res.triWoopBytes, d_idx.data_ptr())
    view.validate()
    rng = np.random.default_rng(77)
    n = 24000
    rays = np.zeros(n, dtype=nt.RAY_DTYPE)
    target = np.array([0.5, 0.5, 0.5]) + rng.uniform(-0.06, 0.06, size=(n, 3)) * [1, 1, 0]     # points on and just around the triangles
    origin = target + rng.normal(size=(n, 3)) * rng.choice([0.3, 5.0, 300.0], size=(n, 1))
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for k, a in zip(("ox", "oy", "oz"), origin.T):
        rays[k] = a.astype(np.float32)
    for k, a in zip(("dx", "dy", "dz"), d.T):
        rays[k] = a.astype(np.float32)
    rays["tmin"] = 0.0
    rays["tmax"] = 4000.0
    exp, _ = oracle.trace(nodes, woop, idx, rays, any_hit=False, threads=8)
    assert (exp["id"] >= 0).sum() > n // 8
    d_rays = up(rays)
    try:
        for slice_ in ("1", "3", "8", "0"):
            monkeypatch.setenv("NTR_TRACE_SPLIT_SLICE", slice_)
            nt.set_tunables()
            for m in (n, 4097, 65):
                d_res = torch.zeros(m * 16, dtype=torch.uint8, device="cuda:0")
                view.trace("kepler_dynamic_fetch", m, False, d_rays.data_ptr(), d_res.data_ptr())
                torch.cuda.synchronize()
                assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), exp[:m], "coincident triangles, split slice %s, n=%d" % (slice_, m))
        assert nt.trace_status() == 0
    finally:
        monkeypatch.delenv("NTR_TRACE_SPLIT_SLICE", raising=False)
        nt.set_tunables()


def test_trace_on_gpu_built_lbvh_matches_oracle_trace():
    """Parity hazard 8: Woop data from the GPU builder differs from the host
    woopifyTri, so parity is defined on the downloaded GPU-built buffers fed to both tracers."""
    import torch
    from gpu_util import assert_parity, up
    tri, pos, cam = scenes.random_soup(30000, seed=31)
    nodes, woop, idx, res, ref, (d_nodes, d_woop, d_idx) = check_against_oracle(tri, pos)
    view = nt.BvhView(d_nodes.data_ptr(), res.nodesBytes, d_woop.data_ptr(), res.triWoopBytes, d_idx.data_ptr())
    view.validate()
    rays = np.concatenate([scenes.primary_rays(cam, 256, 192)[0], scenes.random_rays(40000, seed=9)])
    d_rays = up(rays)
    for kernel in ("fermi_speculative_while_while", "kepler_dynamic_fetch"):
        for any_hit in (False, True):
            d_res = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device="cuda:0")
            view.trace(kernel, rays.shape[0], any_hit, d_rays.data_ptr(), d_res.data_ptr())
            got = d_res.cpu().numpy().view(nt.RESULT_DTYPE)
            exp, _ = oracle.trace(nodes, woop, idx, rays, any_hit=any_hit, threads=8)
            assert_parity(got, exp, "lbvh %s anyHit=%d" % (kernel, any_hit))


def test_lbvh_atrium_262k_every_triangle_once():
    tri, pos, _ = scenes.atrium()
    nodes, woop, idx, res, ref, _ = check_against_oracle(tri, pos)
    assert res.seconds > 0
    # walk the tree from the root (storage the depth rule left unused inside a large leaf is not reachable)
    w = woop.view(np.uint32).reshape(-1, 4)
    ni = nodes.view(np.int32).reshape(-1, 16)
    ids, stack = [], [0]
    while stack:
        node = ni[stack.pop()]
        for child in (int(node[12]), int(node[13])):
            if child >= 0:
                stack.append(child // 64)
                continue
            a = ~child
            while w[a, 0] != 0x80000000:
                ids.append(idx[a])
                a += 3
    assert np.array_equal(np.sort(np.array(ids)), np.arange(tri.shape[0]))


def test_lbvh_hand_derived_known_answer_tree():
    """tests/kat_lbvh.py: the tree worked out on paper from the reference's expressions (clamped Morton cell, stable sort of a scrambled
    input, a 30-level chain ending in the level-0 node with its oversize leaf, median splits of equal codes, epsilon boxes, Woop rows with
    their signed zeros), on every device build path: counts, exact extents, and every word of the tree."""
    import kat_lbvh as kl
    tri, pos = kl.scene()
    nodes, woop, idx, res, keep = gpu_lbvh(tri, pos, kl.LEAF_SIZE, kl.EPSILON)
    assert (res.numNodes, res.numLeaves) == (kl.NUM_INNER, kl.NUM_LEAVES)
    assert nodes.nbytes == 64 * kl.NUM_INNER and woop.nbytes == 16 * (3 * 39 + kl.NUM_LEAVES) and idx.nbytes == 4 * (3 * 39 + kl.NUM_LEAVES)
    assert kl.compare(nodes, woop, idx, "device") == (kl.NUM_INNER, kl.NUM_LEAVES)


def test_lbvh_hand_derived_tree_inside_a_large_scene():
    """The same 39 triangles with 20 000 filler triangles in cells of the upper octant beside them (codes between 1 << 29 and
    0x3FFFFFFF do not exist in the vector): the scene takes the bottom-up path with many tiles, and the vector's left half -- the 30-level
    chain with its level-0 node -- must come out of it unchanged, as the subtree under the root's child 0."""
    import kat_lbvh as kl
    tri, pos = kl.scene()
    rng = np.random.default_rng(5)
    n_f = 20000
    # filler: cells with z in [513, 1023): codes above 1 << 29 (H's cell has x = y = 0, z = 512: codes of the filler with z >= 513 sort after H)
    c = np.stack([rng.integers(0, 1023, n_f), rng.integers(0, 1023, n_f), rng.integers(513, 1023, n_f)], axis=1).astype(np.float32)
    v2 = c + np.float32(0.25)
    fpos = np.zeros((n_f * 3, 3), dtype=np.float32)
    fpos[0::3] = v2 + np.array([0.5, 0, 0], dtype=np.float32)
    fpos[1::3] = v2 + np.array([0, 0.5, 0], dtype=np.float32)
    fpos[2::3] = v2
    ftri = (np.arange(n_f * 3, dtype=np.int32).reshape(n_f, 3) + pos.shape[0])
    tri2 = np.concatenate([tri, ftri]).astype(np.int32)
    pos2 = np.concatenate([pos, fpos]).astype(np.float32)
    nodes, woop, idx, res, ref, keep = check_against_oracle(tri2, pos2, kl.LEAF_SIZE, kl.EPSILON)
    # the root still splits at bit 29 (position 33), and its child 0 is the derived left subtree
    ni = nodes.view(np.int32)
    assert int(ni[14]) == 29 % 3 and int(ni[12]) >= 0
    exp_left = kl.expected_tree()["children"][0]

    def walk(ofs, exp):
        w = ofs // 4
        assert int(ni[w + 14]) == exp["word14"]
        for k in (0, 1):
            child, ce = int(ni[w + 12 + k]), exp["children"][k]
            if "leaf" in ce:
                a = ~child
                assert child < 0 and [int(idx[a + 3 * j]) for j in range(len(ce["leaf"]))] == ce["leaf"]
            else:
                assert child >= 0
                walk(child, ce)
    walk(int(ni[12]), exp_left)
