"""CPU checks of the LBVH oracle itself (no GPU): Morton codes and the stable sort against numpy,
tree invariants, canonical hash independence from node numbering, traversal == brute force."""
import numpy as np

from ntrace_amd import scenes
from oracle import oracle


def np_morton(tri, pos):
    mn, mx = oracle.scene_bbox(pos)
    step = ((mx - mn) / np.float32(1024.0)).astype(np.float32)
    v = pos[tri]  # [n,3,3]
    lo, hi = v.min(axis=1), v.max(axis=1)
    mid = (lo + (hi - lo) / np.float32(2.0)).astype(np.float32)
    with np.errstate(all="ignore"):
        q = ((mid - mn) / step).astype(np.float32)
    cell = np.clip(np.floor(q).astype(np.int64), 0, 1023).astype(np.uint32)

    def spread(n):
        n = n & 0x3ff
        n = (n ^ (n << 16)) & 0xff0000ff
        n = (n ^ (n << 8)) & 0x0300f00f
        n = (n ^ (n << 4)) & 0x030c30c3
        return (n ^ (n << 2)) & 0x09249249
    return (spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)).astype(np.uint32)


def test_morton_and_stable_sort_match_numpy():
    tri, pos, _ = scenes.random_soup(4000, seed=3)
    b = oracle.lbvh_build(tri, pos)
    keys = np_morton(tri, pos)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(b["tri_sorted"], order.astype(np.int32))
    assert np.array_equal(b["morton_sorted"], keys[order])


def walk(nodes, woop, idx):
    ni = nodes.view(np.int32)
    nf = nodes.view(np.float32)
    w = woop.view(np.uint32).reshape(-1, 4)
    leaves, boxes, stack, inner = [], [], [0], 0
    while stack:
        ofs = stack.pop()
        inner += 1
        b = ofs // 4
        for k, c in enumerate((ni[b + 12], ni[b + 13])):
            lo = np.array([nf[b + 4 * k], nf[b + 4 * k + 2], nf[b + 8 + 2 * k]])
            hi = np.array([nf[b + 4 * k + 1], nf[b + 4 * k + 3], nf[b + 9 + 2 * k]])
            if c >= 0:
                assert c % 64 == 0
                stack.append(int(c))
                boxes.append((int(c), lo, hi))
            else:
                a, ids = ~int(c), []
                while w[a, 0] != 0x80000000:
                    ids.append(int(idx[a]))
                    a += 3
                leaves.append((ids, lo, hi))
    return inner, leaves, boxes


def test_lbvh_tree_invariants_and_leaf_boxes():
    tri, pos, cam = scenes.random_soup(3000, seed=8)
    eps = 0.001
    b = oracle.lbvh_build(tri, pos, 8, eps)
    inner, leaves, boxes = walk(b["nodes"], b["woop"], b["tri_index"])
    assert inner == b["num_inner"] and len(leaves) == b["num_leaves"]
    ids = [i for l in leaves for i in l[0]]
    assert sorted(ids) == list(range(tri.shape[0]))
    assert all(len(l[0]) <= 8 for l in leaves)
    for tids, lo, hi in leaves[:200]:
        v = pos[tri[tids]].reshape(-1, 3)
        assert np.array_equal(lo.astype(np.float32), (v.min(0) - np.float32(eps)).astype(np.float32))
        assert np.array_equal(hi.astype(np.float32), (v.max(0) + np.float32(eps)).astype(np.float32))
    # inner child box = union of the grandchildren boxes
    nf = b["nodes"].view(np.float32)
    for ofs, lo, hi in boxes[:300]:
        c = ofs // 4
        glo = np.minimum([nf[c], nf[c + 2], nf[c + 8]], [nf[c + 4], nf[c + 6], nf[c + 10]])
        ghi = np.maximum([nf[c + 1], nf[c + 3], nf[c + 9]], [nf[c + 5], nf[c + 7], nf[c + 11]])
        assert np.array_equal(glo.astype(np.float32), lo.astype(np.float32))
        assert np.array_equal(ghi.astype(np.float32), hi.astype(np.float32))
    rays = scenes.primary_rays(cam, 48, 48)[0]
    ref, _ = oracle.trace(b["nodes"], b["woop"], b["tri_index"], rays)
    bf = oracle.bruteforce_closest(b["woop"], b["tri_index"], rays)
    assert np.array_equal(ref["t"].view(np.uint32), bf["t"].view(np.uint32))


def test_canonical_hash_ignores_numbering_but_sees_content():
    tri, pos, _ = scenes.random_soup(500, seed=2)
    b = oracle.lbvh_build(tri, pos)
    h = oracle.bvh_canonical_hash(b["nodes"], b["woop"], b["tri_index"])
    # swap two non-root node slots and patch the pointers -> same canonical hash
    nodes = b["nodes"].copy().view(np.int32).reshape(-1, 16)
    a, c = 3, 7
    nodes[[a, c]] = nodes[[c, a]]
    ptr = nodes[:, 12:14]
    pa, pc = ptr == a * 64, ptr == c * 64
    ptr[pa] = c * 64
    ptr[pc] = a * 64
    assert oracle.bvh_canonical_hash(nodes.view(np.uint8).reshape(-1), b["woop"], b["tri_index"]) == h
    # flipping one box bit changes it
    bad = b["nodes"].copy()
    bad[5 * 64] ^= 1
    assert oracle.bvh_canonical_hash(bad, b["woop"], b["tri_index"]) != h
