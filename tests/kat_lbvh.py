"""Hand-derived known-answer vector for the LBVH half of the path (HLBVHBuilder::buildLBVH), the counterpart of kat_vectors.py.

Nothing here is produced by an implementation under test: the scene is written down in exact binary fractions and the expected tree --
topology, split words, leaf contents in sorted order, child boxes, Woop rows, all bit for bit -- is worked out below from the reference's
source expressions.  Checked by tests/test_kat_cpu.py (C oracle) and tests/test_kat_gpu.py (every device build path).

THE SCENE (39 triangles, leafSize = 2, epsilon = 0.125; scene box = vertex box = [0, 1024]^3, so step = (max - min) / 1024 = 1 exactly,
HLBVHBuilder.cpp:76-81).  Every triangle is a right triangle with corner v2 and legs of 1/2 along +x and +y: v0 = v2 + (1/2, 0, 0),
v1 = v2 + (0, 1/2, 0).  Its box is [v2, v2 + (1/2, 1/2, 0)], the midpoint lo + (hi - lo) / 2 = v2 + (1/4, 1/4, 0), the Morton cell
floor(midpoint) clamped to [0, 1023] (emitTreeKernel.cu:680-686), the code spread(x) | spread(y) << 1 | spread(z) << 2 (:688): bit 3k of
the code is bit k of x, bit 3k + 1 of y, bit 3k + 2 of z.

  G   five triangles in cell (0, 0, 0), code 0: T_lo with v2 = (0, 0, 0) (pins the scene box's minimum) and four with
      v2 = (1/4, 1/4, z), z = 1/4, 3/8, 1/2, 5/8;
  C_b one triangle per bit b = 1 .. 28 with code 1 << b: v2 = (1/4, 1/4, 1/4) + e_axis * 2^(b div 3), axis = b mod 3;
  H   five triangles in cell (0, 0, 512), code 1 << 29: v2 = (1/4, 1/4, 512 + z), z = 1/4, 3/8, 1/2, 5/8, 3/4;
  T_hi v2 = (1023.5, 1023.5, 1024): midpoint (1023.75, 1023.75, 1024), cell z = 1024 CLAMPED to 1023, code 0x3FFFFFFF; pins the maximum.

The triangles are listed in a scrambled order (original index = position in TRIS); the sort is stable on (code, original index)
(radixSortCuda over idx[i] = i, emitTreeKernel.cu:690), so the sorted order is: G (by original index), C_1 .. C_28, H (by original index),
T_hi -- positions 0-4, 5-32, 33-37, 38.

THE TREE (emitTreeKernel.cu:233-381; the host launches depth d with level = 29 - d, HLBVHBuilder.cpp:337-361).  A node over sorted
positions [s, e) splits at the highest bit <= level in which its first and last codes differ, at the first position whose code has that
bit set (:262-279); with no such bit at the median (s + e) >> 1 (:282); a child is a leaf when it holds <= leafSize triangles OR the
node's level is 0 (:289-292, :338, :358); word 14 of the node is (the bit found) % 3, and -1 when none was found ((-1) % 3 in C, :379).

  root    [0, 39)  level 29: codes 0 .. 0x3FFFFFFF differ in bit 29, first set at position 33 -> [0, 33) | [33, 39), word 14 = 29 % 3 = 2
  L_d     d = 1 .. 28: [0, 34 - d) at level 29 - d: last code 1 << (29 - d) -> splits off the LEAF {C_(29-d)} at position 33 - d,
          word 14 = (29 - d) % 3; the left child [0, 33 - d) is L_(d+1)
  L_29    = G = [0, 5) at depth 29, level 0: all codes 0, no bit found -> median (0 + 5) >> 1 = 2, word 14 = -1, and BOTH children are
          leaves because the level is 0: [0, 2) and the OVERSIZE leaf [2, 5) of three triangles > leafSize (the depth rule)
  R_1     [33, 39) level 28: codes 1 << 29 .. 0x3FFFFFFF differ in bit 28, first set at 38 -> [33, 38) | leaf {T_hi}, word 14 = 1
  R_2     = H = [33, 38) level 27: equal codes -> median (33 + 38) >> 1 = 35: leaf [33, 35) | [35, 38) (three > leafSize, level != 0: a node)
  R_3     [35, 38) level 26: equal codes -> median 36: leaf [35, 36) | leaf [36, 38); word 14 = -1 for both median nodes
  33 inner nodes, 34 leaves, 39 triangles: nodes 33 * 64 B, Woop (3 * 39 + 34) * 16 B, index (3 * 39 + 34) * 4 B (HLBVHBuilder.cpp:382-386).

BOXES (calcLeaf, emitTreeKernel.cu:383-408): a leaf's box is the fminf / fmaxf fold over its triangles of min(a, b, c) - epsilon and
max(a, b, c) + epsilon, i.e. [v2 - 1/8, v2 + (5/8, 5/8, 1/8)] per triangle; an inner child's box is the union of its two child boxes
(calcAABB :417-562).  Everything is exact in binary32.

WOOP ROWS (calcWoop, emitTreeKernel.cu:574-635) of such a triangle, signed zeros included: c0 = (1/2, 0, 0), c1 = (0, 1/2, 0),
c2 = c0 x c1 = (+0, +0, 1/4); the determinant expression is 1/2 * (1/4 * 1/2) = 1/16, det = float(1.0 / (1/16)) = 16 (:589);
  i0 = ((1/8) * 16, -(+0) * 16, (+0) * 16) = (2, -0, +0);  i1 = (-(+0) * 16, (1/8) * 16, -(+0) * 16) = (-0, 2, -0);
  i2 = ((+0) * 16, -(+0) * 16, (1/4) * 16) = (+0, -0, 4)
  row Z = (i2, -fdot(-i2, v2)) = (+0, -0, 4, 4 z2)     row U = (i0, fdot(-i0, v2)) = (2, -0, +0, -2 x2)     row V = (i1, fdot(-i1, v2)) = (-0, 2, -0, -2 y2)
(fdot = a.x * b.x + a.y * b.y + a.z * b.z, :36-38; for x2, y2, z2 > 0 the zero products vanish against the non-zero term).  T_lo has
v2 = (0, 0, 0): fdot(-i2, v2) = (-0 * 0) + (+0 * 0) + (-4 * 0) = (-0 + +0) + -0 = +0 + -0 = +0, so Z.w = -(+0) = -0;
fdot(-i0, v2) = (-2 * 0) + (+0 * 0) + (-0 * 0) = (-0 + +0) + -0 = +0 = U.w; likewise V.w = +0.  Row Z's x is +0 in every triangle, so the
"-0 -> +0" normalisation (:621-622) changes nothing."""
import struct

import numpy as np

LEAF_SIZE = 2
EPSILON = 0.125
SCENE_MIN = (0.0, 0.0, 0.0)
SCENE_MAX = (1024.0, 1024.0, 1024.0)
NEG0 = struct.unpack("<f", struct.pack("<I", 0x80000000))[0]


def _corner_of_bit(b):
    c = [0.25, 0.25, 0.25]
    c[b % 3] += float(1 << (b // 3))
    return tuple(c)


# name -> corner v2, in SORTED order (the order derived above)
SORTED = ([("T_lo", (0.0, 0.0, 0.0))] + [("G%d" % k, (0.25, 0.25, z)) for k, z in enumerate((0.25, 0.375, 0.5, 0.625))] +
          [("C%d" % b, _corner_of_bit(b)) for b in range(1, 29)] +
          [("H%d" % k, (0.25, 0.25, 512.0 + z)) for k, z in enumerate((0.25, 0.375, 0.5, 0.625, 0.75))] +
          [("T_hi", (1023.5, 1023.5, 1024.0))])
assert len(SORTED) == 39
CODES_SORTED = [0] * 5 + [1 << b for b in range(1, 29)] + [1 << 29] * 5 + [0x3FFFFFFF]

# The order the triangles are handed to the builder in: scrambled, but ties of the sort (equal codes) keep their relative order, so the
# members of G and of H must appear in their listed order.  Original index = position in this list.
_ORDER = ([38] + list(range(32, 4, -1)) +            # T_hi first, then C_28 .. C_1
          [33, 0, 34, 1, 35, 2, 36, 3, 37, 4])        # H and G interleaved, each in its own order
assert sorted(_ORDER) == list(range(39))
TRIS = [SORTED[p] for p in _ORDER]                   # (name, v2) by original index
ORIG_OF_SORTED = [_ORDER.index(p) for p in range(39)]   # sorted position -> original index


def scene():
    """(tri int32 [39, 3], pos float32 [117, 3]): three private vertices per triangle, (v0, v1, v2) as the reference reads them."""
    pos = np.zeros((39 * 3, 3), dtype=np.float32)
    tri = np.arange(39 * 3, dtype=np.int32).reshape(39, 3)
    for i, (_, v2) in enumerate(TRIS):
        pos[3 * i + 0] = (v2[0] + 0.5, v2[1], v2[2])
        pos[3 * i + 1] = (v2[0], v2[1] + 0.5, v2[2])
        pos[3 * i + 2] = v2
    return tri, pos


def woop_rows(v2):
    x2, y2, z2 = v2
    if v2 == (0.0, 0.0, 0.0):
        return [(0.0, NEG0, 4.0, NEG0), (2.0, NEG0, 0.0, 0.0), (NEG0, 2.0, NEG0, 0.0)]
    return [(0.0, NEG0, 4.0, 4.0 * z2), (2.0, NEG0, 0.0, -2.0 * x2), (NEG0, 2.0, NEG0, -2.0 * y2)]


def tri_box(v2):
    """(lo.x, hi.x, lo.y, hi.y, lo.z, hi.z) of one triangle, epsilon-inflated."""
    return (v2[0] - 0.125, v2[0] + 0.625, v2[1] - 0.125, v2[1] + 0.625, v2[2] - 0.125, v2[2] + 0.125)


def union(a, b):
    return (min(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), max(a[3], b[3]), min(a[4], b[4]), max(a[5], b[5]))


def leaf(s, e):
    box = tri_box(SORTED[s][1])
    for p in range(s + 1, e):
        box = union(box, tri_box(SORTED[p][1]))
    return {"leaf": [ORIG_OF_SORTED[p] for p in range(s, e)], "rows": [woop_rows(SORTED[p][1]) for p in range(s, e)], "box": box}


def node(word14, c0, c1):
    return {"word14": word14, "children": (c0, c1), "box": union(c0["box"], c1["box"])}


def expected_tree():
    """The tree derived in the module docstring, as nested dicts (children in the reference's order: child 0 = the lower positions)."""
    g = node(-1, leaf(0, 2), leaf(2, 5))                       # L_29: level 0, median, oversize right leaf
    left = g
    for d in range(28, 0, -1):                                 # L_28 .. L_1, built from the bottom up
        left = node((29 - d) % 3, left, leaf(33 - d, 34 - d))
    r3 = node(-1, leaf(35, 36), leaf(36, 38))
    r2 = node(-1, leaf(33, 35), r3)
    r1 = node(28 % 3, r2, leaf(38, 39))
    return node(29 % 3, left, r1)


NUM_INNER, NUM_LEAVES = 33, 34


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def compare(nodes_u8, woop_u8, tri_index, where=""):
    """Walks the built Compact buffers from the root and compares them with expected_tree() bit for bit.  Returns (inner nodes, leaves)."""
    nf = np.frombuffer(bytes(nodes_u8), dtype=np.float32)
    ni = np.frombuffer(bytes(nodes_u8), dtype=np.int32)
    wu = np.frombuffer(bytes(woop_u8), dtype=np.uint32).reshape(-1, 4)
    idx = np.asarray(tri_index, dtype=np.int32)
    count = [0, 0]

    def check_box(got, want, what):
        assert [f32_bits(float(x)) for x in got] == [f32_bits(x) for x in want], "%s %s: box %r, derived %r" % (where, what, [float(x) for x in got], want)

    def walk(ofs, exp, path):
        assert ofs % 64 == 0 and 0 <= ofs < len(nodes_u8), (where, path, ofs)
        count[0] += 1
        w = ofs // 4
        assert int(ni[w + 14]) == exp["word14"], "%s node %s: split word %d, derived %d" % (where, path, int(ni[w + 14]), exp["word14"])
        assert int(ni[w + 15]) == 0
        boxes = ((nf[w + 0], nf[w + 1], nf[w + 2], nf[w + 3], nf[w + 8], nf[w + 9]), (nf[w + 4], nf[w + 5], nf[w + 6], nf[w + 7], nf[w + 10], nf[w + 11]))
        for k in (0, 1):
            child, ce = int(ni[w + 12 + k]), exp["children"][k]
            check_box(boxes[k], ce["box"], "node %s child %d" % (path, k))
            if "leaf" in ce:
                assert child < 0, "%s node %s child %d: an inner node where a leaf was derived" % (where, path, k)
                count[1] += 1
                a = ~child
                for j, (orig, rows) in enumerate(zip(ce["leaf"], ce["rows"])):
                    assert int(idx[a + 3 * j]) == orig, "%s leaf %s.%d triangle %d: index %d, derived %d" % (where, path, k, j, int(idx[a + 3 * j]), orig)
                    for r in range(3):
                        assert [int(x) for x in wu[a + 3 * j + r]] == [f32_bits(x) for x in rows[r]], \
                            "%s leaf %s.%d triangle %d row %d: %r, derived %r" % (where, path, k, j, r, [hex(int(x)) for x in wu[a + 3 * j + r]], rows[r])
                assert int(wu[a + 3 * len(ce["leaf"]), 0]) == 0x80000000, "%s leaf %s.%d: no terminator after %d triangles" % (where, path, k, len(ce["leaf"]))
            else:
                assert child >= 0, "%s node %s child %d: a leaf where an inner node was derived" % (where, path, k)
                walk(child, ce, path + str(k))

    walk(0, expected_tree(), "r")
    return tuple(count)
