"""The host SAH builder (ntrace_amd/host/bvh/SAHBVHBuilder.cpp: presorted orders, stable partition, parallel subtrees) yields the
byte-identical BVHLayout_Compact buffers of the reference-order builder it replaced -- a statement-for-statement rendering of
src/rt/bvh/SAHBVHBuilder.cpp:51-254 (sort the node's references per axis at every node, right subtree first) -- whose output
hashes were recorded before it was removed: same topology, same triangle order inside every leaf, same boxes, same Woop rows.
Covers leaf preferences (1,1) (Renderer.cpp:88-89), (1,8), (2,4), (8,8), degenerate / duplicate triangles, and the bench scene."""
import hashlib

import numpy as np
import pytest

import ntrace_amd as nt
from ntrace_amd import scenes

RECORDED = {
    "atrium_1_1": "c8a99495eb81d1fcfcf62ad94d9483375bc5ee547488f6e3b50a29b832096cbc",
    "atrium_1_8": "234a0c6bf425561d03e5519c8483c4a0b84a8bcb2d3cfbd7e94cf669632dc9a7",
    "cornell_1_1": "28efe2efa09969c557461b06b022ffcb986e07a34fd4846a2919f43367960ae7",
    "cornell_1_8": "2712c3e223790e0cf1b04d7efbd8535e4bb724d2a888257b46a965c3ba1cbc34",
    "cornell_2_4": "0c61e57979ba034a80c817dcaca750bf140d448ad6d0af7a342d4c8026028d4e",
    "cornell_8_8": "a77746d32ad9c2036167ef56e15f342e4f5b365f9608ff99079d0eb13d767520",
    "degenerate_1_1": "83f61072dd6f2da7f6ba7630888133bdd272c63d4634c81e76ce03dc56721d5e",
    "degenerate_1_8": "62cedb86a8d932b64d86226b2873dd0beb14b76a28034ace59f983e5a32a2268",
    "degenerate_2_4": "03361814612729e76a14b69bb9b5a64178b8dd47ce76f65af6431d5982a5ad87",
    "degenerate_8_8": "06330dae775ee0ab2ce2399f9aad42dfcb211bbc0cd5db14dd96c76fe3466654",
    "soup20000_1_1": "acf164882ec4e9c6d231e9448ed65739ee1f700fca60173915ba64251eb8c723",
    "soup20000_1_8": "2b81ebd7e3dd42dfbed0b76254d25c046221766d2d026a16f0b09c3efbed3d90",
    "soup20000_2_4": "b8548e500bcc121e958eb943cda0eb6f50762a9ac3e605ab9fd6c8b41e9a4555",
    "soup20000_8_8": "ac0c04b4cc9d2226bef788e6cda0171f64881902d51e94c5010eb3dcf90bec79",
    "soup3000_1_1": "d9ab27095db8d1b1a71882539544e29f1991ba994a54f4f1944069ec74bd54e1",
    "soup3000_1_8": "760ab097e87132eab46f5d75a90f89fc7063da3d5f4a273549b5feb6016dd11c",
    "soup3000_2_4": "a43ffaf7e7d64103d1774aed91bc3f6cc0410dfce4000ca489c0f11801b22dcb",
    "soup3000_8_8": "4d936683a82109cf4afa0ebd154decb423140f992a3aa4744080fe41b60d8d99"
}


def degenerate_scene():
    tri, pos, _ = scenes.random_soup(4000, seed=9)
    pos = pos.copy()
    for t in range(0, 4000, 7):          # edge collapse: boxes with at most one non-zero extent are dropped (SAHBVHBuilder.cpp:141-151)
        a, b, c = tri[t]
        pos[b] = pos[a]
    for t in range(3, 4000, 11):         # collinear along x
        a, b, c = tri[t]
        pos[b] = pos[a] + np.float32([1, 0, 0])
        pos[c] = pos[a] + np.float32([2, 0, 0])
    tri = np.concatenate([tri, tri[:500]])  # duplicates: equal sort keys, ties by triangle index (:106-115)
    return tri, pos


SCENES = {
    "cornell": lambda: scenes.cornell_box()[:2],
    "soup20000": lambda: scenes.random_soup(20000, seed=11)[:2],
    "soup3000": lambda: scenes.random_soup(3000, seed=5)[:2],
    "degenerate": degenerate_scene,
    "atrium": lambda: scenes.atrium()[:2],
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_sah_builder_output_is_the_recorded_one(name):
    tri, pos = SCENES[name]()
    for key, want in sorted(RECORDED.items()):
        if not key.startswith(name + "_"):
            continue
        mn, mx = (int(x) for x in key.split("_")[-2:])
        b = nt.sah_build(tri, pos, mn, mx)
        got = hashlib.sha256(b.nodes.tobytes() + b.woop.tobytes() + b.tri_index.tobytes()).hexdigest()
        assert got == want, (key, b.info)


def test_sah_builder_is_deterministic_across_thread_counts():
    """Subtrees are built by separate threads from 200 k triangles on; the output does not depend on how many."""
    tri, pos, _ = scenes.atrium()
    a = nt.sah_build(tri, pos, 1, 1)
    b = nt.sah_build(tri, pos, 1, 1)
    assert np.array_equal(a.nodes, b.nodes) and np.array_equal(a.woop, b.woop) and np.array_equal(a.tri_index, b.tri_index)
