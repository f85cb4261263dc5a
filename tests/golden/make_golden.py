"""Generates the golden fixtures under tests/golden/ (inputs + expected outputs).

The reference has no fixtures for this path and cannot run here, so the expected values are
produced by the CPU oracle and accepted only when the independent numpy restatement
(tests/np_tracer.py) reproduces them bit for bit.  Re-run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import ntrace_amd as nt  # noqa: E402
import np_tracer  # noqa: E402
from ntrace_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402
from test_trace_gpu import edge_rays  # noqa: E402


def make(name, tri, pos, rays):
    bvh = nt.sah_build(tri, pos, 1, 1)
    out = dict(nodes=bvh.nodes, woop=bvh.woop, tri_index=bvh.tri_index, rays=rays.view(np.float32).reshape(-1, 8),
               tri=tri, pos=pos)
    for any_hit, key in ((False, "closest"), (True, "any")):
        ref, st = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays, any_hit=any_hit)
        nid, ntt = np_tracer.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays, any_hit=any_hit)
        assert np.array_equal(nid, ref["id"]) and np.array_equal(ntt.view(np.uint32), ref["t"].view(np.uint32)), name
        packed = np.zeros(rays.shape[0], dtype=nt.RESULT_DTYPE)
        packed["id"], packed["t"] = ref["id"], ref["t"]
        out["res_" + key] = packed.view(np.int32).reshape(-1, 4)
        out["stats_" + key] = np.array([st.numInnerVisits, st.numTriTests, st.numLeafVisits, st.numHits], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "tris", tri.shape[0], "rays", rays.shape[0], "hit", float((out["res_closest"][:, 0] >= 0).mean()))


if __name__ == "__main__":
    tri, pos, cam = scenes.cornell_box()
    make("cornell_primary", tri, pos, scenes.primary_rays(cam, 64, 36)[0])
    tri, pos, cam = scenes.random_soup(1500, seed=77)
    rays = np.concatenate([scenes.primary_rays(cam, 32, 32)[0], scenes.random_rays(1024, seed=78), edge_rays()])
    make("soup1500_mixed", tri, pos, rays)
