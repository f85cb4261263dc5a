"""Generates the golden fixtures under tests/golden/ (inputs + expected outputs).

The reference has no fixtures for this path and cannot run here.  The expected values (hit records AND
traversal counters) are produced by the numpy restatement tests/np_tracer.py -- NOT by the C oracle -- so
that the tests check both the C oracle and the HIP kernels against a second implementation; the script
refuses to write a fixture on which the C oracle disagrees.  Hand-derived vectors that depend on neither
are in tests/kat_vectors.py.  Re-run: python tests/golden/make_golden.py
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
from ray_sets import edge_rays  # noqa: E402


def make(name, tri, pos, rays):
    bvh = nt.sah_build(tri, pos, 1, 1)
    out = dict(nodes=bvh.nodes, woop=bvh.woop, tri_index=bvh.tri_index, rays=rays.view(np.float32).reshape(-1, 8),
               tri=tri, pos=pos)
    for any_hit, key in ((False, "closest"), (True, "any")):
        nid, ntt, nst = np_tracer.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays, any_hit=any_hit, return_stats=True)
        packed = np.zeros(rays.shape[0], dtype=nt.RESULT_DTYPE)
        packed["id"], packed["t"] = nid, ntt
        out["res_" + key] = packed.view(np.int32).reshape(-1, 4)
        out["stats_" + key] = np.array([nst["numInnerVisits"], nst["numTriTests"], nst["numLeafVisits"], nst["numHits"]], dtype=np.int64)
        # the C oracle must agree before the fixture is written
        ref, st = oracle.trace(bvh.nodes, bvh.woop, bvh.tri_index, rays, any_hit=any_hit)
        assert np.array_equal(nid, ref["id"]) and np.array_equal(ntt.view(np.uint32), ref["t"].view(np.uint32)), name
        assert [st.numInnerVisits, st.numTriTests, st.numLeafVisits, st.numHits] == list(out["stats_" + key]), (name, key, st.as_dict(), nst)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "tris", tri.shape[0], "rays", rays.shape[0], "hit", float((out["res_closest"][:, 0] >= 0).mean()))


def bbox_camera(pos):
    """Pinhole camera inside the bounding box looking down its longest axis (as bench.py --scene-obj)."""
    lo, hi = pos.min(0).astype(np.float64), pos.max(0).astype(np.float64)
    ax = int(np.argmax(hi - lo))
    eye = 0.5 * (lo + hi)
    eye[ax] = lo[ax] + 0.15 * (hi[ax] - lo[ax])
    tgt = eye.copy()
    tgt[ax] = hi[ax]
    return dict(eye=tuple(eye), target=tuple(tgt), up=(0.0, 1.0, 0.0) if ax != 1 else (0.0, 0.0, 1.0), fov_deg=60.0,
                far=3.0 * float(np.linalg.norm(hi - lo)))


def bbox_rays(pos, n, seed):
    """Random segments between points of the (slightly enlarged) bounding box."""
    rng = np.random.default_rng(seed)
    lo, hi = pos.min(0), pos.max(0)
    c, e = 0.5 * (lo + hi), 0.6 * (hi - lo)
    a = (c + rng.uniform(-1, 1, size=(n, 3)) * e).astype(np.float32)
    b = (c + rng.uniform(-1, 1, size=(n, 3)) * e).astype(np.float32)
    d = b - a
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20)
    rays = np.zeros(n, dtype=nt.RAY_DTYPE)
    d = d.astype(np.float32)
    for k, col in (("ox", a[:, 0]), ("oy", a[:, 1]), ("oz", a[:, 2]), ("dx", d[:, 0]), ("dy", d[:, 1]), ("dz", d[:, 2])):
        rays[k] = col
    rays["tmin"], rays["tmax"] = 0.0, np.float32(2.0 * np.linalg.norm(hi - lo))
    return rays


if __name__ == "__main__":
    # geometry of a reference asset (data/models/Map/Map.obj, 488 triangles) through this repo's OBJ importer;
    # needs the reference checkout, so it is only regenerated in the build container
    obj = "/root/reference/data/models/Map/Map.obj"
    if os.path.exists(obj):
        tri, pos, _ = nt.obj_load(obj)
        cam = bbox_camera(pos)
        rays = np.concatenate([scenes.primary_rays(cam, 48, 48)[0], bbox_rays(pos, 2048, seed=488)])
        make("map_obj_mixed", tri, pos, rays)
    obj = "/root/reference/data/models/Head/head.obj"  # 18 678 triangles
    if os.path.exists(obj):
        tri, pos, _ = nt.obj_load(obj)
        cam = bbox_camera(pos)
        rays = np.concatenate([scenes.primary_rays(cam, 40, 40)[0], bbox_rays(pos, 1024, seed=18678)])
        make("head_obj_mixed", tri, pos, rays)
    tri, pos, cam = scenes.cornell_box()
    make("cornell_primary", tri, pos, scenes.primary_rays(cam, 64, 36)[0])
    tri, pos, cam = scenes.random_soup(1500, seed=77)
    rays = np.concatenate([scenes.primary_rays(cam, 32, 32)[0], scenes.random_rays(1024, seed=78), edge_rays()])
    make("soup1500_mixed", tri, pos, rays)
