"""Scene ingest: camera-signature codec, primary-ray matrix, OBJ importer.

Known answers from the reference itself: the camera signature string shipped in its config.conf
(Benchmark.camera) must decode and re-encode to the same string, and the two OBJ assets in its data/
directory have the triangle counts SURVEY.md section 2 row 24 records (Map.obj 488, head.obj 18 678)."""
import math
import os

import numpy as np
import pytest

import ntrace_amd as nt

# Benchmark.camera of the reference's config.conf (a data value, not source code)
REF_SIGNATURE = "GBSvz1V04qy/Ju69/21iChCz/idyKy10A0Kfx1pzUoy/DuY2/0aNqY10sZpuu/5/5f/0/"


def test_reference_signature_round_trips():
    cam = nt.camera_decode(REF_SIGNATURE)
    assert all(math.isfinite(v) for v in cam["position"] + cam["forward"] + cam["up"])
    assert abs(np.linalg.norm(cam["forward"]) - 1.0) < 1e-5 and abs(np.linalg.norm(cam["up"]) - 1.0) < 1e-5
    assert 1.0 < cam["fov"] < 179.0 and 0.0 < cam["near"] < cam["far"]
    assert nt.camera_reencode(REF_SIGNATURE) == '"' + REF_SIGNATURE + '",'
    # tolerant input form: quotes, trailing comma, blanks (CameraControls.cpp:362-385)
    assert nt.camera_decode(' "' + REF_SIGNATURE + '", ') == cam
    with pytest.raises(nt.NtrError):
        nt.camera_decode("not*a*signature")


def test_nscreen_to_world_maps_screen_centre_along_forward():
    cam = nt.camera_decode(REF_SIGNATURE)
    for (w, h) in ((1024, 768), (1920, 1080)):
        m, pos, far = nt.camera_nscreen_to_world(REF_SIGNATURE, w, h)
        assert np.allclose(pos, cam["position"]) and far == cam["far"]
        p = m @ np.array([0.0, 0.0, 0.0, 1.0], dtype=np.float32)
        d = p[:3] / p[3] - pos
        d /= np.linalg.norm(d)
        assert np.allclose(d, cam["forward"], atol=2e-4)
        # a point at the right image edge deviates by half the horizontal field of view implied by fitToView
        q = m @ np.array([1.0, 0.0, 0.0, 1.0], dtype=np.float32)
        e = q[:3] / q[3] - pos
        e /= np.linalg.norm(e)
        half_v = math.radians(cam["fov"]) * 0.5
        expect = math.atan(math.tan(half_v) * max(w / h, 1.0)) if w >= h else half_v
        assert abs(math.acos(np.clip(np.dot(d, e), -1, 1)) - expect) < 2e-3


OBJ = """# two quads, a polygon, negative indices, materials
mtllib test.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
vt 0 0
vt 1 1
vn 0 0 1
f 1 2 3 4
usemtl red
f 1/1/1 2/2/1 6/1/1 5/2/1
usemtl unknown_material
f 3 4 5
usemtl blue
f -1 -2 -3 -4 -5
usemtl red
f 1/1/1 3 5
"""
MTL = "newmtl red\nKd 1 0 0\nnewmtl blue\nKd 0 0 1\n"


def test_obj_import_numbering(tmp_path):
    (tmp_path / "test.obj").write_text(OBJ)
    (tmp_path / "test.mtl").write_text(MTL)
    tri, pos, nsub = nt.obj_load(str(tmp_path / "test.obj"))
    # submeshes in creation order: default (first face + the face after the unknown usemtl), red, blue
    assert nsub == 3
    # fan triangulation: quad -> 2, quad -> 2, tri -> 1, pentagon -> 3, tri -> 1
    assert tri.shape[0] == 2 + 1 + 2 + 1 + 3
    # default submesh first: quad (v0,v1,v2),(v0,v2,v3) with vertices numbered in first-seen order
    assert tri[0].tolist() == [0, 1, 2] and tri[1].tolist() == [0, 2, 3]
    assert np.array_equal(pos[:4], np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float32))
    # the face after `usemtl unknown_material` falls back to the default submesh (3rd triangle overall)
    assert np.array_equal(pos[tri[2]], np.array([[1, 1, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float32))
    # red submesh: (1/1/1) is a NEW vertex (different p/t/n triple from plain 1)
    assert tri[3][0] >= 4 and np.array_equal(pos[tri[3][0]], [0, 0, 0])
    # red submesh gets both of its faces (2 + 1 triangles) before blue's pentagon
    red = tri[3:6]
    assert np.array_equal(pos[red[2]], np.array([[0, 0, 0], [1, 1, 0], [0, 0, 1]], dtype=np.float32))
    blue = tri[6:9]
    assert np.array_equal(pos[blue[0][0]], [1, 0, 1])  # -1 = last vertex


def test_obj_import_missing_file():
    with pytest.raises(nt.NtrError):
        nt.obj_load("/nonexistent/file.obj")


@pytest.mark.parametrize("rel,ntris", [("data/models/Map/Map.obj", 488), ("data/models/Head/head.obj", 18678)])
def test_reference_assets_triangle_counts(rel, ntris):
    path = os.path.join("/root/reference", rel)
    if not os.path.exists(path):
        pytest.skip("reference checkout not present on this box")
    tri, pos, nsub = nt.obj_load(path)
    assert tri.shape[0] == ntris
    assert tri.min() >= 0 and tri.max() < pos.shape[0] and np.isfinite(pos).all()
