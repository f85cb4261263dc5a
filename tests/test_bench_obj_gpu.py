"""The real-asset path of bench.py (`--scene-obj` / `--camera`, what a supplied sponza.obj would take) on fixture-sized
geometry: the triangles of the reference's Map.obj (tests/golden/map_obj_mixed.npz, read through this repo's importer when
the fixture was made) written back as a Wavefront OBJ, traced once from the bounding-box camera and once from the camera
signature of the reference's config.conf.  The bench compares every hit record of the step with the oracle itself
(cpu_baseline.parity_mismatches_whole_step)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SIGNATURE = "GBSvz1V04qy/Ju69/21iChCz/idyKy10A0Kfx1pzUoy/DuY2/0aNqY10sZpuu/5/5f/0/"  # config.conf of the reference


def _write_obj(path, tri, pos):
    with open(path, "w") as f:
        for p in pos:
            f.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        for t in tri:
            f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


def _bench(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("camera", ["bounding-box", "config.conf signature"])
def test_bench_scene_obj_path(tmp_path, camera):
    g = np.load(os.path.join(ROOT, "tests", "golden", "map_obj_mixed.npz"))
    obj = str(tmp_path / "map.obj")
    _write_obj(obj, g["tri"], g["pos"])
    args = ["--scene-obj", obj, "--width", "320", "--height", "200", "--steps", "2", "--warmup", "1", "--no-extras"]
    if camera != "bounding-box":
        args += ["--camera", REF_SIGNATURE]
    out = _bench(args)
    assert out["data"].startswith("OBJ map.obj") and out["config"]["triangles"] == g["tri"].shape[0]
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["cpu_baseline"]["parity_mismatches_whole_step"] == 0
    assert out["cpu_baseline"]["rays_compared"] >= 320 * 200
    if camera == "bounding-box":
        assert out["config"]["primary_hits_rank0"] > 1000   # looking down the long axis from inside the box


def test_bench_rccl_path_at_world_size_one():
    """The multi-GPU code path of bench.py -- RCCL process group, barriers, BVH broadcast, MAX / SUM reductions, the gather of hit
    records to rank 0 and the assembled-frame check -- exercised on one GPU (NTR_BENCH_FORCE_DIST=1), so that the GPU test tier
    loads RCCL and runs every collective the 2 / 4 / 8-GPU runs use."""
    env = dict(os.environ, NTR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", NTR_BENCH_NATIVE_GATHER="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", "640", "--height", "360", "--steps", "2", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["gather_ms"] is not None and out["gather_ms"] > 0
    chk = out["sharded_frame_check"]
    assert chk and chk["primary_records_equal_single_gpu_frame"] and chk["ao_checksum_equal_single_gpu_frame"]
    assert chk["records_compared"] == 640 * 360
    # the library's own gather (ntr_dist_gather_records, RCCL bound by the C-ABI) assembled the same frame
    assert out["gather_native"] and out["gather_native"]["equal_torch_gather"] is True
    # the frame-per-rank mode reported beside `value` for N > 1 (here: one rank, one whole frame)
    fpr = out["extras"]["frame_per_rank"]
    assert fpr["ranks"] == 1 and fpr["steps"] == 2 and fpr["mrays"] > 0 and fpr["ms_per_frame"] > 0
