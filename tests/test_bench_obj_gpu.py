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


@pytest.mark.parametrize("balance", ["count", "predicted"])
def test_bench_rccl_path_at_world_size_one(balance):
    """The multi-GPU code path of bench.py -- RCCL process group, barriers, the library's own BVH broadcast and gather of hit records
    (ntr_dist_broadcast_bvh, ntr_dist_gather_records / _cuts: the default whenever the ranks run on RCCL), MAX / SUM reductions and the
    assembled-frame check -- exercised on one GPU (NTR_BENCH_FORCE_DIST=1), so that the GPU test tier loads RCCL and runs every
    collective the 2 / 4 / 8-GPU runs use, with equal ranges and with a cut table."""
    env = dict(os.environ, NTR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", "640", "--height", "360", "--steps", "2", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline", "--balance", balance], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["gather_ms"] is not None and out["gather_ms"] > 0
    # the frame rank 0 checks bit for bit against its own single-GPU trace IS the one the library's gather assembled
    chk = out["sharded_frame_check"]
    assert chk and chk["primary_records_equal_single_gpu_frame"] and chk["ao_checksum_equal_single_gpu_frame"]
    assert chk["records_compared"] == 640 * 360
    want = "ntr_dist_gather_records_cuts" if balance == "predicted" else "ntr_dist_gather_records "
    assert out["gather_native"] and want in out["gather_native"]["how"] and out["gather_native"]["bvh_broadcast"] == "ntr_dist_broadcast_bvh"
    assert "ntr_dist" in out["config"]["parallelism"]
    # the frame-per-rank mode reported beside `value` for N > 1 (here: one rank, one whole frame)
    fpr = out["extras"]["frame_per_rank"]
    assert fpr["ranks"] == 1 and fpr["steps"] == 2 and fpr["mrays"] > 0 and fpr["ms_per_frame"] > 0


def _bench_ranks(n, extra, timeout=900):
    """`python bench.py --gpus n` as a FRESH child process that starts its own ranks (bench.py self-launches before it touches the GPU)
    with every rank on cuda:0 over gloo: the N > 1 path -- sharded frame, BVH broadcast, per-rank AO from own hits, gather to rank 0,
    bit-compare against the single-GPU frame -- with real HIP tracing on a one-GPU box."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-backend", "gloo", "--one-device", "--steps", "2",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("ranks,size,balance", [(2, (1920, 1080), "count"), (3, (1000, 700), "count"), (2, (1000, 700), "predicted")])
def test_bench_sharded_frame_with_real_kernels_at_world_size_above_one(ranks, size, balance):
    """nothing above world size 1 had ever traced a ray.  Here N ranks (N processes, one GPU, gloo) each trace
    their PixelTable range of ONE frame with the HIP kernels, generate AO rays from their own hits, and rank 0 gathers the records and
    compares the assembled frame bit for bit (primary) / by checksum of checksums (AO) with its own single-GPU trace -- 1080p at two
    ranks, a 1000 x 700 frame at three (ragged 64-aligned cuts: 700 000 rays do not divide), and cuts of equal predicted cost.  What
    stays unexecuted without a multi-GPU node: the RCCL transport at N > 1."""
    w, h = size
    out = _bench_ranks(ranks, ["--width", str(w), "--height", str(h), "--balance", balance])
    assert out["n_gpus"] == ranks and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["one_device"] is True and out["config"]["dist_backend"] == "gloo"
    chk = out["sharded_frame_check"]
    assert chk and chk["primary_records_equal_single_gpu_frame"] is True and chk["ao_checksum_equal_single_gpu_frame"] is True
    assert chk["records_compared"] == w * h
    assert out["gather_ms"] is not None and out["gather_ms"] > 0
    assert out["extras"]["frame_per_rank"]["ranks"] == ranks and out["extras"]["frame_per_rank"]["mrays"] > 0
    # every rank traced only its share: rank 0's primary rays are about 1 / ranks of the frame, whole 64-ray tiles
    own = out["config"]["primary_rays_rank0"]
    assert 0 < own < w * h and own % 64 == 0
    if balance == "count":
        assert abs(own - w * h / ranks) <= 64
    assert out["config"]["rays_per_step"] > out["config"]["rays_per_step_rank0"]


def test_bench_frame_per_rank_mode_at_world_size_two():
    """--scaling weak (a whole frame of its own camera per rank) at two ranks on one device: the mode in which every launch keeps its
    single-GPU size; records of both frames are gathered."""
    out = _bench_ranks(2, ["--width", "640", "--height", "360", "--scaling", "weak"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["rays_per_step"] > 1.5 * out["config"]["rays_per_step_rank0"]
