"""World-size-2 gloo test of the N>1 path: ray sharding by screen-tile ranges, the final gather of
hit records to rank 0, and the whole-job throughput reduction used by bench.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ntrace_amd import dist as ntd
from ntrace_amd import scenes


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_rays, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ntd.shard_range(num_rays, rank, world)
    # stand-in for this rank's traced slice: record i = (id = i, t = i * 0.5, 0, 0)
    rec = np.zeros(hi - lo, dtype=[("id", "<i4"), ("t", "<f4"), ("a", "<i4"), ("b", "<i4")])
    rec["id"] = np.arange(lo, hi)
    rec["t"] = np.arange(lo, hi) * 0.5
    local = torch.from_numpy(rec.view(np.uint8).copy())
    full = ntd.gather_hit_records(local, num_rays)
    units, secs = ntd.job_throughput(hi - lo, 1.0 + rank, torch.device("cpu"))
    if rank == 0:
        got = full.numpy().view(rec.dtype)
        ok = (got.shape[0] == num_rays and np.array_equal(got["id"], np.arange(num_rays))
              and np.array_equal(got["t"], (np.arange(num_rays) * 0.5).astype(np.float32))
              and units == num_rays and secs == float(world))
        open(out_path, "w").write("ok" if ok else "bad %s %s %s" % (got.shape, units, secs))
    dist.destroy_process_group()


def _standin_primary(lo, hi):
    """Stand-in for the traced primary slice [lo, hi): a deterministic record per global slot."""
    i = np.arange(lo, hi, dtype=np.int64)
    rec = np.zeros(hi - lo, dtype=[("id", "<i4"), ("t", "<f4"), ("a", "<i4"), ("b", "<i4")])
    rec["id"] = np.where((i * 2654435761) % 7 == 0, -1, (i * 40503) % 100000).astype(np.int32)
    rec["t"] = (i % 1000).astype(np.float32) * 0.25
    return rec


def _standin_ao(primary_rec_of_slot, first, cnt, samples):
    """Stand-in for one traced AO batch: record of (global input slot s, sample k); missed inputs give degenerate rays."""
    s_ = np.repeat(np.arange(first, first + cnt, dtype=np.int64), samples)
    k = np.tile(np.arange(samples, dtype=np.int64), cnt)
    rec = np.zeros(cnt * samples, dtype=[("id", "<i4"), ("t", "<f4"), ("a", "<i4"), ("b", "<i4")])
    parent = primary_rec_of_slot(first, first + cnt)
    miss = np.repeat(parent["id"] == -1, samples)
    rec["id"] = np.where(miss | ((s_ + k) % 3 == 0), -1, ((s_ * 31 + k) % 5000)).astype(np.int32)
    rec["t"] = np.where(miss, -1.0, ((s_ * 7 + k) % 640) * 0.0078125).astype(np.float32)
    return rec


def _sharded_frame_worker(rank, world, port, num_primary, samples, max_batch, out_path):
    """The whole N>1 flow of bench.py on gloo with a stand-in trace: BVH bytes broadcast from rank 0, FramePlan sharding,
    AO batches from the rank's own primary hits, gather of the primary records, checksum of checksums over the AO
    records, and rank 0's comparison with the single-rank frame."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    blob = torch.arange(1000, dtype=torch.int64).view(torch.uint8) if rank == 0 else None
    rep = ntd.broadcast_bytes(blob, 0, cpu)
    ok = rep.numel() == 8000 and int(rep.view(torch.int64).sum()) == 999 * 1000 // 2
    plan = ntd.FramePlan(num_primary, rank, world, samples, max_batch)
    own = _standin_primary(plan.lo, plan.hi)
    ao_ck = 0
    for (first, cnt) in plan.ao_batches:
        ok = ok and plan.lo <= first and first + cnt <= plan.hi and cnt * samples <= max(max_batch, samples)
        ao_ck += ntd.records_checksum(torch.from_numpy(_standin_ao(_standin_primary, first, cnt, samples).view(np.uint8).copy()))
    ok = ok and sum(c for _, c in plan.ao_batches) == plan.num_own_primary
    full = ntd.gather_hit_records(torch.from_numpy(own.view(np.uint8).copy()), num_primary)
    ao_sum = ntd.all_sum_int64(ao_ck, cpu)
    units, secs = ntd.job_throughput(plan.num_own_primary * (1 + samples), 0.5 + rank, cpu)
    # the frame-per-rank mode bench.py reports beside `value` for N > 1 (extras.frame_per_rank): SUM of rays, MAX of seconds
    fpr = ntd.frame_per_rank_summary(1000 * (rank + 1), 0.25 * (rank + 1), 5, cpu)
    ok = ok and fpr["ranks"] == world and fpr["steps"] == 5 and fpr["rays_per_frame_all_ranks"] == 1000 * world * (world + 1) // 2
    ok = ok and abs(fpr["ms_per_frame"] - 0.25 * world / 5 * 1e3) < 1e-9 and abs(fpr["mrays"] - fpr["rays_per_frame_all_ranks"] * 5 / (0.25 * world) / 1e6) < 1e-12
    if rank == 0:
        ref = _standin_primary(0, num_primary)
        ok = ok and torch.equal(full, torch.from_numpy(ref.view(np.uint8).copy()))
        ref_ao = 0
        for r in range(world):
            for (first, cnt) in ntd.FramePlan(num_primary, r, world, samples, max_batch).ao_batches:
                ref_ao += ntd.records_checksum(torch.from_numpy(_standin_ao(_standin_primary, first, cnt, samples).view(np.uint8).copy()))
        ok = ok and ntd.wrap_i64(ref_ao) == ntd.wrap_i64(ao_sum)
        # the single-rank plan (what N = 1 traces) covers the same records in other batches: same checksum
        one = 0
        for (first, cnt) in ntd.FramePlan(num_primary, 0, 1, samples, max_batch).ao_batches:
            one += ntd.records_checksum(torch.from_numpy(_standin_ao(_standin_primary, first, cnt, samples).view(np.uint8).copy()))
        ok = ok and ntd.wrap_i64(one) == ntd.wrap_i64(ao_sum)
        ok = ok and units == num_primary * (1 + samples) and secs == 0.5 + (world - 1)
        open(out_path, "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_frame_plan_batches():
    p = ntd.FramePlan(1920 * 1080, 3, 8, 8, 1 << 20)
    assert p.lo % 64 == 0 and p.hi - p.lo == 259200
    assert p.ao_batches == [(p.lo, 131072), (p.lo + 131072, 128128)]
    assert ntd.FramePlan(1000, 0, 1, 0).ao_batches == []
    whole = ntd.FramePlan(1920 * 1080, 0, 1, 8, 1 << 20)
    assert len(whole.ao_batches) == 16 and sum(c for _, c in whole.ao_batches) == 1920 * 1080
    assert ntd.wrap_i64((1 << 63) + 5) == -(1 << 63) + 5 and ntd.wrap_i64(-1) == -1


def test_two_rank_sharded_frame_gloo(tmp_path):
    out = str(tmp_path / "frame.txt")
    mp.spawn(_sharded_frame_worker, args=(2, _free_port(), 20000, 8, 4096, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_shard_ranges_cover_and_align():
    for n in (1, 63, 64, 65, 1000, 1920 * 1080):
        for world in (1, 2, 3, 8):
            r = [ntd.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(lo % 64 == 0 for lo, hi in r if lo < n)
    # a 64-aligned slice of the PixelTable order is whole 8x8 pixel tiles
    tab = scenes.pixel_table(128, 64)
    lo, hi = ntd.shard_range(128 * 64, 1, 4)
    px = tab[lo:hi]
    tiles = set(zip((px % 128) // 8, (px // 128) // 8))
    assert len(tiles) * 64 == hi - lo


def test_partition_arithmetic_of_the_c_abi_equals_the_python_formulation():
    """ntr_frame_shard / ntr_frame_ao_batches (ntr_dist.cpp: what FramePlan, the C++ Renderer::setShard and the native gather use)
    against the formulation they replaced in this module."""
    def py_range(n, rank, world, align=64):
        blocks = (n + align - 1) // align
        per, extra = divmod(blocks, world)
        lo_b = rank * per + min(rank, extra)
        hi_b = lo_b + per + (1 if rank < extra else 0)
        return min(lo_b * align, n), min(hi_b * align, n)
    for n in (0, 1, 63, 64, 65, 1000, 4097, 1920 * 1080, 3840 * 2160):
        for world in (1, 2, 3, 5, 8):
            for rank in range(world):
                lo, hi = ntd.shard_range(n, rank, world)
                assert (lo, hi) == py_range(n, rank, world)
                for samples, mb in ((8, 1 << 20), (32, 1 << 20), (8, 4096), (1, 7)):
                    per = max(mb // samples, 1)
                    want = [(f, min(per, hi - f)) for f in range(lo, hi, per)]
                    assert ntd.FramePlan(n, rank, world, samples, mb).ao_batches == want


def test_two_rank_gather_gloo(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), 1000, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_bench_refuses_mismatched_world_size():
    """`bench.py --gpus N` under a launcher that started another number of ranks is an error, not a silent 1-GPU run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4 but the launcher started WORLD_SIZE=2" in (r.stderr + r.stdout)


def _balanced_frame_worker(rank, world, port, num_primary, samples, max_batch, out_path):
    """The N>1 flow with ranges of equal PREDICTED cost: rank 0 alone knows the block costs, cuts the frame and broadcasts the cut
    points; the ranks' unequal slices are gathered; the assembled frame equals the single-rank frame."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    cuts = None
    if rank == 0:
        cost = np.ones((num_primary + 255) // 256)
        cost[: cost.size // 4] = 9.0          # a heavy first quarter of the screen
        cuts = ntd.balanced_cuts(cost, num_primary, world, flat_share=0.5)
    cuts = ntd.broadcast_cuts(cuts, world, cpu)
    plan = ntd.FramePlan(num_primary, rank, world, samples, max_batch, cuts=cuts)
    ok = plan.lo == cuts[rank] and plan.hi == cuts[rank + 1] and plan.lo % 64 == 0
    own = _standin_primary(plan.lo, plan.hi)
    ao_ck = 0
    for (first, cnt) in plan.ao_batches:
        ok = ok and plan.lo <= first and first + cnt <= plan.hi
        ao_ck += ntd.records_checksum(torch.from_numpy(_standin_ao(_standin_primary, first, cnt, samples).view(np.uint8).copy()))
    full = ntd.gather_hit_records(torch.from_numpy(own.view(np.uint8).copy()), num_primary, cuts=plan.cuts)
    ao_sum = ntd.all_sum_int64(ao_ck, cpu)
    if rank == 0:
        ok = ok and cuts[1] < num_primary // world          # the heavy region makes rank 0's range shorter than an equal share
        ref = _standin_primary(0, num_primary)
        ok = ok and torch.equal(full, torch.from_numpy(ref.view(np.uint8).copy()))
        one = 0
        for (first, cnt) in ntd.FramePlan(num_primary, 0, 1, samples, max_batch).ao_batches:
            one += ntd.records_checksum(torch.from_numpy(_standin_ao(_standin_primary, first, cnt, samples).view(np.uint8).copy()))
        ok = ok and ntd.wrap_i64(one) == ntd.wrap_i64(ao_sum)
        open(out_path, "w").write("ok" if ok else "bad %s" % cuts)
    dist.destroy_process_group()


def test_two_rank_balanced_frame_gloo(tmp_path):
    out = str(tmp_path / "balanced.txt")
    mp.spawn(_balanced_frame_worker, args=(2, _free_port(), 20000, 8, 4096, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_balanced_cuts_properties():
    rng = np.random.default_rng(3)
    n = 1920 * 1080
    cost = rng.integers(0, 60, (n + 255) // 256).astype(np.float64)
    cost[2000:3000] += 100
    for world in (1, 2, 3, 8):
        for flat in (0.0, 1.0, 50.0):
            cuts = ntd.balanced_cuts(cost, n, world, flat)
            assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == n
            assert all(a <= b for a, b in zip(cuts, cuts[1:])) and all(c % 256 == 0 for c in cuts[:-1])
            w = cost + flat * cost.mean()
            shares = [w[a // 256:(b + 255) // 256].sum() for a, b in zip(cuts, cuts[1:])]
            assert max(shares) <= 1.02 * np.mean(shares) + w.max()      # equal shares up to one block
            for r in range(world):
                ntd.FramePlan(n, r, world, 8, 1 << 20, cuts=cuts)           # valid plans
    # a large flat share tends to equal ray counts
    cuts = ntd.balanced_cuts(cost, n, 8, 1e6)
    assert all(abs((b - a) - n / 8) <= 512 for a, b in zip(cuts, cuts[1:]))
    # an interior cut that lands on a partial last block stays aligned (it moves down to the last aligned position)
    cuts = ntd.balanced_cuts(np.array([1.0, 1.0, 1.0, 50.0]), 1000, 8)
    assert cuts[0] == 0 and cuts[-1] == 1000 and all(c % 64 == 0 for c in cuts[:-1]) and all(a <= b for a, b in zip(cuts, cuts[1:]))
    for r in range(8):
        ntd.FramePlan(1000, r, 8, 8, 1 << 20, cuts=cuts)
    # degenerate inputs
    assert ntd.balanced_cuts([], 0, 4) == [0, 0, 0, 0, 0]
    assert ntd.balanced_cuts([5.0], 100, 4)[-1] == 100
    import pytest
    with pytest.raises(ValueError):
        ntd.FramePlan(1000, 0, 2, cuts=[0, 100, 1000])      # not tile-aligned
