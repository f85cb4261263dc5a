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


def test_two_rank_gather_gloo(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), 1000, out), nprocs=2, join=True)
    assert open(out).read() == "ok"
