"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo"
in the CPU tests).  The traced path has no collective: rays are sharded by screen tile, the BVH is
replicated, and only hit records / pixels are gathered to rank 0 at the end of a frame
(SURVEY.md section 8(e)).

FramePlan is the partition of ONE frame: the primary-ray index space [0, W*H) -- already 8x8-pixel
blocks in Morton order (PixelTable, src/rt/ray/PixelTable.cpp:77-122) -- is cut into `world` contiguous
ranges aligned to 64 rays, so a rank owns a compact set of screen tiles; the rank's AO / diffuse rays
derive from its OWN primary hits, in batches of at most maxBatchSize output rays as RayGen::ao
produces them (src/rt/ray/RayGen.cpp:582-602), so no ray or hit ever crosses a rank boundary."""
import torch
import torch.distributed as dist


def _staged(t):
    """gloo moves host memory only: a device tensor takes part in a gloo collective through a host copy (the CPU test tier, and the
    GPU-tier runs of bench.py with every rank on ONE device -- `--dist-backend gloo --one-device` -- where RCCL cannot be used: it
    refuses two ranks on the same GPU)."""
    return dist.get_backend() == "gloo" and t.is_cuda


def broadcast_(t, src):
    if _staged(t):
        h = t.cpu()
        dist.broadcast(h, src)
        t.copy_(h)
    else:
        dist.broadcast(t, src)
    return t


def all_reduce_(t, op):
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def gather_(buf, outs, dst):
    """dist.gather with the staging above; `outs` (on dst) are filled in place."""
    if _staged(buf):
        h_outs = [torch.empty(o.shape, dtype=o.dtype) for o in outs] if outs is not None else None
        dist.gather(buf.cpu(), h_outs, dst=dst)
        if outs is not None:
            for o, h in zip(outs, h_outs):
                o.copy_(h)
    else:
        dist.gather(buf, outs, dst=dst)


def shard_range(num_rays, rank, world, align=64):
    """Contiguous slice [lo, hi) of the primary-ray index space for `rank`.  The index space is
    already 8x8-pixel blocks in Morton order (PixelTable), so a contiguous slice aligned to 64 rays
    is a compact set of screen tiles."""
    # the arithmetic lives in the C-ABI (ntr_frame_shard, ntr_dist.cpp), where the native multi-GPU driver and the C++ Renderer use it too
    from ._capi import frame_shard
    return frame_shard(num_rays, rank, world, align)


def balanced_cuts(block_cost, num_rays, world, flat_share=1.0, block=256):
    """Cut points c[0] = 0 <= c[1] <= ... <= c[world] = num_rays (multiples of `block` rays, hence of the 64-ray tile alignment)
    such that every range [c[r], c[r+1]) carries about the same share of the PREDICTED cost of the frame.

    block_cost   predicted cost of every `block`-ray block of the primary batch (ntr_predict_block_costs: top-of-tree boxes the
                 block's sample ray intersects), any array-like of non-negative numbers
    flat_share   the part of a block's cost that does not depend on where its rays go -- ray load / result store and the AO
                 rays of its hits, which are short and cost about the same everywhere -- as a multiple of the MEAN predicted
                 cost: a block weighs cost[b] + flat_share * mean(cost).  0 balances the predictor alone; a large value tends
                 to equal ray counts (shard_range).  Fitted on the simulated ranks of scripts/studies/shard_balance_study.py.
    The ranges stay contiguous in the PixelTable index space, so a rank still owns a compact set of screen tiles."""
    import numpy as np
    c = np.asarray(block_cost, dtype=np.float64).reshape(-1)
    nb = (int(num_rays) + block - 1) // block
    if c.size < nb:
        c = np.concatenate([c, np.full(nb - c.size, c.mean() if c.size else 1.0)])
    c = c[:nb]
    w = c + flat_share * (c.mean() if nb else 0.0) + 1e-9
    if nb:   # the last block may be partial: weigh it by its ray count
        w[-1] *= (int(num_rays) - (nb - 1) * block) / float(block)
    acc = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        target = acc[-1] * r / world
        b = int(np.searchsorted(acc, target, side="left"))   # first boundary with at least the target before it
        if b > 0 and abs(acc[b - 1] - target) <= abs(acc[min(b, nb)] - target):
            b -= 1
        cuts.append(max(min(b, nb) * block, cuts[-1]))
    cuts.append(int(num_rays))
    # interior cuts stay aligned: one that lands on (or beyond) a partial last block moves down to the last aligned position
    top = (int(num_rays) // block) * block
    return [min(int(x), top) for x in cuts[:-1]] + [int(num_rays)]


def broadcast_cuts(cuts, world, device, src=0):
    """The planning rank's cut points to every rank (identity without a process group)."""
    t = torch.tensor([int(x) for x in cuts] if cuts is not None else [0] * (world + 1), dtype=torch.int64, device=device)
    if dist.is_initialized():
        broadcast_(t, src)
    return [int(x) for x in t.tolist()]


class FramePlan:
    """What `rank` of `world` traces of one frame of `num_primary` primary rays.

    lo, hi       the rank's slice of the primary index space: the rank-th of `world` equal 64-aligned ranges, or [cuts[rank],
                 cuts[rank + 1]) when cut points are given (balanced_cuts: ranges of equal predicted cost)
    ao_batches   [(first_input_slot, num_inputs)], global primary slots: batch b generates
                 num_inputs * samples secondary rays from the rank's primary hits (none when samples == 0)"""

    def __init__(self, num_primary, rank, world, samples=8, max_batch_rays=1 << 20, align=64, cuts=None):
        self.num_primary, self.rank, self.world, self.samples = int(num_primary), int(rank), int(world), int(samples)
        if cuts is not None:
            if len(cuts) != world + 1 or cuts[0] != 0 or cuts[-1] != num_primary or any(cuts[i] > cuts[i + 1] for i in range(world)) \
                    or any(c % align for c in cuts[:-1]):
                raise ValueError("FramePlan: cut points must be %d non-decreasing multiples of %d from 0 to %d" % (world + 1, align, num_primary))
            self.lo, self.hi = int(cuts[rank]), int(cuts[rank + 1])
        else:
            self.lo, self.hi = shard_range(num_primary, rank, world, align)
        self.cuts = list(cuts) if cuts is not None else None
        from ._capi import frame_ao_batches
        self.ao_batches = frame_ao_batches(self.lo, self.hi, samples, max(int(max_batch_rays), 1)) if samples > 0 else []

    @property
    def num_own_primary(self):
        return self.hi - self.lo


def gather_hit_records(local, num_rays, align=64, dst=0, cuts=None):
    """Gather every rank's slice of 16-byte hit records (uint8 tensor of (hi-lo)*16 bytes) into the
    full frame on `dst` (slices as FramePlan cuts them: equal ranges, or `cuts`).  Returns the assembled uint8 tensor on
    dst, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if cuts is not None:
        sizes = [(int(cuts[k + 1]) - int(cuts[k])) * 16 for k in range(world)]
    else:
        sizes = [(lambda r: (r[1] - r[0]) * 16)(shard_range(num_rays, k, world, align)) for k in range(world)]
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=torch.uint8, device=local.device)
    buf[: local.numel()] = local
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    gather_(buf, outs, dst)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(outs, sizes)])


def records_checksum(records_u8):
    """Order-independent 64-bit checksum of a buffer of 16-byte hit records: the wrapping int64 sum of
    (id, t bits) of every record, each mixed with an odd multiplier.  Sums of the ranks' checksums equal the
    checksum of the assembled frame (a checksum of checksums)."""
    if records_u8.numel() == 0:
        return 0
    w = records_u8.view(torch.int32).view(-1, 4).to(torch.int64)
    v = (w[:, 0] * 0x9E3779B1 + w[:, 1] * 0x85EBCA77).sum()
    return int(v.item())


def wrap_i64(x):
    """Python int -> the signed 64-bit value with the same residue modulo 2^64."""
    return ((int(x) + (1 << 63)) % (1 << 64)) - (1 << 63)


def all_sum_int64(value, device):
    """Wrapping int64 sum of `value` over the ranks (identity without a process group)."""
    t = torch.tensor([wrap_i64(value)], dtype=torch.int64, device=device)
    if dist.is_initialized():
        all_reduce_(t, dist.ReduceOp.SUM)
    return int(t.item())


def job_throughput(units, seconds, device):
    """(sum of units over ranks, max of seconds over ranks): value = units / seconds."""
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist.is_initialized():
        all_reduce_(t, dist.ReduceOp.MAX)
        all_reduce_(u, dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())


def frame_per_rank_summary(rays_this_rank, kernel_seconds_this_rank, steps, device):
    """The frame-per-rank scaling mode of bench.py (every rank traces a whole frame of its own camera; no ray or record crosses a rank
    boundary): rays of all ranks over `steps` frames each / MAX over ranks of the summed kernel time.  Returns the dict bench.py emits as
    extras.frame_per_rank for N > 1 (same reduction as `value`: SUM of units, MAX of seconds)."""
    rays, secs = job_throughput(rays_this_rank, kernel_seconds_this_rank, device)
    world = dist.get_world_size() if dist.is_initialized() else 1
    return {"what": "every rank traces a whole 1920x1080 primary + AO frame of its own camera against the replicated BVH (weak scaling): "
                    "rays of all ranks / MAX over ranks of the summed per-batch kernel times",
            "ranks": world, "steps": int(steps), "rays_per_frame_all_ranks": rays,
            "mrays": rays * steps / secs / 1e6 if secs > 0 else None,
            "ms_per_frame": secs / steps * 1e3 if steps > 0 else None}


def broadcast_bytes(buf_u8, src, device):
    """Replicates a byte buffer from `src` (BVH replication: built once, broadcast).  `buf_u8` is a
    uint8 tensor on `src` and may be None elsewhere; returns the tensor on `device` on every rank."""
    if not dist.is_initialized():
        return buf_u8.to(device)
    n = torch.tensor([buf_u8.numel() if dist.get_rank() == src else 0], dtype=torch.int64, device=device)
    broadcast_(n, src)
    out = buf_u8.to(device) if dist.get_rank() == src else torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    broadcast_(out, src)
    return out
