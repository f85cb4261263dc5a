"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo"
in the CPU tests).  The traced path has no collective: rays are sharded, the BVH is replicated, and
only hit records / pixels are gathered to rank 0 at the end of a frame."""
import torch
import torch.distributed as dist


def shard_range(num_rays, rank, world, align=64):
    """Contiguous slice [lo, hi) of the primary-ray index space for `rank`.  The index space is
    already 8x8-pixel blocks in Morton order (PixelTable), so a contiguous slice aligned to 64 rays
    is a compact set of screen tiles."""
    blocks = (num_rays + align - 1) // align
    per, extra = divmod(blocks, world)
    lo_b = rank * per + min(rank, extra)
    hi_b = lo_b + per + (1 if rank < extra else 0)
    return min(lo_b * align, num_rays), min(hi_b * align, num_rays)


def gather_hit_records(local, num_rays, align=64, dst=0):
    """Gather every rank's slice of 16-byte hit records (uint8 tensor of (hi-lo)*16 bytes) into the
    full frame on `dst`.  Returns the assembled uint8 tensor on dst, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [(lambda r: (r[1] - r[0]) * 16)(shard_range(num_rays, k, world, align)) for k in range(world)]
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=torch.uint8, device=local.device)
    buf[: local.numel()] = local
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, outs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(outs, sizes)])


def job_throughput(units, seconds, device):
    """(sum of units over ranks, max of seconds over ranks): value = units / seconds."""
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())
