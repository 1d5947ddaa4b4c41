"""Multi-GPU plumbing of the path: stereo pairs / views are independent units, sharded over
one process per GPU; the only exchange is the gather of per-view depth maps (RCCL over xGMI
when the tensors live on the GPU, gloo on CPU in the tests).  SURVEY.md section 8(e)."""
import torch
import torch.distributed as dist


def shard_units(n_units, world_size, rank):
    """Contiguous, balanced split of units 0..n_units-1; returns range(lo, hi) for `rank`.
    The first (n_units % world_size) ranks take one extra unit."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(n_units, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return range(lo, hi)


def gather_depth_maps(local, dst=0):
    """Gather equally-shaped depth-map tensors to rank `dst`; returns the list (rank order) there,
    None elsewhere.  With world_size 1 returns [local]."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    world, rank = dist.get_world_size(), dist.get_rank()
    out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst)
    return out


def all_gather_depth_maps(local):
    """Every rank receives every rank's maps (MVS cross-check needs all views' maps)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    out = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local)
    return out
