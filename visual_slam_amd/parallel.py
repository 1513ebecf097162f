"""Multi-GPU sharding of independent frame pairs: one process per GPU,
contiguous blocks of pairs per rank, no collective on the data path.  torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" on CPU for the tests) is used only to gather the small
per-pair result records and for timing barriers."""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` pairs owned by `rank`; blocks differ by at most one pair."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_pair_results(local, total, device=None, force=False):
    """all_gather of the per-pair result records (numpy structured array) of every rank, in pair
    order.  Works on any initialised backend; without an initialised group returns `local`.
    `force`: go through the collective even in a group of one rank (a test of the RCCL path on a one-GPU box)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return local
    world = dist.get_world_size()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
    itemsize = local.dtype.itemsize
    cap = max(shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world))
    buf = np.zeros(cap * itemsize, np.uint8)
    buf[: local.size * itemsize] = local.view(np.uint8).ravel()
    mine = torch.from_numpy(buf).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = []
    for r, p in enumerate(parts):
        lo, hi = shard_range(total, r, world)
        out.append(p.cpu().numpy()[: (hi - lo) * itemsize].view(local.dtype))
    return np.concatenate(out)


def max_over_ranks(value, device=None, force=False):
    """MAX all-reduce of a python float (the bench's step time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return float(value)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
