"""Multi-GPU host plumbing: one process per GPU, independent KEM instances
split by contiguous index range, no data-path collective.
torch.distributed is used only for the rendezvous, the barrier and the
max-over-ranks of the timed region."""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of n instances owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def init(backend, device_id=None):
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    kw = {}
    if device_id is not None:
        kw["device_id"] = device_id
    dist.init_process_group(backend, **kw)
    return dist


def max_over_ranks(value, device="cpu"):
    """Largest `value` (float seconds) over all ranks; identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_sharded(n, compute, rank=None, world=None):
    """Run compute(lo, hi) on this rank's slice of n instances.  Returns (lo, hi, result)."""
    r, w, _ = env_rank()
    rank = r if rank is None else rank
    world = w if world is None else world
    lo, hi = shard_bounds(n, rank, world)
    return lo, hi, compute(lo, hi)
