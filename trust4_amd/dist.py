"""Multi-GPU plumbing of bench.py: one process per GPU, read batches sharded by rank, no data-path
collective on the read-only passes; only a barrier and a max-reduce of the timed interval."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend):
    """backend 'nccl' (= RCCL on ROCm) on GPUs, 'gloo' in the CPU tests. Returns the dist module or None."""
    rank, _, world = env_rank()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def shard_seed(base_seed, rank):
    """Weak scaling: every rank synthesises its own batch of the configured size from seed base+rank."""
    return base_seed + rank


def max_over_ranks(dist, seconds, device):
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
