"""Multi-GPU plumbing of bench.py / stage1_dist.py: one process per GPU; the launcher side only needs the process group, a barrier
and the max-reduce of the timed interval (the exchange of a sharded sample happens inside the engine: t4_comm)."""
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


def max_over_ranks(dist, seconds, device):
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
