"""Multi-GPU plumbing for the sharded detector path: frames are independent, so ranks only share the
frozen .tflite bytes (one broadcast from rank 0, RCCL over xGMI when the backend is "nccl") and a max-reduced timer.
No data-path collective exists.  The same code runs under gloo on CPU (tests/test_distributed_cpu.py)."""
from __future__ import annotations

import numpy as np


def shard_range(total_frames: int, world: int, rank: int):
    """Contiguous batch slice of rank `rank`: [r*B/G, (r+1)*B/G) with the remainder spread over the first ranks."""
    base, rem = divmod(total_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_model_bytes(path: str, dist, device):
    """Rank 0 reads the model file; every rank returns the same bytes. `dist` = torch.distributed (initialised) or None."""
    import torch
    if dist is None or dist.get_world_size() == 1:
        return open(path, "rb").read()
    rank = dist.get_rank()
    size = torch.zeros(1, dtype=torch.int64, device=device)
    blob = None
    if rank == 0:
        blob = torch.from_numpy(np.fromfile(path, dtype=np.uint8)).to(device)
        size[0] = blob.numel()
    dist.broadcast(size, 0)
    if rank != 0:
        blob = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    dist.broadcast(blob, 0)
    return blob.cpu().numpy().tobytes()


def max_over_ranks(seconds: float, dist, device) -> float:
    import torch
    if dist is None or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def max_over_ranks_vec(seconds, dist, device):
    """Element-wise max over ranks of a list of timings (one all-reduce)."""
    import torch
    if dist is None or dist.get_world_size() == 1 or not len(seconds):
        return list(seconds)
    t = torch.tensor(list(seconds), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.tolist()


def gather_counts(counts, dist, device):
    """Optional result collection: per-rank detection counts -> list over ranks (fixed-size records, no padding issue)."""
    import torch
    if dist is None or dist.get_world_size() == 1:
        return [counts]
    out = [torch.empty_like(counts) for _ in range(dist.get_world_size())]
    dist.all_gather(out, counts)
    return out
