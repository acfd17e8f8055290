"""Multi-GPU layout of the pose solve: images shard independently over ranks, no collective on
the data path, one final all-gather of the [B, 8] pose records.

The reference's only multi-GPU mechanism for this path is one OS process per category pinned
with CUDA_VISIBLE_DEVICES and files as the gather step (InferenceNeMo.sh:42-50,67-81); here it is
one rank per GPU under torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block of `n_items` owned by `rank`; blocks differ by at most one item."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_records(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather the per-rank [b_r, 8] record blocks (shard_range order) into [n_total, 8].

    Blocks may differ by one row, so every rank pads to the largest block: one collective of
    ceil(n/world) * 8 floats per rank (32 KB in total at B = 1024 -- latency-bound)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    width = local.shape[1]
    per = -(-n_total // world)
    # RCCL ("nccl") gathers device tensors in place; a gloo group (CPU tests, single-GPU rehearsals) goes
    # through host memory
    home = local.device
    work = torch.device('cpu') if (dist.get_backend(group) == 'gloo' and local.is_cuda) else home
    pad = torch.zeros((per, width), dtype=local.dtype, device=work)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per, width), dtype=local.dtype, device=work)
    dist.all_gather_into_tensor(out, pad, group=group)
    rows = []
    for r in range(world):
        s, e = shard_range(n_total, r, world)
        rows.append(out[r * per:r * per + (e - s)])
    return torch.cat(rows, dim=0).to(home)


def max_over_ranks(value: float, device=None, group=None) -> float:
    """Largest `value` over the ranks (bench.py times a step as long as its slowest rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    on_cpu = device is None or dist.get_backend(group) == 'gloo'
    t = torch.tensor([value], dtype=torch.float64, device='cpu' if on_cpu else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def fence(device=None, group=None):
    """Device synchronise + barrier + synchronise: the bracket bench.py puts around its timed region."""
    cuda = device is not None and torch.device(device).type == 'cuda'
    if cuda:
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)
        if cuda:
            torch.cuda.synchronize(device)


def solve_sharded(solve_fn, n_images: int, group=None) -> torch.Tensor:
    """solve_fn(start, stop) -> [stop-start, 8] records of this rank's images; returns all records
    on every rank (bit-identical to a single-rank run: each image's solve is independent)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    s, e = shard_range(n_images, rank, world)
    return gather_records(solve_fn(s, e), n_images, group)
