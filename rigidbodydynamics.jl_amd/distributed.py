"""Multi-GPU: the batch of states shards trivially (SURVEY.md §8 e).  One process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm, over xGMI inside a node); rank g owns the contiguous state range [g·B/G, (g+1)·B/G); the model
constants are replicated (a few KB); there is NO collective in the data path.  The only exchange step is the gather of
`DynamicsResult` fields (v̇) for a caller that wants the whole batch in one place.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of a batch of B states for `rank` of `world`; sizes differ by at most one state."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(B: int, world: int):
    return [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]


def gather_results(local: torch.Tensor, B: int, group: Optional[dist.ProcessGroup] = None, dst: Optional[int] = None) -> Optional[torch.Tensor]:
    """Gather per-rank result shards (AOS: (B_local, n), one state per row) into the full (B, n) tensor.

    dst=None: all-gather (every rank gets the full result); dst=r: only rank r receives it (others return None).
    Equal shards use all_gather_into_tensor / gather (one RCCL collective over xGMI); ragged shards are padded to the
    largest shard and trimmed (states are independent, so padding rows are just dropped)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(B, world)
    assert local.shape[0] == sizes[rank], (local.shape, sizes[rank])
    n = local.shape[1]
    smax = max(sizes)
    if smax != local.shape[0]:
        pad = torch.zeros((smax, n), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        local = pad
    local = local.contiguous()
    if dst is None:
        out = torch.empty((world * smax, n), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
    else:
        out = torch.empty((world * smax, n), dtype=local.dtype, device=local.device) if rank == dst else None
        parts = list(out.split(smax)) if rank == dst else None
        dist.gather(local, parts, dst=dst, group=group)
        if rank != dst:
            return None
    if all(s == smax for s in sizes):
        return out
    return torch.cat([out[r * smax: r * smax + sizes[r]] for r in range(world)], dim=0)
