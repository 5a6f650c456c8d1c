"""Multi-GPU: the batch of states shards trivially (SURVEY.md §8 e).  One process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm, over xGMI inside a node); rank g owns the contiguous state range [g·B/G, (g+1)·B/G); the model
constants are replicated (a few KB); there is NO collective in the data path.  The only exchange step is the gather of
`DynamicsResult` fields (v̇) for a caller that wants the whole batch in one place.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

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


# ---- the same exchange step through the C ABI (rbd_comm_* / rbd_gather: RCCL opened by librbd_hip itself, no torch.distributed) ----
class Comm:
    """RCCL communicator of include/rbd_hip.h.  `Comm.unique_id()` on one rank, the 128 bytes handed to the others by the caller,
    then `Comm(id, world, rank, device)` on every rank (one process per GPU)."""

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int = 0):
        import ctypes
        from . import _capi
        self._L = _capi.lib()
        self.handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        st = self._L.rbd_comm_create(ctypes.cast(buf, ctypes.c_void_p), world, rank, device, ctypes.byref(self.handle))
        if st != 0:
            raise _capi.RBDError(st, "rbd_comm_create", (self._L.rbd_comm_last_error() or b"").decode())
        self.world, self.rank, self.device = world, rank, device

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        from . import _capi
        buf = ctypes.create_string_buffer(128)
        st = _capi.lib().rbd_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p))
        if st != 0:
            raise _capi.RBDError(st, "rbd_comm_unique_id", (_capi.lib().rbd_comm_last_error() or b"").decode())
        return buf.raw

    def gather(self, shard: torch.Tensor, root: Optional[int] = None) -> Optional[torch.Tensor]:
        """All-gather (root None) or gather to `root` of equal-size shards (B_local, n) on the current stream."""
        import ctypes
        from . import _capi
        if shard.dtype not in (torch.float64, torch.float32):
            raise TypeError(f"Comm.gather moves float64 / float32 results, not {shard.dtype}")
        if not shard.is_cuda or shard.device.index != self.device:
            raise ValueError(f"the shard lives on {shard.device}, the communicator on cuda:{self.device}")
        shard = shard.contiguous()
        want = root is None or root == self.rank
        out = torch.empty((self.world * shard.shape[0],) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device) if want else None
        dt = _capi.F64 if shard.dtype == torch.float64 else _capi.F32
        st = self._L.rbd_gather(self.handle, dt, ctypes.c_void_p(shard.data_ptr()), ctypes.c_void_p(out.data_ptr() if out is not None else 0),
                                ctypes.c_int64(shard.numel()), -1 if root is None else int(root), ctypes.c_void_p(torch.cuda.current_stream(shard.device).cuda_stream))
        if st != 0:
            raise _capi.RBDError(st, "rbd_gather", (self._L.rbd_comm_last_error() or b"").decode())
        return out

    def gatherv(self, shard: torch.Tensor, rows: List[int], root: Optional[int] = None) -> Optional[torch.Tensor]:
        """... of shards with DIFFERENT numbers of states (`rows[r]` states on rank r — `shard_sizes(B, world)` for a batch that does not divide): the shards back
        to back in rank order, (sum(rows), n).  `rbd_gatherv`."""
        import ctypes
        from . import _capi
        if shard.dtype not in (torch.float64, torch.float32):
            raise TypeError(f"Comm.gatherv moves float64 / float32 results, not {shard.dtype}")
        if len(rows) != self.world or shard.shape[0] != rows[self.rank]:
            raise ValueError(f"rows must list every rank's states ({self.world} entries) and rows[{self.rank}] must be this shard's {shard.shape[0]}")
        if not shard.is_cuda or shard.device.index != self.device:
            raise ValueError(f"the shard lives on {shard.device}, the communicator on cuda:{self.device}")
        shard = shard.contiguous()
        per = int(shard.numel() // max(1, shard.shape[0])) if shard.shape[0] else int(torch.Size(shard.shape[1:]).numel())
        want = root is None or root == self.rank
        out = torch.empty((sum(rows),) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device) if want else None
        counts = (ctypes.c_int64 * self.world)(*[int(r) * per for r in rows])
        dt = _capi.F64 if shard.dtype == torch.float64 else _capi.F32
        st = self._L.rbd_gatherv(self.handle, dt, ctypes.c_void_p(shard.data_ptr() if shard.numel() else 0), ctypes.c_void_p(out.data_ptr() if out is not None else 0),
                                 counts, -1 if root is None else int(root), ctypes.c_void_p(torch.cuda.current_stream(shard.device).cuda_stream))
        if st != 0:
            raise _capi.RBDError(st, "rbd_gatherv", (self._L.rbd_comm_last_error() or b"").decode())
        return out

    def close(self):
        if self.handle:
            self._L.rbd_comm_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
