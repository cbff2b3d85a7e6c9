"""Spectral sharding across the GPUs of one node.

Work items (wavelength, k-term) are independent (drt.f:425-561); the only coupling is
stdout1's weighted sums (drt.f:964-1054).  Ranks own contiguous blocks of spectral points;
the integrated accumulators are combined with ONE reduce (RCCL on GPUs, gloo in CPU tests).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(nwl: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of spectral-point indices owned by `rank`:
    ceil(nwl/world)-sized blocks, the last ones possibly shorter or empty."""
    per = (nwl + world - 1) // world
    lo = min(nwl, rank * per)
    hi = min(nwl, lo + per)
    return lo, hi


def reduce_accumulators(acc, dst: int = 0, group=None):
    """Sum the per-rank accumulator tensor onto rank `dst` (in place).  One collective."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return acc
