"""Spectral sharding across the GPUs of one node.

Work items (wavelength, k-term) are independent (drt.f:425-561); the only coupling is
stdout1's weighted sums (drt.f:964-1054).  Ranks own contiguous blocks of spectral points;
the integrated accumulators are combined with ONE reduce (RCCL on GPUs, gloo in CPU tests).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(nwl: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of spectral-point (or work-item) indices owned by `rank`:
    balanced blocks, the first nwl % world one longer.  Same rule as the C ABI's sbd_shard_range
    (include/sbdart_amd.h), which the fleet and the Fortran host use; tests pin the two together."""
    world = max(1, world)
    base, extra = divmod(nwl, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_range_points(point_of, rank: int, world: int) -> Tuple[int, int]:
    """Work items [lo, hi) of `rank` for a batch in compact form: shard_range's item boundaries, each moved UP to the
    next item that starts a spectral point (point_of non-decreasing) -- the k-terms of a point stay on one device.
    Same rule as the C ABI's sbd_shard_range_points (include/sbdart_amd.h); tests pin the two together."""
    n = len(point_of)

    def snap(i):
        while 0 < i < n and point_of[i] == point_of[i - 1]:
            i += 1
        return i
    lo, hi = shard_range(n, rank, world)
    return snap(lo), snap(hi)


def reduce_accumulators(acc, dst: int = 0, group=None):
    """Sum the per-rank accumulator tensor onto rank `dst` (in place).  One collective."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return acc
