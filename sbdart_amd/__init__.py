"""sbdart_amd -- MI355X-native batched DISORT engine for SBDART's wavelength loop.

The product is the HIP library behind include/sbdart_amd.h (sbdart_amd/csrc); this package
is its ctypes mirror (engine.py), the record format shared with the Fortran host (records.py),
the synthetic benchmark workload (workload.py) and the spectral sharding helper (shard.py).
"""
from .engine import DisortEngine, DisortFleet, RetryNstr, SbdError  # noqa: F401

__all__ = ["DisortEngine", "DisortFleet", "RetryNstr", "SbdError"]
