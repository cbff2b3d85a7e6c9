"""CPU-side checks of the product package: the C-ABI library loads and exports every
symbol include/sbdart_amd.h declares (no compute without a GPU), fails loudly without a
device, and the host-side helpers (workload, sharding) behave."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from sbdart_amd import _lib
    L = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "sbdart_amd.h")).read()
    declared = set(re.findall(r"\b(sbd_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(L, name), name
    assert set(_lib.EXPORTS) <= declared
    assert L.sbd_abi_version() == _lib.ABI_VERSION


def test_create_fails_loudly_without_gpu_or_with_bad_config():
    import torch
    from sbdart_amd import _lib
    from sbdart_amd.engine import DisortEngine, SbdError
    with pytest.raises(SbdError) as ei:
        DisortEngine(nlyr=2, nstr=5, nmom=6, temper=[250, 260, 270], umu0=0.5)   # odd NSTR
    assert ei.value.code == _lib.E_INVALID
    with pytest.raises(SbdError) as ei:
        DisortEngine(nlyr=2, nstr=8, nmom=10, temper=[250, 260, 270], umu0=0.5, lamber=False)   # no surface model named
    assert ei.value.code == _lib.E_INVALID
    with pytest.raises(SbdError) as ei:
        DisortEngine(nlyr=2, nstr=8, nmom=10, temper=[250, 260, 270], umu0=0.5, onlyfl=False, umu=[0.5, 0.2], phi=[0.0])
    assert ei.value.code == _lib.E_INVALID                                                   # UMU must ascend
    if not torch.cuda.is_available():
        with pytest.raises(SbdError) as ei:
            DisortEngine(nlyr=2, nstr=8, nmom=10, temper=[250, 260, 270], umu0=0.5)
        assert ei.value.code == _lib.E_NO_DEVICE   # no silent CPU fallback


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sbdart_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".f90", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in txt and "disort_oracle" not in txt and "liboracle" not in txt, f


def test_workload_is_seeded_and_physical():
    from sbdart_amd.workload import splitmix64, sw_sweep
    assert np.array_equal(splitmix64(1, 5), splitmix64(1, 5))
    a, b = sw_sweep(64, nstr=16), sw_sweep(64, nstr=16)
    assert np.array_equal(a.dtauc, b.dtauc) and a.nwork == int(a.nk.sum())
    assert not np.array_equal(a.dtauc, sw_sweep(64, nstr=16, shard=1).dtauc)
    assert a.dtauc.sum(axis=1).max() <= 50.0 + 1e-9
    assert a.ssalb.min() >= 0 and a.ssalb.max() < 1
    assert np.abs(a.pmom).max() <= 1.0 and np.all(a.pmom[:, :, 0] == 1.0)
    assert a.pmom.shape == (a.nwork, 33, 19)


def test_shard_range_partitions():
    """The C ABI's sbd_shard_range (what the fleet and the Fortran host shard with; pure host code,
    callable without a GPU) and its Python mirror: the same balanced contiguous blocks."""
    import ctypes as C
    from sbdart_amd import _lib
    from sbdart_amd.shard import shard_range
    L = _lib.load()
    lo, hi = C.c_int32(), C.c_int32()
    for nwl in (0, 1, 7, 751, 49152, 131254):
        for world in (1, 2, 3, 8):
            parts = [shard_range(nwl, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == nwl
            for (a0, a1), (b0, b1) in zip(parts, parts[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
            for r in range(world):
                L.sbd_shard_range(nwl, world, r, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == parts[r]
