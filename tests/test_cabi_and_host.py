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


def test_shard_range_points_cuts_between_spectral_points():
    """sbd_shard_range_points (a batch in compact form over several devices): shard_range's item boundaries moved up to
    the next item that starts a spectral point.  Contiguous cover, every cut at a point start, never more than one
    point's k-terms (<= 3 - 1 items here) away from the balanced boundary, C == Python mirror, empty shards allowed."""
    import ctypes as C
    from sbdart_amd import _lib
    from sbdart_amd.shard import shard_range, shard_range_points
    from sbdart_amd.workload import splitmix64
    L = _lib.load()
    lo, hi = C.c_int32(), C.c_int32()
    for npoint, seed in ((1, 1), (2, 2), (5, 3), (751, 4), (20000, 5)):
        nk = np.where(splitmix64(seed, npoint) < 0.835, 3, 1)
        po = np.ascontiguousarray(np.repeat(np.arange(npoint), nk), dtype=np.int32)
        po += 11                                                  # (a part of a run: block indices need not start at 0)
        n = len(po)
        for world in (1, 2, 3, 8):
            parts = [shard_range_points(po, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            for (a0, a1), (b0, b1) in zip(parts, parts[1:]):
                assert a1 == b0 and a0 <= a1
            for r, (a, b) in enumerate(parts):
                assert a == 0 or a == n or po[a] != po[a - 1]      # every cut starts a spectral point
                bl, bh = shard_range(n, r, world)
                assert 0 <= a - bl <= 2 and 0 <= b - bh <= 2
                L.sbd_shard_range_points(n, po.ctypes.data_as(C.c_void_p), world, r, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == (a, b)
    # one point with three k-terms on eight devices: the first shard whose boundary passes its end takes all of it
    po = np.zeros(3, dtype=np.int32)
    parts = [shard_range_points(po, r, 8) for r in range(8)]
    assert sum(b - a for a, b in parts) == 3 and max(b - a for a, b in parts) == 3


def test_flux_albedo_of_the_surface_models_on_the_host():
    """sbd_surface_flux_albedo (DREF, disort.f:5178-5284: what drt.f:478-484 needs for ISALB -7, -8, -9) runs on the
    host with the device's model functions: against the oracle's DREF for the three models over the incidence cosines,
    and the reference's argument check."""
    import ctypes as C
    import pyoracle
    from sbdart_amd import _lib
    L = _lib.load()
    L.sbd_surface_flux_albedo.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
    L.sbd_surface_flux_albedo.restype = C.c_int
    O = pyoracle.lib()
    O.sbdo_dref.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]
    O.sbdo_dref.restype = C.c_double
    cover = 2.951e-6 * 7.0 ** 3.52
    models = [(1, [7.0, cover, cover * 0.22, 0.5, 34.3, 0, 0, 0], [1.34, 1.2e-8, 0.03, 0.0]),
              (2, [0.6, 0.3, 0.4, 0.1, 0, 0, 0, 0], None),
              (3, [0.08, 0.03, 0.0005, 1.0, 2.0, 0, 0, 0], None)]
    worst = 0.0
    for ibdrf, bpar, bitem in models:
        bp = (C.c_double * 8)(*bpar)
        bi = (C.c_double * 4)(*bitem) if bitem else None
        for mu in (-0.3, 0.0, 0.01, 0.3, 0.7660444431, 1.0):          # (below zero: the sun under the horizon, as drt.f passes it)
            out = C.c_double(0.0)
            assert L.sbd_surface_flux_albedo(ibdrf, bp, bi, mu, C.byref(out)) == _lib.OK
            want = O.sbdo_dref(ibdrf, bp, bi, mu)
            assert mu < 0 or 0.0 <= want <= 1.0
            worst = max(worst, abs(out.value - want) / max(abs(want), 1e-300))
            assert abs(out.value - want) <= 1e-13 * max(abs(want), 1e-3), (ibdrf, mu, out.value, want)
        out = C.c_double(0.0)
        assert L.sbd_surface_flux_albedo(ibdrf, bp, bi, 1.5, C.byref(out)) != _lib.OK        # DREF--input argument error(s)
    assert L.sbd_surface_flux_albedo(1, (C.c_double * 8)(*models[0][1]), None, 0.5, C.byref(C.c_double())) != _lib.OK
    print("worst relative difference from the oracle's DREF: %.2e" % worst)


def test_status_bits_agree_across_the_header_the_bindings_and_the_oracle():
    """SBD_ST_* of include/sbdart_amd.h, the Python and Fortran bindings' copies, and the oracle's status bits (the
    parity tests compare status WORDS between engine and oracle): one table, five places."""
    import re
    from conftest import ROOT
    from sbdart_amd import _lib
    hdr = dict((m.group(1), int(m.group(2), 16)) for m in
               re.finditer(r"#define\s+SBD_ST_(\w+)\s+(0x[0-9a-fA-F]+)", open(os.path.join(ROOT, "include", "sbdart_amd.h")).read()))
    assert hdr == {"WARN_SOLVE0": 1, "WARN_UPBEAM": 2, "WARN_UPISOT": 4, "ERR_EIGEN": 8, "RETRY_NSTR": 16, "ERR_INPUT": 32,
                   "WARN_PLKAVG": 64, "WARN_PLKCONV": 128}
    for name, val in hdr.items():
        assert getattr(_lib, "ST_" + name) == val, name
    f90 = open(os.path.join(ROOT, "sbdart_amd", "fortran", "sbd_engine_mod.f90")).read()
    for name, val in hdr.items():
        assert re.search(r"SBD_ST_%s\s*=\s*%d\b" % (name, val), f90), name
    ohdr = dict((m.group(1), int(m.group(2), 16)) for m in
                re.finditer(r"#define\s+SBDO_(\w+)\s+(0x[0-9a-fA-F]+)", open(os.path.join(ROOT, "oracle", "disort_oracle.h")).read()))
    same = {"WARN_SOLVE0": "WARN_SOLVE0_RCOND", "WARN_UPBEAM": "WARN_UPBEAM_RCOND", "WARN_UPISOT": "WARN_UPISOT_RCOND",
            "ERR_EIGEN": "ERR_ASYMTX", "RETRY_NSTR": "RETRY_NSTR", "ERR_INPUT": "ERR_INPUT", "WARN_PLKAVG": "WARN_PLKAVG",
            "WARN_PLKCONV": "WARN_PLKCONV"}
    for name, oname in same.items():
        assert ohdr[oname] == hdr[name], name


@pytest.mark.parametrize("namelist", [
    "idatm=4 wlinf=.5 wlsup=.7 wlinc=.01 nstr=8 iout=10 tcloud=5 zcloud=2 iaer=1 vis=20",
    "idatm=2 wlinf=3.5 wlsup=12 wlinc=-.01 nstr=16 iout=20 nzen=3 uzen=0,40,80 nphi=2 phi=0,90 isalb=8 sc=.6,.2,.1,.06",
])
def test_work_item_files_survive_a_round_trip_through_the_host(tmp_path, namelist):
    """The work-item file a first phase writes (`sbdart_amd --serve` / `--batch`: the band model's items, layer arrays from its
    batch arrays) read back by the host and written again from the records' own arrays: byte for byte the same file.  No GPU:
    SBD_DUMP_OPTICS stops before the engine."""
    import subprocess
    host = os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")
    if not os.access(host, os.X_OK):
        pytest.skip("Fortran host not built")
    d = str(tmp_path)
    with open(os.path.join(d, "INPUT"), "w") as f:
        f.write(f"\n &INPUT\n {namelist}\n /\n")
    a, b = os.path.join(d, "a.sbdrec"), os.path.join(d, "b.sbdrec")
    subprocess.run([host], cwd=d, env=dict(os.environ, SBD_OPTICS=os.path.join(d, "none"), SBD_DUMP_OPTICS=a), capture_output=True)
    subprocess.run([host], cwd=d, env=dict(os.environ, SBD_OPTICS=a, SBD_DUMP_OPTICS=b), capture_output=True)
    assert os.path.getsize(a) > 1000
    assert open(a, "rb").read() == open(b, "rb").read()
