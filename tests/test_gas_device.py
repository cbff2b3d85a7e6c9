"""The gas part of the band model in the engine (sbdart_amd/csrc/sbd_gas.hpp: one source for host and device).

CPU: sbd_gas_terms_host -- that source compiled for the host, with the host's libm -- returns BIT FOR BIT the number of
k-terms, their weights and every layer's gas optical depth the Fortran host's band model computes for the same run
(SBD_DUMP_MIX), and that band model is bit-equal to the live reference (tests/test_band_model.py): the kernel's source
is pinned through the chain reference == Fortran host == C++ host evaluation.
GPU: the kernel against the host evaluation of the same source.  exp / log / log10 / pow come from the device math
library there: the optical depths agree to GAS_RTOL relative (stated bound; measured worst in
tests/golden/measured_errors.json), the k-term counts exactly."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")
GAS_RTOL = 2e-12      # device vs host evaluation of the same formulas, per layer depth, relative to the column's largest


def read_mix_dump_with_gas(path):
    b = open(path, "rb").read()
    h = np.frombuffer(b[:44], dtype=np.int32)
    nz, nch, npt, nterm, nrec = int(h[0]), int(h[1]), int(h[2]), int(h[3]), int(h[10])
    assert nz > 0
    o = 44
    lay = np.frombuffer(b[o:o + 8 * nz * nch * npt]).reshape(npt, nch, nz).copy(); o += 8 * nz * nch * npt
    dtaug = np.frombuffer(b[o:o + 8 * nz * nrec]).reshape(nrec, nz).copy(); o += 8 * nz * nrec
    po = np.frombuffer(b[o:o + 4 * nrec], dtype=np.int32).copy(); o += 4 * nrec
    has = int(np.frombuffer(b[o:o + 4], dtype=np.int32)[0]); o += 4
    assert has == 1
    kdist = int(np.frombuffer(b[o:o + 4], dtype=np.int32)[0]); o += 4
    amu = np.frombuffer(b[o:o + 16]).copy(); o += 16
    xo4 = float(np.frombuffer(b[o:o + 8])[0]); o += 8
    uu = np.frombuffer(b[o:o + 8 * 63 * nz]).reshape(nz, 63).copy(); o += 8 * 63 * nz
    z = np.frombuffer(b[o:o + 8 * nz]).copy(); o += 8 * nz
    wl = np.frombuffer(b[o:o + 8 * npt]).copy(); o += 8 * npt
    rec = np.frombuffer(b[o:o + 16 * nrec], dtype=np.dtype([("kd", "<i4"), ("nk", "<i4"), ("wt", "<f8")])).copy()
    return dict(nz=nz, nch=nch, lay=lay, dtaug=dtaug, point_of=po, kdist=kdist, amu=amu, xo4=xo4, uu=uu, z=z, wl=wl,
                kd=rec["kd"], nk=rec["nk"], wt=rec["wt"], family=[int(x) for x in h[4:4 + nterm]])


def gas_model(d):
    from sbdart_amd import _lib
    img = np.fromfile(_lib.TABLES_FILE, dtype=np.uint8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    g = _lib.GasModel(d["nz"], d["kdist"], vp(d["uu"]), vp(d["z"]), float(d["amu"][0]), float(d["amu"][1]), d["xo4"],
                      vp(img), img.nbytes)
    return g, img


def host_gas_terms(d):
    from sbdart_amd import _lib
    L = _lib.load()
    g, img = gas_model(d)
    npt, nz = len(d["wl"]), d["nz"]
    nk = np.zeros(npt, dtype=np.int32)
    wt = np.zeros((npt, 3))
    fail = np.zeros(npt, dtype=np.int32)
    slots = np.zeros((npt, 3, nz))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.sbd_gas_terms_host(C.byref(g), nz, npt, vp(d["wl"]), vp(d["lay"]), d["nch"], vp(nk), vp(wt), vp(fail), vp(slots))
    assert rc == 0
    return nk, wt, fail, slots


def dump(tmp_path, namelist):
    d = str(tmp_path)
    with open(os.path.join(d, "INPUT"), "w") as f:
        f.write(f"\n &INPUT\n {namelist}\n /\n")
    mixf = os.path.join(d, "mix.bin")
    subprocess.run([HOST], cwd=d, env=dict(os.environ, SBD_OPTICS=os.path.join(d, "none"), SBD_DUMP_MIX=mixf),
                   capture_output=True, text=True)
    return read_mix_dump_with_gas(mixf)


GAS_RUNS = [
    "idatm=4 wlinf=.25 wlsup=4.0 wlinc=.01 nstr=4 iout=10 sza=30",                  # BASELINE configs[1], every other point
    "idatm=6 wlinf=4 wlsup=80 wlinc=-.01 nstr=16 tcloud=10 zcloud=1 nre=8 iout=10 sza=95",      # configs[2]: thermal, no sun
    "idatm=6 wlinf=.25 wlsup=100 wlinc=40 nstr=32 ngrid=50 iout=10 sza=30",          # configs[4]'s atmosphere, 50 layers
    "idatm=2 wlinf=.3 wlsup=3.0 wlinc=.03 kdist=0 sza=60 nstr=4 iout=10 xco2=800 xch4=3",
    "idatm=1 wlinf=.3 wlsup=3.0 wlinc=.03 kdist=1 sza=0 nstr=4 iout=10 uw=4 uo3=.2",
    "idatm=5 wlinf=.3 wlsup=5.0 wlinc=.05 kdist=2 sza=75 nstr=4 iout=10 xo4=2 xn2o=.4",
    "idatm=3 wlinf=.2 wlsup=.35 wlinc=.0005 sza=45 nstr=4 iout=10",                  # Hartley / Huggins, Herzberg, Schumann-Runge edge
    "idatm=4 wlinf=.5 wlsup=2.5 wlinc=.02 sza=50 nstr=8 iout=10 tcloud=20 zcloud=3 iaer=1 vis=10",   # the roll-off under a thick cloud
]


@pytest.mark.parametrize("namelist", GAS_RUNS)
def test_gas_source_on_the_host_is_the_band_model_bit_for_bit(tmp_path, namelist):
    if not os.access(HOST, os.X_OK):
        pytest.skip("Fortran host not built")
    d = dump(tmp_path, namelist)
    nk, wt, fail, slots = host_gas_terms(d)
    assert not fail.any()
    npt = len(d["wl"])
    # the Fortran host's items, by point and k-term
    assert np.array_equal(np.bincount(d["point_of"], minlength=npt), nk), "k-term counts"
    got = slots[d["point_of"], d["kd"] - 1]
    assert np.array_equal(d["nk"], nk[d["point_of"]])
    assert np.array_equal(got, d["dtaug"]), float(np.abs(got - d["dtaug"]).max())
    assert np.array_equal(wt[d["point_of"], d["kd"] - 1], d["wt"])
    assert (nk == 3).any() or "kdist=0" in namelist or "wlsup=.35" in namelist


@pytest.mark.gpu
@pytest.mark.parametrize("namelist,devices", [(GAS_RUNS[0], [0]), (GAS_RUNS[1], [0, 0]), (GAS_RUNS[2], [0, 0, 0]),
                                               (GAS_RUNS[5], [0]), (GAS_RUNS[6], [0]), (GAS_RUNS[7], [0, 0])])
def test_gas_kernel_against_its_host_evaluation(tmp_path, namelist, devices):
    from ratchet import ratchet
    from sbdart_amd import _lib
    from sbdart_amd.engine import DisortFleet
    d = dump(tmp_path, namelist)
    nk_h, wt_h, fail_h, slots_h = host_gas_terms(d)
    nz, npt = d["nz"], len(d["wl"])
    with DisortFleet(nlyr=nz, nstr=4, nmom=6, temper=np.linspace(220, 290, nz + 1), umu0=0.5, onlyfl=True,
                     level_out=[0, nz], devices=devices) as fl:
        g, img = gas_model(d)                                   # (img: the table image g points into)
        nk, wt, fail, slots = fl.gas_terms(g, d["wl"], d["lay"], want_depths=True)
    assert np.array_equal(nk, nk_h) and np.array_equal(fail, fail_h)
    scale = np.abs(slots_h).max(axis=2, keepdims=True) + 1e-300
    err = float((np.abs(slots - slots_h) / scale).max())
    werr = float(np.abs(wt - wt_h).max())
    assert err <= GAS_RTOL and werr <= GAS_RTOL, (err, werr)
    ratchet("gas_kernel_vs_host/" + "_".join(namelist.split()[:3]), max(err, werr))


@pytest.mark.gpu
def test_solve_with_the_gas_depths_left_on_the_devices(tmp_path):
    """sbd_fleet_gas_terms then sbd_fleet_solve_mix_host with dtaug = NULL and the items' k-terms: the gas depths never
    cross PCIe.  Bit-equal to the same solve fed with the depths copied back from the device (one device and three
    engines: an item goes to the engine that holds its point)."""
    from sbdart_amd.engine import DisortFleet
    d = dump(tmp_path, GAS_RUNS[7])
    nz, npt = d["nz"], len(d["wl"])
    nstr = 8
    temper = np.linspace(220, 290, nz + 1)
    lo = 1.0e4 / (d["wl"] + 0.01)
    hi = 1.0e4 / (d["wl"] - 0.01)
    res = {}
    for devices in ([0], [0, 0, 0]):
        with DisortFleet(nlyr=nz, nstr=nstr, nmom=nstr + 2, temper=temper, umu0=float(np.cos(np.deg2rad(50.0))), onlyfl=True,
                         level_out=[0, nz], devices=devices) as fl:
            g, img = gas_model(d)
            nk, wt, fail, depths = fl.gas_terms(g, d["wl"], d["lay"], want_depths=True)
            po = np.repeat(np.arange(npt, dtype=np.int32), nk)
            kt = np.concatenate([np.arange(k, dtype=np.int32) for k in nk])
            w = wt[po, kt]
            a = fl.solve_mix(po, None, d["lay"], d["family"], lo, hi, 1.0, 0.2, 0, weight=w, kterm=kt)
            b = fl.solve_mix(po, depths[po, kt], d["lay"], d["family"], lo, hi, 1.0, 0.2, 0, weight=w)
            # ABI v7: residency of the layer blocks is the caller's explicit statement (the token), never inferred
            token = fl.lay_token
            assert token != 0
            c = fl.solve_mix(po, None, d["lay"], d["family"], lo, hi, 1.0, 0.2, 0, weight=w, kterm=kt, lay_token=token)
            assert np.array_equal(a[0], c[0]) and np.array_equal(a[2], c[2])
            # the hazard ADVICE r05 names: the caller edits `lay` in place between the gas call and a solve.  Without a
            # token the edited blocks are staged (results change); with the token the device copy answers (unchanged).
            lay2 = d["lay"].copy()
            lay2[:, 2, :] *= 3.0                                   # three times the Rayleigh depth
            lay2[:, 3, :] = lay2[:, 0, :] * 0 + d["lay"][:, 3, :] + 2.0 * d["lay"][:, 2, :]
            e_ = fl.solve_mix(po, None, lay2, d["family"], lo, hi, 1.0, 0.2, 0, weight=w, kterm=kt)
            assert not np.array_equal(a[0], e_[0])
            f_ = fl.solve_mix(po, None, lay2, d["family"], lo, hi, 1.0, 0.2, 0, weight=w, kterm=kt, lay_token=token)
            assert np.array_equal(a[0], f_[0])
            from sbdart_amd.engine import SbdError
            with pytest.raises(SbdError):                         # a token that is not the fleet's current one
                fl.solve_mix(po, None, d["lay"], d["family"], lo, hi, 1.0, 0.2, 0, weight=w, kterm=kt, lay_token=token + 12345)
            with pytest.raises(SbdError):                         # a token beside explicit gas depths
                fl.solve_mix(po, depths[po, kt], d["lay"], d["family"], lo, hi, 1.0, 0.2, 0, weight=w, lay_token=token)
            if (nk == 1).any():                                    # a k-term the point does not have
                bad = kt.copy()
                bad[np.flatnonzero(nk[po] == 1)[0]] = 1
                with pytest.raises(SbdError):
                    fl.solve_mix(po, None, d["lay"], d["family"], lo, hi, 1.0, 0.2, 0, weight=w, kterm=bad)
            fl.gas_terms(g, d["wl"], d["lay"])                     # a new gas call: the old token is stale
            assert fl.lay_token not in (0, token)
            with pytest.raises(SbdError):
                fl.solve_mix(po, None, d["lay"], d["family"], lo, hi, 1.0, 0.2, 0, weight=w, kterm=kt, lay_token=token)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and (a[2] == 0).all()
        assert np.allclose(a[3], b[3], rtol=1e-13, atol=0)
        assert np.isfinite(a[0]).all() and np.abs(a[0]).max() > 0
        res[len(devices)] = a
    assert np.array_equal(res[1][0], res[3][0]) and np.array_equal(res[1][2], res[3][2])


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_gas_source_on_the_host_bit_for_bit_on_random_runs(tmp_path, seed):
    """The pin above over seeded random atmospheres, absorber amounts, KDIST policies, zenith angles and grids (12 runs per
    seed; a run the band model's input checks stop is skipped)."""
    import random
    if not os.access(HOST, os.X_OK):
        pytest.skip("Fortran host not built")
    rnd = random.Random(seed)
    pick = lambda *a: rnd.choice(a)
    compared = 0
    for it in range(12):
        lo = pick(.2, .25, .3, .4, .55, 1., 2., 3.5, 5., 8., 15.)
        hi = min(lo * pick(1.1, 1.5, 2., 4.), 95.)
        p = ["idatm=%d" % pick(1, 2, 3, 4, 5, 6), "wlinf=%g wlsup=%g wlinc=%g" % (lo, hi, pick(.01 * lo, -.01, -.003, .02 * lo)),
             "sza=%g" % pick(0, 30, 60, 80, 88, 95), "nstr=4 iout=10", "kdist=%d" % pick(0, 1, 2, 3, 3, 3)]
        if rnd.random() < .3: p.append("uw=%g" % pick(.1, .5, 2, 4, 8))
        if rnd.random() < .3: p.append("uo3=%g" % pick(.1, .2, .35, .5))
        if rnd.random() < .3: p.append("xco2=%g xch4=%g" % (pick(0, 280, 420, 800, 5000), pick(0, .8, 1.8, 3)))
        if rnd.random() < .3: p.append("xo4=%g xn2o=%g" % (pick(0, 1, 2), pick(0, .1, .4)))
        if rnd.random() < .2: p.append("xco=%g xno2=%g xso2=%g" % (pick(0, .5, 5), pick(0, 1e-4, 1e-2), pick(0, 1e-3, .1)))
        if rnd.random() < .2: p.append("sclh2o=%g" % pick(1., 2.5))
        if rnd.random() < .2: p.append("pbar=%g" % pick(800, 900, 1030))
        if rnd.random() < .3: p.append("tcloud=%g zcloud=%g" % (pick(1, 20, 80), pick(1, 4, 9)))
        if rnd.random() < .3: p.append("iaer=%d vis=%g" % (pick(1, 2, 3, 4), pick(5, 23)))
        if rnd.random() < .25: p.append("ngrid=%d zgrid1=%g zgrid2=%g" % (pick(20, 40, 65), pick(.5, 1, 2), pick(10, 30)))
        sub = tmp_path / f"r{it}"
        sub.mkdir()
        try:
            d = dump(sub, " ".join(p))
        except (FileNotFoundError, AssertionError):
            continue
        nk, wt, fail, slots = host_gas_terms(d)
        npt = len(d["wl"])
        assert not fail.any(), " ".join(p)
        assert np.array_equal(np.bincount(d["point_of"], minlength=npt), nk), " ".join(p)
        assert np.array_equal(slots[d["point_of"], d["kd"] - 1], d["dtaug"]), " ".join(p)
        assert np.array_equal(wt[d["point_of"], d["kd"] - 1], d["wt"]), " ".join(p)
        compared += 1
    assert compared >= 8, compared
