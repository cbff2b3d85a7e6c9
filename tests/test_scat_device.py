"""The scatterers' part of the band model in the engine (sbdart_amd/csrc/sbd_scat.hpp: one source for host and device).

CPU: sbd_scatter_blocks_host -- that source compiled for the host, with the host's libm -- returns BIT FOR BIT the layer
blocks (DTAUC, DTAUA, DTAUR, scattering depth, asymmetry factor and the two factors of every scattering term) the Fortran
host's band model computes for the same run (SBD_DUMP_MIX), and that band model is bit-equal to the live reference
(tests/test_band_model.py): the kernel's source is pinned through reference == Fortran host == C++ host evaluation.
GPU: the kernel (sbd_fleet_point_terms) against the host evaluation of the same source.  log / pow come from the device
math library there: the blocks agree to SCAT_RTOL relative to each channel's largest value of the point (stated bound;
measured worst in tests/golden/measured_errors.json); and the Fortran host's results with the blocks made on the device
against the same run with the host's blocks."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_gas_device import gas_model, read_mix_dump_with_gas

HOST = os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")
SCAT_RTOL = 2e-13     # device vs host evaluation of the same formulas, per channel, relative to the channel's largest


def read_mix_dump_with_scat(path):
    """SBD_DUMP_MIX's file including its last section, the scatterers' model (sbdart_amd_main.f90)."""
    d = read_mix_dump_with_gas(path)
    b = open(path, "rb").read()
    nz, nch, npt, nrec = d["nz"], d["nch"], len(d["wl"]), len(d["point_of"])
    o = 44 + 8 * nz * nch * npt + 8 * nz * nrec + 4 * nrec + 4 + 4 + 16 + 8 + 8 * 63 * nz + 8 * nz + 8 * npt + 16 * nrec
    i4 = lambda n: np.frombuffer(b[o:o + 4 * n], dtype=np.int32).copy()
    f8 = lambda n: np.frombuffer(b[o:o + 8 * n]).copy()
    has = int(i4(1)[0]); o += 4
    d["scat_ok"] = has == 1
    if not has:
        return d
    s = {}
    for k in ("z", "p", "t"):
        s[k] = f8(nz); o += 8 * nz
    s["xrsc"] = float(f8(1)[0]); o += 8
    h = i4(7); o += 28
    s["cloud_term"], s["cld_nslot"], s["cld_layer"] = int(h[0]), int(h[1]), h[2:7]
    for k in ("cld_tcloud", "cld_lwp", "cld_nre"):
        s[k] = f8(5); o += 40
    h = i4(3); o += 12
    s["iaer"], s["nosct"], s["aer_nwl"] = int(h[0]), int(h[1]), int(h[2])
    s["abaer"] = float(f8(1)[0]); o += 8
    if s["iaer"] != 0:
        n = s["aer_nwl"]
        for k in ("aer_wl", "aer_ext", "aer_absb", "aer_asym"):
            s[k] = f8(n); o += 8 * n
        s["aer_column"] = f8(nz); o += 8 * nz
    h = i4(11); o += 44
    s["nstrat"], s["jaer"], s["strat_layer"] = int(h[0]), h[1:6], h[6:11]
    s["taerst"] = f8(5); o += 40
    assert o == len(b), (o, len(b))
    d["scat"] = s
    return d


def scat_model(d):
    from sbdart_amd import _lib
    s = d["scat"]
    img = np.fromfile(_lib.TABLES_FILE, dtype=np.uint8)
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    i5 = lambda a: (C.c_int32 * 5)(*[int(x) for x in a])
    f5 = lambda a: (C.c_double * 5)(*[float(x) for x in a])
    m = _lib.ScatModel(d["nz"], vp(s["z"]), vp(s["p"]), vp(s["t"]), s["xrsc"], s["cloud_term"], s["cld_nslot"],
                       i5(s["cld_layer"]), f5(s["cld_tcloud"]), f5(s["cld_lwp"]), f5(s["cld_nre"]),
                       s["iaer"], s["nosct"], s["aer_nwl"], vp(s.get("aer_wl")), vp(s.get("aer_ext")), vp(s.get("aer_absb")),
                       vp(s.get("aer_asym")), s["abaer"], vp(s.get("aer_column")), s["nstrat"], i5(s["jaer"]),
                       i5(s["strat_layer"]), f5(s["taerst"]), vp(img), img.nbytes)
    return m, img


def host_blocks(d):
    from sbdart_amd import _lib
    L = _lib.load()
    m, img = scat_model(d)
    npt = len(d["wl"])
    out = np.full((npt, d["nch"], d["nz"]), np.nan)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.sbd_scatter_blocks_host(C.byref(m), npt, vp(d["wl"]), d["nch"], vp(out))
    assert rc == 0
    return out


def dump(tmp_path, namelist):
    d = str(tmp_path)
    with open(os.path.join(d, "INPUT"), "w") as f:
        f.write(f"\n &INPUT\n {namelist}\n /\n")
    mixf = os.path.join(d, "mix.bin")
    subprocess.run([HOST], cwd=d, env=dict(os.environ, SBD_OPTICS=os.path.join(d, "none"), SBD_DUMP_MIX=mixf),
                   capture_output=True, text=True)
    return read_mix_dump_with_scat(mixf)


SCAT_RUNS = [
    "idatm=4 wlinf=.25 wlsup=4.0 wlinc=.01 nstr=4 iout=10 sza=30",                               # clear sky: Rayleigh alone
    "idatm=6 wlinf=4 wlsup=80 wlinc=-.01 nstr=16 tcloud=10 zcloud=1 nre=8 iout=10 sza=95",         # configs[2]: one water cloud, thermal
    "idatm=4 wlinf=.5 wlsup=2.5 wlinc=.02 sza=50 nstr=8 iout=10 tcloud=20 zcloud=3 iaer=1 vis=10",   # cloud + rural aerosol
    "idatm=2 wlinf=.3 wlsup=3.5 wlinc=.02 sza=40 nstr=8 iout=10 iaer=2 vis=5 rhaer=.9 xrsc=.5",     # urban aerosol, humid; XRSC
    "idatm=1 wlinf=.3 wlsup=20 wlinc=-.005 sza=20 nstr=8 iout=10 iaer=3 tbaer=.4 jaer=1,3,4 zaer=15,20,25 taerst=.02,.05,.01",
    "idatm=3 wlinf=.3 wlsup=4 wlinc=.01 sza=20 nstr=8 iout=10 zcloud=2,-5,8 tcloud=6,.5,3 nre=6,12,-40",   # a graded cloud and an ice cloud
    "idatm=5 wlinf=.4 wlsup=12 wlinc=-.01 sza=60 nstr=6 iout=10 zcloud=1,9 lwp=80,15 nre=10,-25 iaer=4 vis=30 imoma=1",
    "idatm=4 wlinf=.25 wlsup=3 wlinc=.01 sza=10 nstr=4 iout=10 iaer=5 wlbaer=.3,.55,1,2.5 qbaer=1.4,1,.5,.1 wbaer=.95,.9,.8,.4 gbaer=.75,.7,.6,.5 tbaer=.3",
    "idatm=4 wlinf=.25 wlsup=3 wlinc=.01 sza=10 nstr=4 iout=10 iaer=5 wlbaer=.55 qbaer=1 wbaer=.9 gbaer=.7 tbaer=.3 abaer=1.3 nosct=1",
    "idatm=2 wlinf=.3 wlsup=3 wlinc=.01 sza=10 nstr=4 iout=10 iaer=1 vis=15 nosct=3 jaer=2 zaer=18 taerst=.1 abaer=1.1",
    "idatm=6 wlinf=.25 wlsup=100 wlinc=40 nstr=32 ngrid=50 iout=10 sza=30 tcloud=5 zcloud=2 iaer=1 vis=23",   # configs[4]'s atmosphere with a cloud
]


@pytest.mark.parametrize("namelist", SCAT_RUNS)
def test_scatter_source_on_the_host_is_the_band_model_bit_for_bit(tmp_path, namelist):
    if not os.access(HOST, os.X_OK):
        pytest.skip("Fortran host not built")
    d = dump(tmp_path, namelist)
    assert d["scat_ok"]
    got = host_blocks(d)
    same = got == d["lay"]
    if not same.all():
        p, ch, l = [int(x[0]) for x in np.nonzero(~same)]
        raise AssertionError(f"{int((~same).sum())} of {same.size} differ; first: point {p} (wl {d['wl'][p]}) channel {ch} layer {l}: "
                             f"{got[p, ch, l]!r} vs {d['lay'][p, ch, l]!r}; channels {sorted(set(np.nonzero(~same)[1].tolist()))}")


def test_runs_the_source_does_not_cover_keep_the_host_path(tmp_path):
    if not os.access(HOST, os.X_OK):
        pytest.skip("Fortran host not built")
    d = str(tmp_path)
    with open(os.path.join(d, "usrcld.dat"), "w") as f:
        f.write("0 8 0 -1 1\n40 6 0 -1 1\n120 9 0 -1 .6\n")
    with open(os.path.join(d, "INPUT"), "w") as f:
        f.write("\n &INPUT\n idatm=4 wlinf=.5 wlsup=.6 wlinc=.01 nstr=4 iout=10 nre=0\n /\n")
    mixf = os.path.join(d, "mix.bin")
    subprocess.run([HOST], cwd=d, env=dict(os.environ, SBD_OPTICS=os.path.join(d, "none"), SBD_DUMP_MIX=mixf),
                   capture_output=True, text=True)
    got = read_mix_dump_with_scat(mixf)
    assert not got["scat_ok"]


@pytest.mark.gpu
@pytest.mark.parametrize("namelist,devices", [(SCAT_RUNS[0], [0]), (SCAT_RUNS[1], [0, 0]), (SCAT_RUNS[2], [0]), (SCAT_RUNS[4], [0, 0, 0]),
                                               (SCAT_RUNS[5], [0]), (SCAT_RUNS[6], [0]), (SCAT_RUNS[7], [0]), (SCAT_RUNS[10], [0])])
def test_scatter_kernel_against_its_host_evaluation(tmp_path, namelist, devices):
    from ratchet import ratchet
    from sbdart_amd.engine import DisortFleet
    d = dump(tmp_path, namelist)
    want = host_blocks(d)
    nz = d["nz"]
    with DisortFleet(nlyr=nz, nstr=4, nmom=6, temper=np.linspace(220, 290, nz + 1), umu0=0.5, onlyfl=True,
                     level_out=[0, nz], devices=devices) as fl:
        g, img = gas_model(d)
        m, img2 = scat_model(d)
        nk, wt, fail, depths, blocks = fl.point_terms(g, m, d["wl"], d["nch"], want_depths=True, want_blocks=True)
        assert fl.lay_token != 0
        # the gas terms of the same call read the device-made blocks: against the call that is given the host's
        nk2, wt2, fail2, depths2 = fl.gas_terms(g, d["wl"], want, want_depths=True)
    assert np.array_equal(nk, nk2) and np.array_equal(fail, fail2)
    scale = np.abs(want).max(axis=2, keepdims=True) + 1e-300
    err = float((np.abs(blocks - want) / scale).max())
    assert err <= SCAT_RTOL, err
    assert (blocks == 0).sum() == (want == 0).sum()                  # (absent scatterers are exact zeros on both sides)
    gscale = np.abs(depths2).max(axis=2, keepdims=True) + 1e-300
    assert float((np.abs(depths - depths2) / gscale).max()) <= 2e-12
    ratchet("scatter_kernel_vs_host/" + "_".join(namelist.split()[:3]), err)


@pytest.mark.gpu
@pytest.mark.parametrize("namelist", [
    "idatm=4 wlinf=.25 wlsup=4.0 wlinc=.0025 nstr=4 iout=10 sza=30 isalb=6 tcloud=8 zcloud=2 iaer=1 vis=23",
    "idatm=2 wlinf=.3 wlsup=20 wlinc=-.002 nstr=8 iout=10 sza=50 iaer=3 tbaer=.3 jaer=1,3 zaer=15,22 taerst=.02,.05",
    "idatm=4 wlinf=.3 wlsup=3 wlinc=.005 nstr=8 iout=1 sza=30 zcloud=2,-5 tcloud=6,.5 nre=6,12",
])
def test_fortran_host_with_device_blocks_against_host_blocks(tmp_path, namelist):
    """The whole program: layer blocks made on the device (default for runs whose gas terms are there) against the same run
    with SBD_HOST_SCAT=1.  The k-term counts decide the number of solves: equal; the run's sums at full precision
    (SBD_SUMS_FILE) to 1e-9 relative, the printed numbers to one unit of their last printed digit."""
    if not os.access(HOST, os.X_OK):
        pytest.skip("Fortran host not built")
    outs, sums = [], []
    for env in ({"SBD_DEVICE_GAS": "1"}, {"SBD_DEVICE_GAS": "1", "SBD_HOST_SCAT": "1"}):
        d = os.path.join(str(tmp_path), "h" if env.get("SBD_HOST_SCAT") else "d")
        os.makedirs(d)
        with open(os.path.join(d, "INPUT"), "w") as f:
            f.write(f"\n &INPUT\n {namelist}\n /\n")
        r = subprocess.run([HOST], cwd=d, env=dict(os.environ, SBD_TIMING="1", SBD_SUMS_FILE=os.path.join(d, "sums.txt"), **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r)
        sums.append(np.loadtxt(os.path.join(d, "sums.txt")))
    assert " compact=3 " in outs[0].stderr and " compact=2 " in outs[1].stderr, outs[0].stderr[-500:]
    a = np.array([float(x) for x in outs[0].stdout.split() if _isnum(x)])
    b = np.array([float(x) for x in outs[1].stdout.split() if _isnum(x)])
    assert a.shape == b.shape and a.size > 5
    assert np.allclose(a, b, rtol=2e-4, atol=1e-12 * float(np.abs(b).max())), float(np.abs(a - b).max())   # (+ tails of absorption bands: absolute)
    assert sums[0].shape == sums[1].shape and np.allclose(sums[0], sums[1], rtol=1e-9, atol=1e-300), float(np.abs(sums[0] - sums[1]).max())


def _isnum(x):
    try:
        float(x)
        return True
    except ValueError:
        return False


def _random_scatter_namelists(seed, count):
    """Seeded picks over what sbd_scat.hpp covers (the choices of tools/fuzz_end_to_end.py for clouds and aerosols)."""
    import random
    rnd = random.Random(seed)
    pick = lambda *a: rnd.choice(a)
    out = []
    for _ in range(count):
        lo = pick(.25, .3, .4, .55, 1., 2., 3.5, 5., 8.)
        hi = min(lo * pick(1.2, 1.5, 2., 4.), 90.)
        p = ["idatm=%d" % pick(1, 2, 3, 4, 5, 6), "wlinf=%g wlsup=%g wlinc=%g" % (lo, hi, pick(.01 * lo, -.01, -.003, .02 * lo)),
             "sza=%g" % pick(0, 30, 60, 85, 95), "nstr=%d iout=10" % pick(4, 8, 16)]
        if rnd.random() < .6:
            p.append(pick("tcloud=%g zcloud=%g nre=%g" % (pick(.5, 5, 40), pick(.5, 2, 6, 11), pick(4, 8, 20, -25, -60)),
                          "lwp=%g zcloud=%g nre=%g" % (pick(20, 150), pick(1, 3), pick(6, 12, -30)),
                          "tcloud=%g,%g zcloud=%g,-%g nre=%g,%g" % (pick(3, 12), pick(1, 3, .5), pick(1, 2), pick(4, 7), pick(6, 10), pick(8, 16)),
                          "lwp=%g,%g zcloud=%g,-%g nre=%g,%g" % (pick(50, 200), pick(.3, 2), pick(1, 2), pick(3, 6), pick(5, 9), pick(12, 20)),
                          "tcloud=%g,0,%g zcloud=%g,0,%g nre=%g,8,%g" % (pick(2, 9), pick(.5, 3), pick(1, 2), pick(8, 10), pick(7, 12), pick(-20, -50))))
            if rnd.random() < .3:
                p.append("imomc=%d" % pick(1, 2, 3))
        if rnd.random() < .6:
            p.append(pick("iaer=%d vis=%g" % (pick(1, 2, 3, 4), pick(5, 23, 60)),
                          "iaer=%d tbaer=%g rhaer=%g" % (pick(1, 2, 3, 4), pick(.05, .5), pick(.3, .75, .9, .99)),
                          "iaer=5 wlbaer=.4,.7,1.5 qbaer=%g,1,.4 wbaer=.97,.9,%g gbaer=.8,.7,.55 tbaer=%g" % (pick(1.2, 2.), pick(.5, 0.), pick(.1, .8)),
                          "iaer=5 wlbaer=.55 qbaer=1 wbaer=%g gbaer=.65 tbaer=.2 abaer=%g" % (pick(.8, 1.), pick(0, .7, 1.6))))
            if rnd.random() < .3:
                p.append("nosct=%d" % pick(1, 3))
            if rnd.random() < .3:
                p.append("imoma=%d" % pick(1, 2, 3))
        if rnd.random() < .4:
            k = pick(1, 2, 3)
            p.append("jaer=%s zaer=%s taerst=%s" % (",".join(str(pick(1, 2, 3, 4)) for _ in range(k)),
                                                   ",".join("%g" % z for z in sorted(rnd.sample([12, 15, 18, 22, 26, 30], k))),
                                                   ",".join("%g" % pick(.005, .02, .1) for _ in range(k))))
            if "abaer" not in " ".join(p) and rnd.random() < .4:
                p.append("abaer=%g" % pick(.5, 1.3))
        if rnd.random() < .25:
            p.append("ngrid=%d zgrid1=%g zgrid2=%g" % (pick(20, 40, 65), pick(.5, 1, 2), pick(10, 30)))
        if rnd.random() < .2:
            p.append("xrsc=%g" % pick(0, .5, 2))
        if rnd.random() < .2:
            p.append("pbar=%g" % pick(900, 1030))
        out.append(" ".join(p))
    return out


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_scatter_source_on_the_host_bit_for_bit_on_random_runs(tmp_path, seed):
    """The same pin as above over seeded random clouds / aerosols / grids: 15 runs per seed; a run the reference's
    band model refuses, or one that does not fit the compact form, is skipped (and counted)."""
    if not os.access(HOST, os.X_OK):
        pytest.skip("Fortran host not built")
    compared = points = 0
    for i, nl in enumerate(_random_scatter_namelists(seed, 15)):
        sub = tmp_path / f"r{i}"
        sub.mkdir()
        try:
            d = dump(sub, nl)
        except (FileNotFoundError, AssertionError):
            continue                                         # (stopped by the band model's input checks, or arrays form)
        if not d["scat_ok"]:
            continue
        got = host_blocks(d)
        same = got == d["lay"]
        if not same.all():
            p, ch, l = [int(x[0]) for x in np.nonzero(~same)]
            raise AssertionError(f"{nl} :: {int((~same).sum())} of {same.size} differ; first: point {p} (wl {d['wl'][p]}) channel {ch} "
                                 f"layer {l}: {got[p, ch, l]!r} vs {d['lay'][p, ch, l]!r}")
        compared += 1
        points += len(d["wl"])
    assert compared >= 8, compared
