"""The compact form of a batch (sbd_mix_in, include/sbdart_amd.h): DTAUC / SSALB / PMOM formed ON
THE DEVICE from what scatters at a spectral point and the gas of the item's k-term.

CPU: the restatement (oracle/mix_restatement.py) is pinned against the compiled reference's GETMOM (integer powers,
the Rayleigh 0.1) and against the Fortran host's band model on a run with a Henyey-Greenstein cloud.
GPU: the engine fed with the compact form returns bit for bit the fluxes it returns when fed with the restatement's
arrays -- the device-side assembly IS the restatement."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import REF_DIR, ROOT
from mix_restatement import RAY2, assemble, powi_fortran

REFLIB = os.path.join(REF_DIR, "libsbdart_ref.so")
HOST = os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")


@pytest.mark.skipif(not os.path.exists(REFLIB), reason="oracle/_ref not built")
def test_integer_powers_and_rayleigh_moment_are_the_references():
    """GETMOM (disutil.f:2104-2209) of the reference as compiled here: iphas 3 gives GG**K -- bit-equal to square-and-
    multiply, NOT to pow() for every g; iphas 2 gives the REAL*4 0.1."""
    L = C.CDLL(REFLIB)
    nmom = 40
    rng = np.random.default_rng(5)
    differs_from_pow = 0
    for g in list(rng.uniform(0.0, 0.95, 200)) + [0.0, 0.5, 0.85, 0.9]:
        pm = np.zeros(nmom + 1)
        L.getmom_(C.byref(C.c_int(3)), C.byref(C.c_double(g)), C.byref(C.c_int(nmom)), pm.ctypes.data_as(C.c_void_p))
        mine = np.array([float(powi_fortran(g, k)) for k in range(nmom + 1)])
        assert np.array_equal(pm, mine), (g, np.nonzero(pm != mine)[0][:5])
        differs_from_pow += int(np.any(pm != np.power(g, np.arange(nmom + 1))))
    assert differs_from_pow > 0            # (the reason the device does not call pow)
    pm = np.zeros(nmom + 1)
    L.getmom_(C.byref(C.c_int(2)), C.byref(C.c_double(0.0)), C.byref(C.c_int(nmom)), pm.ctypes.data_as(C.c_void_p))
    assert pm[2] == RAY2 and pm[2] != 0.1 and pm[0] == 1.0 and np.count_nonzero(pm) == 2


def read_mix_dump(path):
    """SBD_DUMP_MIX's file: (lay [P][channels][L], dtaug [W][L], point_of [W], family list) or None for the arrays form."""
    b = open(path, "rb").read()
    h = np.frombuffer(b[:44], dtype=np.int32)
    nz, nch, npt, nterm, nrec = int(h[0]), int(h[1]), int(h[2]), int(h[3]), int(h[10])
    if nz == 0:
        return None
    o = 44
    lay = np.frombuffer(b[o:o + 8 * nz * nch * npt]).reshape(npt, nch, nz)
    o += 8 * nz * nch * npt
    dtaug = np.frombuffer(b[o:o + 8 * nz * nrec]).reshape(nrec, nz)
    o += 8 * nz * nrec
    po = np.frombuffer(b[o:o + 4 * nrec], dtype=np.int32)
    return lay, dtaug, po, [int(x) for x in h[4:4 + nterm]]


COMPACT_RUNS = [
    # clear sky; a Henyey-Greenstein cloud; cloud + rural aerosol + a stratospheric layer; two aerosol layers aloft,
    # isotropic cloud; thermal with a low stratus (BASELINE configs[2]); extended cloud; no-scattering aerosols; usrcld
    ("idatm=4 wlinf=.3 wlsup=1.0 wlinc=.1 nstr=4 iout=10", 0),
    ("idatm=4 wlinf=.5 wlsup=.8 wlinc=.05 tcloud=8 zcloud=2 nre=10 imomc=3 nstr=8 iout=10 kdist=1", 1),
    ("idatm=4 wlinf=.5 wlsup=.8 wlinc=.05 tcloud=8 zcloud=2 nre=10 nstr=8 iout=10 iaer=1 vis=20 jaer=1 zaer=20 taerst=.02", 3),
    ("idatm=2 wlinf=.4 wlsup=2.4 wlinc=.4 tcloud=3 zcloud=4 imomc=1 nstr=16 iout=1 iaer=3 tbaer=.3 jaer=2,4 zaer=15,22 taerst=.01,.03 imoma=2", 4),
    ("idatm=6 wlinf=4 wlsup=80 wlinc=-.2 nstr=16 tcloud=10 zcloud=1 nre=8 iout=10 sza=95", 1),
    ("idatm=4 wlinf=.6 wlsup=.7 wlinc=.05 tcloud=6,2 zcloud=1,-4 nre=8,20 nstr=4 iout=10", 1),
    ("idatm=4 wlinf=.6 wlsup=.7 wlinc=.05 nstr=4 iout=10 iaer=2 vis=5 nosct=1", 1),
]


@pytest.mark.parametrize("namelist,nterm", COMPACT_RUNS)
def test_compact_form_of_the_band_model_reproduces_its_arrays(tmp_path, namelist, nterm):
    """The Fortran host's band model in its two output forms on the same INPUT: DISORT's arguments as arrays
    (SBD_DUMP_OPTICS; bit-equal to the live reference's: tests/test_band_model.py) and the compact form the host hands to
    sbd_fleet_solve_mix_host (SBD_DUMP_MIX).  The restatement of the device's assembly (oracle/mix_restatement.py) turns
    the second into the first BIT FOR BIT: DTAUC's four-term sum, SSALB, and every moment of every scattering term in the
    reference's association -- clouds, boundary-layer and stratospheric aerosols, isotropic / Rayleigh / Henyey-Greenstein."""
    from sbdart_amd.records import read_records
    if not os.access(HOST, os.X_OK):
        pytest.skip("Fortran host not built")
    d = str(tmp_path)
    with open(os.path.join(d, "INPUT"), "w") as f:
        f.write(f"\n &INPUT\n {namelist}\n /\n")
    out, mixf = os.path.join(d, "items.sbdrec"), os.path.join(d, "mix.bin")
    env = dict(os.environ, SBD_OPTICS=os.path.join(d, "none"))
    subprocess.run([HOST], cwd=d, env=dict(env, SBD_DUMP_OPTICS=out), capture_output=True, text=True)
    subprocess.run([HOST], cwd=d, env=dict(env, SBD_DUMP_MIX=mixf), capture_output=True, text=True)
    recs = read_records(out)
    mix = read_mix_dump(mixf)
    assert mix is not None and len(recs) > 0
    lay, dtaug, po, fam = mix
    assert len(fam) == nterm and len(po) == len(recs) and lay.shape[1] == 4 + 3 * nterm
    dt, ss, pm = assemble(po, dtaug, lay, fam, recs[0].nmom)
    for i, r in enumerate(recs):
        assert po[i] == r.iwl - 1
        assert np.array_equal(dt[i], r.dtauc), (i, np.abs(dt[i] - r.dtauc).max())
        assert np.array_equal(ss[i], r.ssalb), i
        assert np.array_equal(pm[po[i]], r.pmom), (i, np.abs(pm[po[i]] - r.pmom).max())
    if nterm:
        assert np.count_nonzero(lay[:, 5::3]) > 0                  # (a scattering term really carries a factor)


@pytest.mark.parametrize("namelist,why", [
    ("idatm=4 wlinf=.6 wlsup=.7 wlinc=.05 tcloud=6,2 zcloud=1,1 nre=8,20 nstr=4 iout=10", "two clouds in one layer"),
    ("idatm=4 wlinf=.6 wlsup=.7 wlinc=.05 tcloud=6 zcloud=1 imomc=5 nstr=4 iout=10", "tabulated cloud phase function"),
    ("idatm=4 wlinf=.6 wlsup=.7 wlinc=.05 nstr=8 iout=20 nzen=2 uzen=100,170 nphi=2 phi=0,90 corint=t", "intensity corrections"),
    ("idatm=4 wlinf=.6 wlsup=.7 wlinc=.05 nstr=8 iout=10 isalb=7 sc=5,.5,34", "ocean surface"),
])
def test_runs_that_keep_the_arrays_form_say_why(tmp_path, namelist, why):
    if not os.access(HOST, os.X_OK):
        pytest.skip("Fortran host not built")
    d = str(tmp_path)
    with open(os.path.join(d, "INPUT"), "w") as f:
        f.write(f"\n &INPUT\n {namelist}\n /\n")
    r = subprocess.run([HOST], cwd=d, env=dict(os.environ, SBD_OPTICS=os.path.join(d, "none"), SBD_DUMP_MIX=os.path.join(d, "mix.bin")),
                       capture_output=True, text=True)
    assert read_mix_dump(os.path.join(d, "mix.bin")) is None
    assert why in r.stderr, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("nstr,nwl", [(16, 700), (4, 300), (32, 40)])
def test_device_assembly_is_the_restatement(nstr, nwl):
    """sbd_fleet_solve_mix_host on the compact sweep == sbd_fleet_solve_host on the restatement's arrays, bit for bit
    (fluxes, status words, weighted sums) -- several passes (SBD_CHUNK in a subprocess is the other tests' business:
    here the batch is large enough for more than one pass at NSTR 16)."""
    from sbdart_amd.engine import DisortFleet
    from sbdart_amd.workload import sw_sweep_mix
    m = sw_sweep_mix(nwl=nwl, nstr=nstr, seed=77)
    dtauc, ssalb, pmom = assemble(m.point_of, m.dtaug, m.lay, m.family, m.nmom)
    d2, s2, p2 = m.arrays()                                # (the product's own numpy statement, used by bench.py)
    assert np.array_equal(d2, dtauc) and np.array_equal(s2, ssalb) and np.array_equal(p2, pmom)
    with DisortFleet(nlyr=m.nlyr, nstr=m.nstr, nmom=m.nmom, temper=m.temper, umu0=m.umu0, btemp=m.btemp, ttemp=m.ttemp,
                     temis=m.temis, onlyfl=True, level_out=[0, m.nlyr], devices=[0]) as fl:
        a = fl.solve(dtauc, ssalb, pmom, m.wvnmlo[m.point_of], m.wvnmhi[m.point_of], m.fbeam[m.point_of],
                     m.albedo[m.point_of], m.plank[m.point_of], weight=m.weight, pmom_row=m.point_of)
        b = fl.solve_mix(*m.mix_args(), weight=m.weight)
    assert np.array_equal(a[2], b[2]) and (a[2] == 0).all()
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[3], b[3])
    assert np.isfinite(a[0]).all() and np.abs(a[0]).max() > 0


@pytest.mark.gpu
def test_compact_batch_argument_errors():
    from sbdart_amd.engine import DisortFleet, SbdError
    from sbdart_amd.workload import sw_sweep_mix
    m = sw_sweep_mix(nwl=8, nstr=8, seed=3)
    with DisortFleet(nlyr=m.nlyr, nstr=m.nstr, nmom=m.nmom, temper=m.temper, umu0=m.umu0, btemp=m.btemp, ttemp=m.ttemp,
                     temis=m.temis, onlyfl=True, level_out=[0, m.nlyr], devices=[0]) as fl:
        bad = m.point_of.copy()
        bad[0], bad[-1] = bad[-1], bad[0]                          # not non-decreasing
        with pytest.raises(SbdError):
            fl.solve_mix(bad, *m.mix_args()[1:])
        with pytest.raises(SbdError):                              # a tabulated phase-function family: arrays form only
            fl.solve_mix(m.point_of, m.dtaug, m.lay, (3, 5), m.wvnmlo, m.wvnmhi, m.fbeam, m.albedo, m.plank)
