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


def test_restatement_against_the_band_model(tmp_path):
    """A run whose only particles are a Henyey-Greenstein cloud (imomc = 3) over Rayleigh scattering: the work items
    the Fortran host's band model makes (bit-equal to the reference's: tests/test_band_model.py) are recovered from
    their compact form -- moments and single-scattering albedo exactly, DTAUC to the last bit of a three-term sum."""
    from sbdart_amd.records import read_records
    if not os.access(HOST, os.X_OK):
        pytest.skip("Fortran host not built")
    d = str(tmp_path)
    with open(os.path.join(d, "INPUT"), "w") as f:
        f.write("\n &INPUT\n idatm=4 wlinf=.5 wlsup=.8 wlinc=.05 tcloud=8 zcloud=2 nre=10 imomc=3 nstr=8 iout=10 kdist=1\n /\n")
    out = os.path.join(d, "items.sbdrec")
    subprocess.run([HOST], cwd=d, env=dict(os.environ, SBD_DUMP_OPTICS=out, SBD_OPTICS=os.path.join(d, "none")),
                   capture_output=True, text=True)
    recs = read_records(out)
    assert len(recs) >= 7
    r = recs[0]
    L, nmom = r.nlyr, r.nmom
    # compact form of item 0, read back from its arguments: scattering depth, its split by the second moment's excess
    dt, w, pm = r.dtauc, r.ssalb, r.pmom                       # pm [L][nmom+1]
    scat = w * dt
    cloud = np.argmax(pm[:, 1])                                  # the cloud's layer: the only one with a first moment
    assert pm[cloud, 1] > 0.3 and np.count_nonzero(pm[:, 1]) == 1
    # in the cloud layer: PMOM(1) = s_hg g / scat, PMOM(3) = s_hg g^3 / scat -> g, then s_hg
    g = np.sqrt(pm[cloud, 3] / pm[cloud, 1])
    s_hg = pm[cloud, 1] * scat[cloud] / g
    tsc_hg = np.zeros(L); tsc_hg[cloud] = s_hg
    g_hg = np.zeros(L); g_hg[cloud] = g
    tsc_ray = scat - tsc_hg
    _, w2, pm2 = assemble(np.zeros(1, dtype=np.int32), dt[None, :] * 0.5, dt[None, :] * 0.5, tsc_hg[None, :], g_hg[None, :],
                          tsc_ray[None, :], nmom)
    assert np.allclose(w2[0], w, rtol=4e-16, atol=0)
    assert np.allclose(pm2[0], pm, rtol=0, atol=3e-15)           # (g and s_hg were recovered from rounded quotients)
    assert pm2[0][cloud, 2] > pm2[0][cloud, 3]                   # Rayleigh's 0.1 sits in the second moment


@pytest.mark.gpu
@pytest.mark.parametrize("nstr,nwl", [(16, 700), (4, 300), (32, 40)])
def test_device_assembly_is_the_restatement(nstr, nwl):
    """sbd_fleet_solve_mix_host on the compact sweep == sbd_fleet_solve_host on the restatement's arrays, bit for bit
    (fluxes, status words, weighted sums) -- several passes (SBD_CHUNK in a subprocess is the other tests' business:
    here the batch is large enough for more than one pass at NSTR 16)."""
    from sbdart_amd.engine import DisortFleet
    from sbdart_amd.workload import sw_sweep_mix
    m = sw_sweep_mix(nwl=nwl, nstr=nstr, seed=77)
    dtauc, ssalb, pmom = assemble(m.point_of, m.dtaug, m.dtaux, m.tsc_hg, m.g_hg, m.tsc_ray, m.nmom)
    with DisortFleet(nlyr=m.nlyr, nstr=m.nstr, nmom=m.nmom, temper=m.temper, umu0=m.umu0, btemp=m.btemp, ttemp=m.ttemp,
                     temis=m.temis, onlyfl=True, level_out=[0, m.nlyr], devices=[0]) as fl:
        a = fl.solve(dtauc, ssalb, pmom, m.wvnmlo[m.point_of], m.wvnmhi[m.point_of], m.fbeam[m.point_of],
                     m.albedo[m.point_of], m.plank[m.point_of], weight=m.weight, pmom_row=m.point_of)
        b = fl.solve_mix(m.point_of, m.dtaug, m.dtaux, m.tsc_hg, m.g_hg, m.tsc_ray, m.wvnmlo, m.wvnmhi, m.fbeam,
                         m.albedo, m.plank, weight=m.weight)
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
            fl.solve_mix(bad, m.dtaug, m.dtaux, m.tsc_hg, m.g_hg, m.tsc_ray, m.wvnmlo, m.wvnmhi, m.fbeam, m.albedo, m.plank)
