"""TEST INFRASTRUCTURE -- ctypes binding of the C oracle (oracle/disort_oracle.c).

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (sbdart_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None
_LIB_FMA: Optional[C.CDLL] = None

WARN_SOLVE0_RCOND, WARN_UPBEAM_RCOND, WARN_UPISOT_RCOND = 1, 2, 4
ERR_ASYMTX, RETRY_NSTR, ERR_INPUT, WARN_PLKAVG, WARN_PLKCONV = 8, 16, 32, 64, 128

_dp = C.POINTER(C.c_double)


class _In(C.Structure):
    _fields_ = [(k, C.c_int) for k in
                ("nlyr", "nstr", "nmom", "numu", "nphi", "plank", "onlyfl", "lamber",
                 "usrang", "usrtau", "ntau")] + \
               [(k, C.c_double) for k in
                ("wvnmlo", "wvnmhi", "fbeam", "umu0", "phi0", "fisot", "albedo", "btemp",
                 "ttemp", "temis", "accur")] + \
               [("corint", C.c_int)] + \
               [(k, _dp) for k in ("dtauc", "ssalb", "temper", "pmom", "umu", "phi", "utau")] + \
               [("ibdrf", C.c_int), ("bpar", C.c_double * 8), ("bitem", C.c_double * 4), ("ibcnd", C.c_int)]


class _Out(C.Structure):
    _fields_ = [("nstr_out", C.c_int), ("status", C.c_int), ("ntau", C.c_int)] + \
               [(k, _dp) for k in ("rfldir", "rfldn", "flup", "dfdt", "uavg", "uu", "u0c")] + \
               [("dbg_mode", C.c_int)] + \
               [(k, _dp) for k in ("dbg_gc", "dbg_kk", "dbg_ll", "dbg_zz", "dbg_zplk0", "dbg_zplk1")] + \
               [("albmed", _dp), ("trnmed", _dp), ("dbg_ipvt", C.POINTER(C.c_int))]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "disort_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.sbdo_disort.argtypes = [C.POINTER(_In), C.POINTER(_Out)]
        L.sbdo_disort.restype = C.c_int
        L.sbdo_qgausn.argtypes = [C.c_int, _dp, _dp]
        L.sbdo_lepoly.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.sbdo_plkavg.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int)]
        L.sbdo_plkavg.restype = C.c_double
        L.sbdo_asymtx.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp]
        L.sbdo_asymtx.restype = C.c_int
        L.sbdo_sgbfa.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                 C.POINTER(C.c_int)]
        L.sbdo_sgbsl.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _dp]
        L.sbdo_pi.restype = C.c_double
        L.sbdo_dither.restype = C.c_double
        _LIB = L
    return _LIB


def lib_fma() -> C.CDLL:
    """The same restatement compiled with fused multiply-adds: a rounding-perturbed twin (see oracle/Makefile)."""
    global _LIB_FMA
    if _LIB_FMA is None:
        so = os.path.join(_HERE, "liboracle_fma.so")
        src = os.path.join(_HERE, "disort_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_fma.so"])
        L = C.CDLL(so)
        L.sbdo_disort.argtypes = [C.POINTER(_In), C.POINTER(_Out)]
        L.sbdo_disort.restype = C.c_int
        _LIB_FMA = L
    return _LIB_FMA


def _p(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def disort(rec, utau=None, accur: float = 0.0, want_u0c: bool = False, debug_mode=None, perturbed: bool = False):
    """Solve one record (sbdart_amd.records.SolveRecord-like). Returns a dict.  perturbed: through the
    FMA-contracted twin of the library (rounding sensitivity of the record, never a reference answer)."""
    L = lib_fma() if perturbed else lib()
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    dtauc, ssalb, temper, pmom = f(rec.dtauc), f(rec.ssalb), f(rec.temper), f(rec.pmom)
    umu = f(rec.umu) if len(rec.umu) else np.zeros(1)
    phi = f(rec.phi) if len(rec.phi) else np.zeros(1)
    usrtau = utau is not None
    ut = f(utau) if usrtau else np.zeros(1)
    ntau = len(ut) if usrtau else rec.nlyr + 1
    numu, nphi = len(rec.umu), len(rec.phi)
    nout = numu if (rec.usrang or rec.onlyfl) else rec.nstr     # USRANG off: intensities at the NSTR quadrature angles
    i = _In(nlyr=rec.nlyr, nstr=rec.nstr, nmom=rec.nmom, numu=numu, nphi=nphi,
            plank=int(rec.plank), onlyfl=int(rec.onlyfl), lamber=int(rec.lamber),
            usrang=int(rec.usrang), usrtau=int(usrtau), ntau=ntau,
            wvnmlo=rec.wvnmlo, wvnmhi=rec.wvnmhi, fbeam=rec.fbeam, umu0=rec.umu0,
            phi0=rec.phi0, fisot=rec.fisot, albedo=rec.albedo, btemp=rec.btemp,
            ttemp=rec.ttemp, temis=rec.temis, accur=accur, corint=int(getattr(rec, 'corint', False)),
            dtauc=_p(dtauc), ssalb=_p(ssalb), temper=_p(temper), pmom=_p(pmom),
            umu=_p(umu), phi=_p(phi), utau=_p(ut), ibdrf=int(getattr(rec, "ibdrf", 0)),
            ibcnd=int(getattr(rec, "ibcnd", 0)))
    for k_ in range(8):
        i.bpar[k_] = float(getattr(rec, "bpar", np.zeros(8))[k_])
    for k_ in range(4):
        i.bitem[k_] = float(getattr(rec, "bitem", np.zeros(4))[k_])
    flx = np.zeros((5, ntau))
    uu = np.zeros((max(nphi, 1), ntau, max(nout, 1)))
    u0c = np.zeros((ntau, rec.nstr))
    o = _Out(rfldir=_p(flx[0]), rfldn=_p(flx[1]), flup=_p(flx[2]), dfdt=_p(flx[3]),
             uavg=_p(flx[4]), uu=_p(uu), u0c=_p(u0c) if want_u0c else None)
    albtrn = None
    if getattr(rec, "ibcnd", 0) == 1:
        albtrn = np.zeros((2, max(numu if rec.usrang else rec.nstr // 2, 1)))
        o.albmed, o.trnmed = _p(albtrn[0]), _p(albtrn[1])
    dbg = None
    if debug_mode is not None:
        n_, L_ = rec.nstr, rec.nlyr
        dbg = dict(gc=np.zeros((L_, n_, n_)), kk=np.zeros((L_, n_)), ll=np.zeros((L_, n_)),
                   zz=np.zeros((L_, n_)), zplk0=np.zeros((L_, n_)), zplk1=np.zeros((L_, n_)))
        o.dbg_mode = int(debug_mode)
        for k_, v_ in dbg.items():
            setattr(o, "dbg_" + k_, _p(v_))
        ipvt = np.zeros(n_ * L_, dtype=np.int32)
        o.dbg_ipvt = ipvt.ctypes.data_as(C.POINTER(C.c_int))
        dbg["ipvt"] = ipvt
    st = L.sbdo_disort(C.byref(i), C.byref(o))
    res = dict(status=st, nstr_out=o.nstr_out, rfldir=flx[0], rfldn=flx[1], flup=flx[2],
               dfdt=flx[3], uavg=flx[4])
    if albtrn is not None:
        res["albmed"], res["trnmed"] = albtrn[0], albtrn[1]
    if not rec.onlyfl:
        res["uu"] = uu[:nphi, :, :nout]
    if want_u0c:
        res["u0c"] = u0c
    if dbg is not None:
        dbg["gc"] = dbg["gc"].transpose(0, 2, 1).copy()   # -> [lc][i][j]
        res["dbg"] = dbg
    return res
