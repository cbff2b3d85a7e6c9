"""The "SBDREC1" stream format: one DISORT solve (one (wavelength, k-term) work
item of drt.f's wl_loop/kd_loop, drt.f:425-561) per record.

This is the single definition of the format; the Fortran readers/writers in
``oracle/ref/*.f90`` and ``sbdart_amd/fortran/*.f90`` follow it.

All little-endian, no padding::

    file header : char magic[8] = "SBDREC1\\0"; int32 nrec (-1 = until EOF);
                  int32 has_out
    per record  : int32 hdr[12] = nlyr, nstr, nmom, numu, nphi, flags, kd, nk,
                                  iwl, ibcnd, 0, 0
                  flags: bit0 PLANK, bit1 ONLYFL, bit2 LAMBER, bit3 USRANG, bit4 CORINT
                  float64 sc[16] = wl, wt, ff, wvnmlo, wvnmhi, fbeam, umu0,
                                   phi0, albedo, btemp, ttemp, temis, fisot,
                                   accur, 0, 0
                  float64 dtauc[nlyr], ssalb[nlyr], temper[nlyr+1],
                          pmom[nlyr][nmom+1], umu[numu], phi[nphi]
      if ibdrf != 0 (LAMBER off: a bidirectional surface, spectra.f:249-296; 1 ocean, 2 Hapke, 3 Ross-Li):
                  float64 bpar[8]  = the model's run parameters (albblk, spectra.f:15-26):
                                     1: wndspd, foam cover, foam reflectance, chlor, salin
                                     2: hssa, hasym, hotspt, hotwdth     3: rliso, rlvol, rlgeo, rlhot, rlwdth
                  float64 bitem[4] = per-item constants of the ocean model: refractive index nr, ni of the
                                     water and its sub-surface reflectance rsw at this wavelength
      if has_out: int32 ohdr[4] = nstr_out (<0: "retry with another NSTR",
                                  disort.f:2645-2650), ntau, numu, 0
                  float64 rfldir[ntau], rfldn[ntau], flup[ntau], dfdt[ntau],
                          uavg[ntau]
                  if not ONLYFL: float64 uu[nphi][ntau][numu]
                  if ibcnd == 1: float64 albmed[numu_out], trnmed[numu_out]  (ALBTRN, disort.f:6718-7000; ohdr[2] = numu_out)

The argument names are DISORT's own (disort.f:1-6, Documents/disort.doc:561-986);
``wl, wt, ff, kd, nk, iwl`` are the driver-side quantities stdout1 needs
(drt.f:909-1091): wavelength, k-distribution weight, filter value, k index,
number of k terms, 1-based wavelength index.
"""
from __future__ import annotations

import dataclasses
import struct
from typing import BinaryIO, Iterable, List, Optional

import numpy as np

MAGIC = b"SBDREC1\0"

F_PLANK, F_ONLYFL, F_LAMBER, F_USRANG, F_CORINT = 1, 2, 4, 8, 16


@dataclasses.dataclass
class SolveRecord:
    nlyr: int
    nstr: int
    nmom: int
    flags: int
    wvnmlo: float
    wvnmhi: float
    fbeam: float
    umu0: float
    phi0: float
    albedo: float
    btemp: float
    ttemp: float
    temis: float
    dtauc: np.ndarray           # [nlyr]
    ssalb: np.ndarray           # [nlyr]
    temper: np.ndarray          # [nlyr+1]
    pmom: np.ndarray            # [nlyr, nmom+1]
    umu: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0))
    phi: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0))
    fisot: float = 0.0
    accur: float = 0.0
    wl: float = 0.0
    wt: float = 1.0
    ff: float = 1.0
    kd: int = 1
    nk: int = 1
    iwl: int = 0
    ibcnd: int = 0
    ib: int = 1                          # KDIST = -1: sub-band ib (counting down from nb) of a k-distribution file's point
    nb: int = 1
    ibdrf: int = 0                       # 0 Lambertian, 1 ocean, 2 Hapke, 3 Ross-Li (LAMBER flag off)
    bpar: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(8))
    bitem: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(4))
    # outputs (None when the record carries inputs only)
    nstr_out: Optional[int] = None
    rfldir: Optional[np.ndarray] = None
    rfldn: Optional[np.ndarray] = None
    flup: Optional[np.ndarray] = None
    dfdt: Optional[np.ndarray] = None
    uavg: Optional[np.ndarray] = None
    uu: Optional[np.ndarray] = None   # [nphi, ntau, numu]
    albmed: Optional[np.ndarray] = None   # IBCND = 1: albedo / transmissivity of the medium at the output angles
    trnmed: Optional[np.ndarray] = None

    @property
    def plank(self) -> bool:
        return bool(self.flags & F_PLANK)

    @property
    def onlyfl(self) -> bool:
        return bool(self.flags & F_ONLYFL)

    @property
    def lamber(self) -> bool:
        return bool(self.flags & F_LAMBER)

    @property
    def usrang(self) -> bool:
        return bool(self.flags & F_USRANG)

    @property
    def corint(self) -> bool:
        return bool(self.flags & F_CORINT)

    @property
    def numu(self) -> int:
        return int(len(self.umu))

    @property
    def nphi(self) -> int:
        return int(len(self.phi))

    def has_out(self) -> bool:
        return self.rfldir is not None

    def inputs_only(self) -> "SolveRecord":
        return dataclasses.replace(self, nstr_out=None, rfldir=None, rfldn=None,
                                   flup=None, dfdt=None, uavg=None, uu=None, albmed=None, trnmed=None)


def _rd(f: BinaryIO, dtype, n: int) -> np.ndarray:
    nbytes = np.dtype(dtype).itemsize * n
    b = f.read(nbytes)
    if len(b) != nbytes:
        raise EOFError
    return np.frombuffer(b, dtype=dtype, count=n).copy()


def read_records(path: str) -> List[SolveRecord]:
    out: List[SolveRecord] = []
    with open(path, "rb") as f:
        magic = f.read(8)
        if magic != MAGIC:
            raise ValueError(f"{path}: not an SBDREC1 file")
        nrec, has_out = struct.unpack("<ii", f.read(8))
        while nrec < 0 or len(out) < nrec:
            try:
                hdr = _rd(f, "<i4", 12)
            except EOFError:
                break
            sc = _rd(f, "<f8", 16)
            nlyr, nstr, nmom, numu, nphi, flags = (int(x) for x in hdr[:6])
            dtauc = _rd(f, "<f8", nlyr)
            ssalb = _rd(f, "<f8", nlyr)
            temper = _rd(f, "<f8", nlyr + 1)
            pmom = _rd(f, "<f8", nlyr * (nmom + 1)).reshape(nlyr, nmom + 1)
            umu = _rd(f, "<f8", numu)
            phi = _rd(f, "<f8", nphi)
            ibdrf = int(hdr[10])
            bpar = _rd(f, "<f8", 8) if ibdrf else np.zeros(8)
            bitem = _rd(f, "<f8", 4) if ibdrf else np.zeros(4)
            r = SolveRecord(
                nlyr=nlyr, nstr=nstr, nmom=nmom, flags=flags,
                wvnmlo=sc[3], wvnmhi=sc[4], fbeam=sc[5], umu0=sc[6], phi0=sc[7],
                albedo=sc[8], btemp=sc[9], ttemp=sc[10], temis=sc[11],
                fisot=sc[12], accur=sc[13], wl=sc[0], wt=sc[1], ff=sc[2],
                kd=int(hdr[6]), nk=int(hdr[7]), iwl=int(hdr[8]), ibcnd=int(hdr[9]),
                ibdrf=ibdrf, bpar=bpar, bitem=bitem,
                ib=(int(hdr[11]) & 65535) or 1, nb=(int(hdr[11]) >> 16) or 1,
                dtauc=dtauc, ssalb=ssalb, temper=temper, pmom=pmom, umu=umu, phi=phi)
            if has_out:
                ohdr = _rd(f, "<i4", 4)
                ntau = int(ohdr[1])
                flx = _rd(f, "<f8", 5 * ntau).reshape(5, ntau)
                r.nstr_out = int(ohdr[0])
                r.rfldir, r.rfldn, r.flup, r.dfdt, r.uavg = (flx[i] for i in range(5))
                if not r.onlyfl:
                    nout = int(ohdr[2])                    # (USRANG off: the NSTR quadrature angles, disort.f:2655-2669)
                    r.uu = _rd(f, "<f8", nphi * ntau * nout).reshape(nphi, ntau, nout)
                if r.ibcnd == 1:
                    r.albmed, r.trnmed = _rd(f, "<f8", int(ohdr[2])), _rd(f, "<f8", int(ohdr[2]))
            out.append(r)
    return out


def write_records(path: str, recs: Iterable[SolveRecord], with_out: Optional[bool] = None) -> None:
    recs = list(recs)
    if with_out is None:
        with_out = bool(recs) and all(r.has_out() for r in recs)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<ii", len(recs), 1 if with_out else 0))
        for r in recs:
            hdr = np.zeros(12, "<i4")
            hdr[:11] = [r.nlyr, r.nstr, r.nmom, r.numu, r.nphi, r.flags, r.kd, r.nk,
                        r.iwl, r.ibcnd, r.ibdrf]
            hdr[11] = (r.ib + 65536 * r.nb) if r.nb > 1 else 0
            sc = np.zeros(16, "<f8")
            sc[:14] = [r.wl, r.wt, r.ff, r.wvnmlo, r.wvnmhi, r.fbeam, r.umu0, r.phi0,
                       r.albedo, r.btemp, r.ttemp, r.temis, r.fisot, r.accur]
            f.write(hdr.tobytes())
            f.write(sc.tobytes())
            for a, shape in ((r.dtauc, (r.nlyr,)), (r.ssalb, (r.nlyr,)),
                             (r.temper, (r.nlyr + 1,)), (r.pmom, (r.nlyr, r.nmom + 1)),
                             (r.umu, (r.numu,)), (r.phi, (r.nphi,))):
                a = np.ascontiguousarray(a, dtype="<f8")
                assert a.shape == shape, (a.shape, shape)
                f.write(a.tobytes())
            if r.ibdrf:
                f.write(np.ascontiguousarray(r.bpar, dtype="<f8").tobytes())
                f.write(np.ascontiguousarray(r.bitem, dtype="<f8").tobytes())
            if with_out:
                ntau = len(r.rfldir)
                nout = r.numu if (r.onlyfl or r.uu is None) else np.asarray(r.uu).shape[2]
                if r.ibcnd == 1 and r.albmed is not None:
                    nout = len(r.albmed)
                f.write(np.array([r.nstr_out, ntau, nout, 0], "<i4").tobytes())
                for a in (r.rfldir, r.rfldn, r.flup, r.dfdt, r.uavg):
                    f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
                if not r.onlyfl:
                    a = np.ascontiguousarray(r.uu, dtype="<f8")
                    assert a.shape == (r.nphi, ntau, nout)
                    f.write(a.tobytes())
                if r.ibcnd == 1:
                    f.write(np.ascontiguousarray(r.albmed, dtype="<f8").tobytes())
                    f.write(np.ascontiguousarray(r.trnmed, dtype="<f8").tobytes())
