"""Host-side mirror of the reference's DISORT operator for SBDART's wavelength loop.

Argument names and meaning are DISORT's (disort.f:1-6, Documents/disort.doc:561-986) as
drt.f:541-546 passes them; the per-run arguments go to the constructor, the per-
(wavelength, k-term) arguments to :meth:`DisortEngine.solve`, which takes a whole batch.
Everything runs in the HIP library behind the C ABI (include/sbdart_amd.h); numpy inputs
use the host entry point, torch CUDA(=HIP) tensors the device entry point.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import BatchIn, BatchOut, RunCfg


class SbdError(RuntimeError):
    def __init__(self, code: int, where: str):
        L = _lib.load()
        self.code = code
        super().__init__(f"{where}: {L.sbd_strerror(code).decode()} [{code}] {L.sbd_last_error().decode()}")


class RetryNstr(SbdError):
    """Beam angle equals a quadrature angle (disort.f:2645-2650): use NSTR-2 / NSTR+2 as
    drt.f:536-555 does."""


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class DisortEngine:
    def __init__(self, nlyr: int, nstr: int, nmom: int, temper: Sequence[float], umu0: float,
                 phi0: float = 0.0, onlyfl: bool = True, usrang: Optional[bool] = None,
                 umu: Optional[Sequence[float]] = None, phi: Optional[Sequence[float]] = None,
                 btemp: float = 0.0, ttemp: float = 0.0, temis: float = 0.0, fisot: float = 0.0,
                 lamber: bool = True, level_out: Optional[Sequence[int]] = None, device: int = 0,
                 max_batch: int = 0, allow_retry_nstr: bool = False, corint: bool = False,
                 ibdrf: int = 0, bpar: Optional[Sequence[float]] = None, ibcnd: int = 0, pivot_exact: bool = False):
        self._L = _lib.load()
        self._h = C.c_void_p()
        self.nlyr, self.nstr, self.nmom = int(nlyr), int(nstr), int(nmom)
        self.onlyfl = bool(onlyfl)
        self._temper = _f64(temper)
        assert self._temper.shape == (self.nlyr + 1,), "TEMPER has nlyr+1 levels"
        self._umu = _f64(umu if umu is not None else [])
        self._phi = _f64(phi if phi is not None else [])
        if usrang is None:
            usrang = not self.onlyfl
        # (radiances without user angles: at the NSTR quadrature angles, CMPINT -- the engine reports NUMU = NSTR)
        quad = not self.onlyfl and not usrang
        self.numu, self.nphi = (0, 0) if self.onlyfl else ((self.nstr if quad else len(self._umu)), len(self._phi))
        cfg_numu = 0 if quad else self.numu
        self._lev = None if level_out is None else np.ascontiguousarray(level_out, dtype=np.int32)
        self._cfg = cfg = RunCfg(
            abi_version=_lib.ABI_VERSION, nlyr=self.nlyr, nstr=self.nstr, nmom=self.nmom,
            onlyfl=int(self.onlyfl), lamber=int(lamber), usrang=int(usrang), numu=cfg_numu,
            nphi=self.nphi, nlevel_out=0 if self._lev is None else len(self._lev), device=device,
            max_batch=max_batch, corint=int(bool(corint)), ibdrf=0 if lamber else int(ibdrf), umu0=umu0, phi0=phi0, fisot=fisot, btemp=btemp, ttemp=ttemp,
            temis=temis,
            temper=self._temper.ctypes.data_as(C.POINTER(C.c_double)),
            umu=self._umu.ctypes.data_as(C.POINTER(C.c_double)) if cfg_numu else None,
            phi=self._phi.ctypes.data_as(C.POINTER(C.c_double)) if self.nphi else None,
            level_out=None if self._lev is None else self._lev.ctypes.data_as(C.POINTER(C.c_int32)))
        self.ibdrf = 0 if lamber else int(ibdrf)
        self.ibcnd = int(ibcnd)
        cfg.ibcnd = self.ibcnd
        cfg.pivot_exact = int(bool(pivot_exact))          # NSTR <= 16: LINPACK's first-maximum pivot rule to the letter
        if self.ibcnd == 1:                                 # ALBTRN: results at the positive user / quadrature cosines
            cfg.numu = len(self._umu) if usrang else 0
            cfg.umu = self._umu.ctypes.data_as(C.POINTER(C.c_double)) if (usrang and len(self._umu)) else None
            self.nout = len(self._umu) if usrang else self.nstr // 2
        for k_, v_ in enumerate(list(bpar if bpar is not None else [])[:8]):
            cfg.bpar[k_] = float(v_)
        rc = self._create(cfg)
        self.retry_nstr = rc == _lib.E_RETRY_NSTR
        if rc == _lib.E_RETRY_NSTR and not allow_retry_nstr:
            self.close()
            raise RetryNstr(rc, "sbd_engine_create")
        if rc not in (_lib.OK, _lib.E_RETRY_NSTR):
            self._h = C.c_void_p()
            raise SbdError(rc, "sbd_engine_create")
        self.nlev = self._nlevel()
        self.device = device

    def _create(self, cfg) -> int:
        return self._L.sbd_engine_create(C.byref(cfg), C.byref(self._h))

    def _nlevel(self) -> int:
        return self._L.sbd_engine_nlevel(self._h)

    # ---- lifecycle ----
    def _destroy(self):
        self._L.sbd_engine_destroy(self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._destroy()
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- introspection ----
    @property
    def chunk(self) -> int:
        return self._L.sbd_engine_chunk(self._h)

    @property
    def workspace_bytes(self) -> int:
        return self._L.sbd_engine_workspace_bytes(self._h)

    @property
    def stream(self) -> int:
        return self._L.sbd_engine_stream(self._h) or 0

    def quadrature(self):
        nn = self.nstr // 2
        cmu, cwt = np.zeros(nn), np.zeros(nn)
        self._L.sbd_engine_quadrature(self._h, cmu.ctypes.data_as(C.POINTER(C.c_double)),
                                      cwt.ctypes.data_as(C.POINTER(C.c_double)))
        return cmu, cwt

    def enable_timing(self, on: bool = True):
        self._L.sbd_engine_enable_timing(self._h, int(on))

    def last_ms(self, phase: int = -1) -> float:
        return self._L.sbd_engine_last_ms(self._h, phase)

    def last_fallback_layers(self) -> int:
        return int(self._L.sbd_engine_last_fallback_layers(self._h))

    # ---- the hot path ----
    def debug_pivots(self, on: bool = True):
        """Test hook: record the band LU's pivot choices (NSTR <= 16, all output levels)."""
        rc = self._L.sbd_engine_debug_pivots(self._h, int(on))
        if rc != _lib.OK:
            raise SbdError(rc, "sbd_engine_debug_pivots")

    def debug_array(self, which: int, dtype, count: int):
        """Test hook: `count` elements of workspace array `which` (sbd_engine_debug_copy) of the last pass."""
        buf = np.zeros(count, dtype=dtype)
        got = self._L.sbd_engine_debug_copy(self._h, which, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        if got < 0:
            raise SbdError(int(got), "sbd_engine_debug_copy")
        return buf[: got // buf.itemsize]

    def solve_albtrn(self, dtauc, ssalb, pmom, albedo, wvnmlo=1.0, wvnmhi=2.0):
        """IBCND = 1 (engine created with ibcnd=1): albedo and transmissivity of the medium for beam incidence at the
        output cosines.  Returns (albtrn[W, 2, nout], status[W])."""
        assert self.ibcnd == 1
        dtauc, ssalb, pmom = _f64(dtauc), _f64(ssalb), _f64(pmom)
        W = dtauc.shape[0]
        assert pmom.shape == (W, self.nlyr, self.nmom + 1)
        lo, hi, al = (_f64(np.broadcast_to(x, (W,))) for x in (wvnmlo, wvnmhi, albedo))
        albtrn = np.zeros((W, 2, self.nout))
        status = np.zeros(W, dtype=np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        bi = BatchIn(W, vp(dtauc), vp(ssalb), vp(pmom), vp(lo), vp(hi), None, vp(al), None, None)
        bo = BatchOut(None, None, vp(status), vp(albtrn))
        rc = self._L.sbd_engine_solve_host(self._h, C.byref(bi), C.byref(bo))
        if rc != _lib.OK:
            raise SbdError(rc, "sbd_engine_solve_host")
        return albtrn, status

    def pass_count(self, nwork: int) -> int:
        """Equal passes a resident batch of nwork items goes through (alternating between two workspaces / streams)."""
        return int(self._L.sbd_engine_pass_count(self._h, int(nwork)))

    def solve(self, dtauc, ssalb, pmom, wvnmlo, wvnmhi, fbeam, albedo, plank, bitem=None, pmom_row=None):
        """Solve a batch.  Shapes: dtauc/ssalb [W, nlyr]; pmom [W, nlyr, nmom+1] -- or, with pmom_row [W] (int32 block
        index per item), [npmom, nlyr, nmom+1]: the k-terms of a spectral point share their moments;
        wvnmlo/wvnmhi/fbeam/albedo [W]; plank [W] bool; bitem [W, 4] with the ocean surface (ibdrf = 1) only.
        Returns (flux[W,5,nlev], uu[W,nphi,nlev,numu] or None, status[W])."""
        try:
            import torch
            is_t = isinstance(dtauc, torch.Tensor)
        except Exception:  # torch is plumbing only
            is_t = False
        if is_t:
            return self._solve_device(dtauc, ssalb, pmom, wvnmlo, wvnmhi, fbeam, albedo, plank, bitem=bitem, pmom_row=pmom_row)
        dtauc, ssalb, pmom = _f64(dtauc), _f64(ssalb), _f64(pmom)
        W = dtauc.shape[0]
        assert dtauc.shape == (W, self.nlyr) and ssalb.shape == (W, self.nlyr)
        rows = None if pmom_row is None else np.ascontiguousarray(pmom_row, dtype=np.int32)
        assert pmom.shape == ((W if rows is None else pmom.shape[0]), self.nlyr, self.nmom + 1)
        lo, hi, fb, al = (_f64(np.broadcast_to(x, (W,))) for x in (wvnmlo, wvnmhi, fbeam, albedo))
        pl = np.ascontiguousarray(np.broadcast_to(plank, (W,)), dtype=np.uint8)
        flux = np.zeros((W, _lib.NFLUX, self.nlev))
        uu = None if self.onlyfl else np.zeros((W, self.nphi, self.nlev, self.numu))
        status = np.zeros(W, dtype=np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        bt = None if bitem is None else _f64(bitem).reshape(W, 4)
        bi = BatchIn(W, vp(dtauc), vp(ssalb), vp(pmom), vp(lo), vp(hi), vp(fb), vp(al), vp(pl), None if bt is None else vp(bt),
                     None if rows is None else vp(rows), 0 if rows is None else pmom.shape[0])
        bo = BatchOut(vp(flux), None if uu is None else vp(uu), vp(status))
        rc = self._L.sbd_engine_solve_host(self._h, C.byref(bi), C.byref(bo))
        if rc != _lib.OK:
            raise SbdError(rc, "sbd_engine_solve_host")
        return flux, uu, status

    def _solve_device(self, dtauc, ssalb, pmom, wvnmlo, wvnmhi, fbeam, albedo, plank,
                      out=None, stream: Optional[int] = None, bitem=None, pmom_row=None):
        import torch
        W = dtauc.shape[0]
        for t in (dtauc, ssalb, pmom, wvnmlo, wvnmhi, fbeam, albedo):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        assert plank.is_cuda and plank.dtype == torch.uint8 and plank.is_contiguous()
        dev = dtauc.device
        assert dev.index == self.device, "tensors live on cuda:%s, the engine on device %d" % (dev.index, self.device)
        assert all(t.device == dev for t in (ssalb, pmom, wvnmlo, wvnmhi, fbeam, albedo, plank))
        assert dtauc.shape == (W, self.nlyr) and ssalb.shape == (W, self.nlyr)
        assert pmom.shape == ((W if pmom_row is None else pmom.shape[0]), self.nlyr, self.nmom + 1)
        assert all(t.shape == (W,) for t in (wvnmlo, wvnmhi, fbeam, albedo, plank))
        if out is None:
            flux = torch.empty((W, _lib.NFLUX, self.nlev), dtype=torch.float64, device=dev)
            uu = None if self.onlyfl else torch.empty((W, self.nphi, self.nlev, self.numu),
                                                       dtype=torch.float64, device=dev)
            status = torch.empty(W, dtype=torch.int32, device=dev)
        else:
            flux, uu, status = out
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        bi = BatchIn(W, dtauc.data_ptr(), ssalb.data_ptr(), pmom.data_ptr(), wvnmlo.data_ptr(),
                     wvnmhi.data_ptr(), fbeam.data_ptr(), albedo.data_ptr(), plank.data_ptr(),
                     None if bitem is None else bitem.data_ptr(),
                     None if pmom_row is None else pmom_row.data_ptr(), 0 if pmom_row is None else pmom.shape[0])
        bo = BatchOut(flux.data_ptr(), 0 if uu is None else uu.data_ptr(), status.data_ptr())
        rc = self._L.sbd_engine_solve_device(self._h, C.byref(bi), C.byref(bo), C.c_void_p(stream))
        if rc != _lib.OK:
            raise SbdError(rc, "sbd_engine_solve_device")
        return flux, uu, status

    solve_device = _solve_device

    def accumulate(self, weight, flux, uu=None, acc_flux=None, acc_uu=None):
        """stdout1's weighted sums (drt.f:964-1054) on the GPU; numpy in/out."""
        weight, flux = _f64(weight), _f64(flux)
        W = len(weight)
        if acc_flux is None:
            acc_flux = np.zeros((_lib.NFLUX, self.nlev))
        if uu is not None and acc_uu is None:
            acc_uu = np.zeros((self.nphi, self.nlev, self.numu))
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        if uu is not None:
            uu = _f64(uu)
        rc = self._L.sbd_engine_accumulate_host(self._h, W, vp(weight), vp(flux), vp(uu),
                                                vp(acc_flux), vp(acc_uu))
        if rc != _lib.OK:
            raise SbdError(rc, "sbd_engine_accumulate_host")
        return acc_flux, acc_uu


class DisortFleet(DisortEngine):
    """One engine per GPU of the node behind one handle (sbd_fleet_*, include/sbdart_amd.h): a
    batch is cut into contiguous shards, every device solves its shard, and stdout1's weighted
    sums come back reduced (RCCL over xGMI between distinct devices, host-side otherwise).
    `devices=None` takes every visible device; a device may be listed twice (test set-ups)."""

    def __init__(self, *args, devices: Optional[Sequence[int]] = None, **kw):
        self._devices = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        kw.pop("device", None)
        super().__init__(*args, **kw)

    def _create(self, cfg) -> int:
        d = self._devices
        return self._L.sbd_fleet_create(C.byref(cfg), 0 if d is None else len(d),
                                        None if d is None else d.ctypes.data_as(C.POINTER(C.c_int32)),
                                        C.byref(self._h))

    def _nlevel(self) -> int:
        return self._L.sbd_engine_nlevel(self._L.sbd_fleet_engine(self._h, 0))

    def _destroy(self):
        self._L.sbd_fleet_destroy(self._h)

    @property
    def size(self) -> int:
        return self._L.sbd_fleet_size(self._h)

    @property
    def uses_rccl(self) -> bool:
        return bool(self._L.sbd_fleet_uses_rccl(self._h))

    def last_enqueue(self):
        """Per device (begin, end) of its enqueue in the last solve, host seconds since that call began, and the
        number of input arrays the call page-locked."""
        t0, t1, npin = C.c_double(), C.c_double(), C.c_int32()
        out = []
        for i in range(self.size):
            rc = self._L.sbd_fleet_last_enqueue(self._h, i, C.byref(t0), C.byref(t1), C.byref(npin))
            if rc != _lib.OK:
                raise SbdError(rc, "sbd_fleet_last_enqueue")
            out.append((t0.value, t1.value))
        return out, npin.value

    def shard_range(self, nwork: int, rank: int):
        lo, hi = C.c_int32(), C.c_int32()
        self._L.sbd_shard_range(nwork, self.size, rank, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def solve(self, dtauc, ssalb, pmom, wvnmlo, wvnmhi, fbeam, albedo, plank, weight=None, items=True, bitem=None,
              pmom_row=None):
        """Host (numpy) batch through every device.  Returns (flux, uu, status) and, when `weight`
        is given, also (acc_flux[5,nlev], acc_uu or None) = sum_i weight[i] * outputs[i]."""
        dtauc, ssalb, pmom = _f64(dtauc), _f64(ssalb), _f64(pmom)
        W = dtauc.shape[0]
        assert dtauc.shape == (W, self.nlyr) and ssalb.shape == (W, self.nlyr)
        rows = None if pmom_row is None else np.ascontiguousarray(pmom_row, dtype=np.int32)
        assert pmom.shape == ((W if rows is None else pmom.shape[0]), self.nlyr, self.nmom + 1)
        lo, hi, fb, al = (_f64(np.broadcast_to(x, (W,))) for x in (wvnmlo, wvnmhi, fbeam, albedo))
        pl = np.ascontiguousarray(np.broadcast_to(plank, (W,)), dtype=np.uint8)
        flux = np.zeros((W, _lib.NFLUX, self.nlev)) if items else None
        uu = None if (self.onlyfl or not items) else np.zeros((W, self.nphi, self.nlev, self.numu))
        status = np.zeros(W, dtype=np.int32)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        bt = None if bitem is None else _f64(bitem).reshape(W, 4)
        bi = BatchIn(W, vp(dtauc), vp(ssalb), vp(pmom), vp(lo), vp(hi), vp(fb), vp(al), vp(pl), vp(bt), vp(rows),
                     0 if rows is None else pmom.shape[0])
        bo = BatchOut(vp(flux), vp(uu), vp(status))
        acc_f = acc_u = None
        if weight is not None:
            weight = _f64(np.broadcast_to(weight, (W,)))
            acc_f = np.zeros((_lib.NFLUX, self.nlev))
            acc_u = None if self.onlyfl else np.zeros((self.nphi, self.nlev, self.numu))
        rc = self._L.sbd_fleet_solve_host(self._h, C.byref(bi), C.byref(bo), vp(weight), vp(acc_f), vp(acc_u))
        if rc != _lib.OK:
            raise SbdError(rc, "sbd_fleet_solve_host")
        if weight is None:
            return flux, uu, status
        return flux, uu, status, acc_f, acc_u

    def gas_terms(self, gas_model, wl, lay, want_depths=False):
        """The gas part of the band model on the fleet's devices (sbd_fleet_gas_terms): gas_model a _lib.GasModel, wl
        [npoint], lay [npoint][channels][nlyr] the points' layer blocks.  Returns (nk [npoint], wt [npoint][3], fail
        [npoint], depths [npoint][3][nlyr] or None); the depths stay on the devices for solve_mix(dtaug=None, kterm=...).
        The layer blocks stay there too: `self.lay_token` is the generation number the library returned for them, and
        solve_mix(..., lay_token=self.lay_token) reads the device copy instead of sending `lay` again.  Residency is that
        explicit statement only -- a solve_mix without the token always stages the `lay` it is given (the
        library used to recognise the caller's array by its address, which a temporary array can share with a freed one)."""
        wl, lay = _f64(wl), _f64(lay)
        npt = wl.shape[0]
        assert lay.shape[0] == npt and lay.shape[2] == self.nlyr
        nk = np.zeros(npt, dtype=np.int32)
        wt = np.zeros((npt, 3))
        fail = np.zeros(npt, dtype=np.int32)
        depths = np.zeros((npt, 3, self.nlyr)) if want_depths else None
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        token = C.c_int64(0)
        self.lay_token = 0
        rc = self._L.sbd_fleet_gas_terms(self._h, C.byref(gas_model), npt, vp(wl), vp(lay), lay.shape[1], vp(nk), vp(wt),
                                         vp(fail), vp(depths), C.byref(token))
        if rc != _lib.OK:
            raise SbdError(rc, "sbd_fleet_gas_terms")
        self.lay_token = int(token.value)
        return nk, wt, fail, depths

    def point_terms(self, gas_model, scat_model, wl, nch, want_depths=False, want_blocks=False):
        """gas_terms with the layer blocks MADE on the devices from the scatterers' model (sbd_fleet_point_terms; scat_model a
        _lib.ScatModel, nch = 4 + 3 x its terms).  Returns (nk, wt, fail, depths or None, blocks [npoint][nch][nlyr] or None);
        `self.lay_token` names the blocks for solve_mix(lay=None, lay_token=...)."""
        wl = _f64(wl)
        npt = wl.shape[0]
        nk = np.zeros(npt, dtype=np.int32)
        wt = np.zeros((npt, 3))
        fail = np.zeros(npt, dtype=np.int32)
        depths = np.zeros((npt, 3, self.nlyr)) if want_depths else None
        blocks = np.zeros((npt, int(nch), self.nlyr)) if want_blocks else None
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        token = C.c_int64(0)
        self.lay_token = 0
        rc = self._L.sbd_fleet_point_terms(self._h, C.byref(gas_model), C.byref(scat_model), npt, vp(wl), int(nch), vp(nk), vp(wt),
                                           vp(fail), vp(depths), vp(blocks), C.byref(token))
        if rc != _lib.OK:
            raise SbdError(rc, "sbd_fleet_point_terms")
        self.lay_token = int(token.value)
        return nk, wt, fail, depths, blocks

    def solve_mix(self, point_of, dtaug, lay, family, wvnmlo, wvnmhi, fbeam, albedo, plank, weight=None, items=True, kterm=None,
                  lay_token=0):
        """A batch in COMPACT form (sbd_mix_in, include/sbdart_amd.h): per spectral point a block lay[point] of
        [4 + 3 nterm][nlyr] doubles (dtauc, dtaua, dtaur, tsc, then g, m1, m2 of every scattering term; `family` =
        GETMOM's iphas per term), per work item the gas of its k-term; DTAUC / SSALB / PMOM are formed on the device
        (what depthscl, GETMOM, taucloud / tauaero and normom do on the host: taugas.f:7598-7603, disutil.f:2104-2209,
        drt.f:1390-1395).  A fleet of several devices cuts the batch between spectral points.  Same returns as `solve`."""
        from ._lib import MIX_MAX_TERMS, MixIn
        rows = np.ascontiguousarray(point_of, dtype=np.int32)
        W = rows.shape[0]
        dtaug = None if dtaug is None else _f64(dtaug)          # (None: the depths gas_terms left on the devices, by kterm)
        kt = None if kterm is None else np.ascontiguousarray(kterm, dtype=np.int32)
        family = [int(x) for x in family]
        nterm = len(family)
        if lay is None:                                         # (the blocks point_terms made on the devices: lay_token)
            assert lay_token != 0
            NP = int(np.shape(wvnmlo)[0])
        else:
            lay = _f64(lay)
            NP = lay.shape[0]
            assert lay.shape == (NP, 4 + 3 * nterm, self.nlyr)
        assert (dtaug is None or dtaug.shape == (W, self.nlyr)) and nterm <= MIX_MAX_TERMS
        lo, hi, fb, al = (_f64(np.broadcast_to(x, (NP,))) for x in (wvnmlo, wvnmhi, fbeam, albedo))
        pl = np.ascontiguousarray(np.broadcast_to(plank, (NP,)), dtype=np.uint8)
        flux = np.zeros((W, _lib.NFLUX, self.nlev)) if items else None
        uu = None if (self.onlyfl or not items) else np.zeros((W, self.nphi, self.nlev, self.numu))
        status = np.zeros(W, dtype=np.int32)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        fam = (C.c_int32 * MIX_MAX_TERMS)(*(family + [0] * (MIX_MAX_TERMS - nterm)))
        mi = MixIn(W, NP, vp(rows), vp(dtaug), nterm, fam, vp(lay), vp(lo), vp(hi), vp(fb), vp(al), vp(pl), vp(kt), int(lay_token))
        bo = BatchOut(vp(flux), vp(uu), vp(status))
        acc_f = acc_u = None
        if weight is not None:
            weight = _f64(np.broadcast_to(weight, (W,)))
            acc_f = np.zeros((_lib.NFLUX, self.nlev))
            acc_u = None if self.onlyfl else np.zeros((self.nphi, self.nlev, self.numu))
        rc = self._L.sbd_fleet_solve_mix_host(self._h, C.byref(mi), C.byref(bo), vp(weight), vp(acc_f), vp(acc_u))
        if rc != _lib.OK:
            raise SbdError(rc, "sbd_fleet_solve_mix_host")
        if weight is None:
            return flux, uu, status
        return flux, uu, status, acc_f, acc_u

    _solve_device = solve_device = None      # device-pointer entry points belong to single engines
    accumulate = None
    chunk = workspace_bytes = stream = None
    quadrature = enable_timing = last_ms = None


def engine_for_record(rec, level_out=None, device=0, max_batch=0, allow_retry_nstr=True, pivot_exact=False):
    """Engine whose per-run arguments are those of one SolveRecord."""
    return DisortEngine(
        nlyr=rec.nlyr, nstr=rec.nstr, nmom=rec.nmom, temper=rec.temper, umu0=rec.umu0,
        phi0=rec.phi0, onlyfl=rec.onlyfl, usrang=rec.usrang, umu=rec.umu, phi=rec.phi,
        btemp=rec.btemp, ttemp=rec.ttemp, temis=rec.temis, fisot=rec.fisot, lamber=rec.lamber,
        level_out=level_out, device=device, max_batch=max_batch, allow_retry_nstr=allow_retry_nstr,
        corint=getattr(rec, "corint", False), ibdrf=getattr(rec, "ibdrf", 0), bpar=getattr(rec, "bpar", None),
        pivot_exact=pivot_exact)


def run_key(rec):
    return (rec.nlyr, rec.nstr, rec.nmom, rec.flags & ~1, rec.umu0, rec.phi0, rec.btemp, rec.ttemp,
            rec.temis, rec.fisot, rec.temper.tobytes(), rec.umu.tobytes(), rec.phi.tobytes(),
            getattr(rec, "ibdrf", 0), np.asarray(getattr(rec, "bpar", np.zeros(8))).tobytes(),
            int(getattr(rec, "ibcnd", 0)))


def solve_records(recs, level_out=None, device=0):
    """Solve SolveRecords on the GPU, grouping records that share per-run arguments.
    Returns lists (flux[5,nlev], uu or None, status) in input order."""
    groups = {}
    for i, r in enumerate(recs):
        groups.setdefault(run_key(r), []).append(i)
    flux_out, uu_out, st_out = [None] * len(recs), [None] * len(recs), [0] * len(recs)
    for idx in groups.values():
        r0 = recs[idx[0]]
        if getattr(r0, "ibcnd", 0) == 1:
            # the reference returns zeros in every flux / intensity argument and ALBMED / TRNMED beside them (disort.f:545-556):
            # another result shape -- DisortEngine(ibcnd=1).solve_albtrn is the entry point, not this one
            raise ValueError("solve_records: IBCND = 1 records are solved with DisortEngine(ibcnd=1).solve_albtrn")
        with engine_for_record(r0, level_out=level_out, device=device, max_batch=len(idx)) as eng:
            flux, uu, st = eng.solve(
                np.stack([recs[i].dtauc for i in idx]), np.stack([recs[i].ssalb for i in idx]),
                np.stack([recs[i].pmom for i in idx]), [recs[i].wvnmlo for i in idx],
                [recs[i].wvnmhi for i in idx], [recs[i].fbeam for i in idx],
                [recs[i].albedo for i in idx], [recs[i].plank for i in idx],
                bitem=np.stack([recs[i].bitem for i in idx]) if getattr(r0, "ibdrf", 0) == 1 else None)
        for k, i in enumerate(idx):
            flux_out[i], st_out[i] = flux[k], int(st[k])
            if uu is not None:
                uu_out[i] = uu[k]
    return flux_out, uu_out, st_out
