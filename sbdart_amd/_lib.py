"""ctypes view of the C ABI in include/sbdart_amd.h.  Fails loudly when the HIP
library is missing: there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SBDART_AMD_LIB points at another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("SBDART_AMD_LIB") or os.path.join(_HERE, "lib", "libsbdart_amd.so")

ABI_VERSION = 7
NFLUX = 5
RFLDIR, RFLDN, FLUP, DFDT, UAVG = range(5)

OK, E_INVALID, E_RETRY_NSTR, E_NO_DEVICE, E_HIP, E_UNSUPPORTED, E_NOMEM = 0, -1, -2, -3, -4, -5, -6
ST_WARN_SOLVE0, ST_WARN_UPBEAM, ST_WARN_UPISOT, ST_ERR_EIGEN = 0x01, 0x02, 0x04, 0x08
ST_RETRY_NSTR, ST_ERR_INPUT, ST_WARN_PLKAVG, ST_WARN_PLKCONV = 0x10, 0x20, 0x40, 0x80

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


class RunCfg(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("abi_version", "nlyr", "nstr", "nmom", "onlyfl", "lamber", "usrang", "numu",
                 "nphi", "nlevel_out", "device", "max_batch", "corint", "ibdrf")] + \
               [(k, C.c_double) for k in ("umu0", "phi0", "fisot", "btemp", "ttemp", "temis")] + \
               [("temper", _dp), ("umu", _dp), ("phi", _dp), ("level_out", _ip), ("bpar", C.c_double * 8),
                ("ibcnd", C.c_int32), ("pivot_exact", C.c_int32)]


class BatchIn(C.Structure):
    _fields_ = [("nwork", C.c_int32), ("dtauc", C.c_void_p), ("ssalb", C.c_void_p),
                ("pmom", C.c_void_p), ("wvnmlo", C.c_void_p), ("wvnmhi", C.c_void_p),
                ("fbeam", C.c_void_p), ("albedo", C.c_void_p), ("plank", C.c_void_p), ("bitem", C.c_void_p),
                ("pmom_row", C.c_void_p), ("npmom", C.c_int32)]


class BatchOut(C.Structure):
    _fields_ = [("flux", C.c_void_p), ("uu", C.c_void_p), ("status", C.c_void_p), ("albtrn", C.c_void_p)]


MIX_MAX_TERMS = 6


class MixIn(C.Structure):
    """sbd_mix_in (ABI v6; lay_token v7): a batch in compact form -- per spectral point a block [4 + 3 nterm][nlyr] (dtauc, dtaua, dtaur,
    tsc, then g, m1, m2 of every scattering term), per work item the gas of its k-term."""
    _fields_ = [("nwork", C.c_int32), ("npoint", C.c_int32), ("point_of", C.c_void_p), ("dtaug", C.c_void_p),
                ("nterm", C.c_int32), ("family", C.c_int32 * MIX_MAX_TERMS), ("lay", C.c_void_p),
                ("wvnmlo", C.c_void_p), ("wvnmhi", C.c_void_p), ("fbeam", C.c_void_p), ("albedo", C.c_void_p),
                ("plank", C.c_void_p), ("kterm", C.c_void_p), ("lay_token", C.c_int64)]


class GasModel(C.Structure):
    """sbd_gas_model: the gas part of the band model for a run."""
    _fields_ = [("nz", C.c_int32), ("kdist", C.c_int32), ("uu", C.c_void_p), ("z", C.c_void_p),
                ("amu0_first", C.c_double), ("amu0_rest", C.c_double), ("xo4", C.c_double),
                ("tables", C.c_void_p), ("tables_bytes", C.c_size_t)]


class ScatModel(C.Structure):
    """sbd_scat_model: the scatterers' part of the band model for a run (Rayleigh, cloud deck, aerosols)."""
    _fields_ = [("nz", C.c_int32), ("z", C.c_void_p), ("p", C.c_void_p), ("t", C.c_void_p), ("xrsc", C.c_double),
                ("cloud_term", C.c_int32), ("cld_nslot", C.c_int32), ("cld_layer", C.c_int32 * 5),
                ("cld_tcloud", C.c_double * 5), ("cld_lwp", C.c_double * 5), ("cld_nre", C.c_double * 5),
                ("iaer", C.c_int32), ("nosct", C.c_int32), ("aer_nwl", C.c_int32),
                ("aer_wl", C.c_void_p), ("aer_ext", C.c_void_p), ("aer_absb", C.c_void_p), ("aer_asym", C.c_void_p),
                ("abaer", C.c_double), ("aer_column", C.c_void_p),
                ("nstrat", C.c_int32), ("jaer", C.c_int32 * 5), ("strat_layer", C.c_int32 * 5), ("taerst", C.c_double * 5),
                ("tables", C.c_void_p), ("tables_bytes", C.c_size_t)]


GAS_SLOTS = 63
TABLES_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "sbdart_tables.bin")


EXPORTS = (
    "sbd_engine_create", "sbd_engine_destroy", "sbd_engine_solve_device", "sbd_engine_solve_host",
    "sbd_engine_accumulate_device", "sbd_engine_accumulate_host", "sbd_abi_version",
    "sbd_engine_nlevel", "sbd_engine_workspace_bytes", "sbd_engine_chunk", "sbd_engine_stream",
    "sbd_engine_quadrature", "sbd_engine_last_ms", "sbd_engine_enable_timing", "sbd_engine_last_fallback_layers",
    "sbd_strerror",
    "sbd_last_error", "sbd_engine_debug_copy", "sbd_engine_debug_pivots",
    "sbd_fleet_create", "sbd_fleet_destroy", "sbd_fleet_size", "sbd_fleet_engine", "sbd_fleet_uses_rccl",
    "sbd_shard_range", "sbd_fleet_solve_host", "sbd_fleet_last_enqueue", "sbd_host_alloc", "sbd_host_free",
    "sbd_surface_flux_albedo", "sbd_fleet_solve_mix_host", "sbd_engine_pass_count", "sbd_shard_range_points", "sbd_band_rcond_host", "sbd_fleet_gas_terms", "sbd_gas_terms_host", "sbd_fleet_point_terms", "sbd_scatter_blocks_host",
)

_LIB = None


def load() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"sbdart_amd: HIP library not built ({LIB_PATH}); run `python -c 'import "
            f"__graft_entry__ as g; g.build()'` or `make -C sbdart_amd/csrc`. There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.sbd_engine_create.argtypes = [C.POINTER(RunCfg), C.POINTER(vp)]
    L.sbd_engine_create.restype = C.c_int
    L.sbd_engine_destroy.argtypes = [vp]
    L.sbd_engine_destroy.restype = None
    L.sbd_engine_solve_device.argtypes = [vp, C.POINTER(BatchIn), C.POINTER(BatchOut), vp]
    L.sbd_engine_solve_device.restype = C.c_int
    L.sbd_engine_solve_host.argtypes = [vp, C.POINTER(BatchIn), C.POINTER(BatchOut)]
    L.sbd_engine_solve_host.restype = C.c_int
    L.sbd_engine_accumulate_device.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.sbd_engine_accumulate_device.restype = C.c_int
    L.sbd_engine_accumulate_host.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
    L.sbd_engine_accumulate_host.restype = C.c_int
    L.sbd_abi_version.restype = C.c_int32
    L.sbd_engine_nlevel.argtypes = [vp]
    L.sbd_engine_nlevel.restype = C.c_int32
    L.sbd_engine_workspace_bytes.argtypes = [vp]
    L.sbd_engine_workspace_bytes.restype = C.c_size_t
    L.sbd_engine_chunk.argtypes = [vp]
    L.sbd_engine_chunk.restype = C.c_int32
    L.sbd_engine_pass_count.argtypes = [vp, C.c_int32]
    L.sbd_engine_pass_count.restype = C.c_int32
    L.sbd_engine_stream.argtypes = [vp]
    L.sbd_engine_stream.restype = vp
    L.sbd_engine_quadrature.argtypes = [vp, _dp, _dp]
    L.sbd_engine_quadrature.restype = C.c_int
    L.sbd_engine_last_ms.argtypes = [vp, C.c_int]
    L.sbd_engine_last_ms.restype = C.c_double
    L.sbd_engine_enable_timing.argtypes = [vp, C.c_int]
    L.sbd_engine_enable_timing.restype = None
    L.sbd_engine_last_fallback_layers.argtypes = [vp]
    L.sbd_engine_last_fallback_layers.restype = C.c_int64
    L.sbd_strerror.argtypes = [C.c_int]
    L.sbd_strerror.restype = C.c_char_p
    L.sbd_last_error.restype = C.c_char_p
    L.sbd_engine_debug_copy.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.sbd_engine_debug_copy.restype = C.c_longlong
    L.sbd_engine_debug_pivots.argtypes = [vp, C.c_int]
    L.sbd_engine_debug_pivots.restype = C.c_int
    L.sbd_fleet_create.argtypes = [C.POINTER(RunCfg), C.c_int32, _ip, C.POINTER(vp)]
    L.sbd_fleet_create.restype = C.c_int
    L.sbd_fleet_destroy.argtypes = [vp]
    L.sbd_fleet_destroy.restype = None
    L.sbd_fleet_size.argtypes = [vp]
    L.sbd_fleet_size.restype = C.c_int32
    L.sbd_fleet_engine.argtypes = [vp, C.c_int32]
    L.sbd_fleet_engine.restype = vp
    L.sbd_fleet_uses_rccl.argtypes = [vp]
    L.sbd_fleet_uses_rccl.restype = C.c_int32
    L.sbd_shard_range.argtypes = [C.c_int32, C.c_int32, C.c_int32, _ip, _ip]
    L.sbd_shard_range.restype = None
    L.sbd_fleet_solve_host.argtypes = [vp, C.POINTER(BatchIn), C.POINTER(BatchOut), vp, vp, vp]
    L.sbd_fleet_solve_host.restype = C.c_int
    L.sbd_fleet_solve_mix_host.argtypes = [vp, C.POINTER(MixIn), C.POINTER(BatchOut), vp, vp, vp]
    L.sbd_fleet_solve_mix_host.restype = C.c_int
    L.sbd_shard_range_points.argtypes = [C.c_int32, vp, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.sbd_shard_range_points.restype = None
    L.sbd_fleet_gas_terms.argtypes = [vp, C.POINTER(GasModel), C.c_int32, vp, vp, C.c_int32, vp, vp, vp, vp, C.POINTER(C.c_int64)]
    L.sbd_fleet_gas_terms.restype = C.c_int
    L.sbd_gas_terms_host.argtypes = [C.POINTER(GasModel), C.c_int32, C.c_int32, vp, vp, C.c_int32, vp, vp, vp, vp]
    L.sbd_gas_terms_host.restype = C.c_int
    L.sbd_fleet_point_terms.argtypes = [vp, C.POINTER(GasModel), C.POINTER(ScatModel), C.c_int32, vp, C.c_int32, vp, vp, vp, vp, vp,
                                        C.POINTER(C.c_int64)]
    L.sbd_fleet_point_terms.restype = C.c_int
    L.sbd_scatter_blocks_host.argtypes = [C.POINTER(ScatModel), C.c_int32, vp, C.c_int32, vp]
    L.sbd_scatter_blocks_host.restype = C.c_int
    L.sbd_band_rcond_host.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, vp, vp, vp, C.POINTER(C.c_double)]
    L.sbd_band_rcond_host.restype = C.c_int
    L.sbd_fleet_last_enqueue.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.sbd_fleet_last_enqueue.restype = C.c_int
    L.sbd_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.sbd_host_alloc.restype = C.c_int
    L.sbd_host_free.argtypes = [vp]
    L.sbd_host_free.restype = None
    _LIB = L
    return L
