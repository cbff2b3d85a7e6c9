// Launch interface between the host side of the C ABI (sbd_engine.hip) and the kernel
// translation units (sbd_k_*.hip).  One TU per kernel family so that the library builds in
// parallel; every function dispatches its compile-time shape (NSTR/2, lanes per layer, ...)
// from run-time values and is a no-op for shapes it has no instance of.
#pragma once
#include <hip/hip_runtime.h>
#include "sbd_common.hpp"

namespace sbd {

// (NN, G) table of the fast layer kernel: G lanes per layer (a power of two up to NSTR 32: the exchanges ride on the DPP
// network; 20 for NSTR 34-40, whose groups talk through ds_bpermute anyway -- three layers per wave instead of two)
#define SBD_L2_CASES(M)                                                                        \
    M(2, 4) M(3, 4) M(4, 4) M(5, 8) M(6, 8) M(7, 8) M(8, 8) M(9, 16) M(10, 16) M(11, 16)   \
    M(12, 16) M(13, 16) M(14, 16) M(15, 16) M(16, 16) M(17, 20) M(18, 20) M(19, 20) M(20, 20)
inline int l2_group(int nn)
{
#define SBD_L2_G(NNv, Gv) if (nn == NNv) return Gv;
    SBD_L2_CASES(SBD_L2_G)
#undef SBD_L2_G
    return 64;
}

// kernels with more than 48 KB of dynamic LDS need the attribute raised once
inline hipError_t raise_lds(const void *fn, int bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

void launch_setup(unsigned grid, hipStream_t st, const Params &P);
hipError_t prepare_layer_v1(int G, int lds);
void launch_layer_v1(int G, unsigned grid, int lds, hipStream_t st, const Params &P, int32_t *only_flagged);
hipError_t prepare_layer2(int nn, bool rad, int lds);
void launch_layer2(int nn, bool rad, unsigned grid, int lds, hipStream_t st, const Params &P, int32_t *eigflag);
hipError_t prepare_band_lds(int nn, int lds);
void launch_band_lds(int nn, unsigned grid, int lds, hipStream_t st, const Params &P);
bool has_band_rows(int nn);
int band_rows_lds_bytes(int nn);
void launch_band_rows(int nn, unsigned grid, hipStream_t st, const Params &P, bool fused);
void launch_band4(int nn, unsigned grid, hipStream_t st, const Params &P, bool fused);
void launch_band4_pivdbg(int nn, unsigned grid, hipStream_t st, const Params &P);
void launch_band4_exact(int nn, unsigned grid, hipStream_t st, const Params &P, bool fused, bool pivdbg);
void launch_band1(int nn, unsigned grid, hipStream_t st, const Params &P, bool fused);
hipError_t prepare_backsolve(int nn, int lds);
void launch_backsolve(int nn, unsigned grid, int lds, hipStream_t st, const Params &P);
void launch_backsolve4(int nn, unsigned grid, hipStream_t st, const Params &P);
void launch_backsolve1(int nn, unsigned grid, hipStream_t st, const Params &P);
void launch_usrint(unsigned grid, int lds, hipStream_t st, const Params &P);
void launch_cmpint(unsigned grid, hipStream_t st, const Params &P);
void launch_azimuth(unsigned grid, hipStream_t st, const Params &P, int naz_run);
void launch_intcor(unsigned grid, hipStream_t st, const Params &P, int naz_run);
// errmsg 2 on LINPACK's own estimate (sbd_refband.hpp, sbd_k_refband.hip): doubles of scratch per block; the kernel that
// serves the pass's list of flagged systems (Params::rclist)
size_t band_rcond_scratch_doubles(int n, int L);
void launch_band_rcond(unsigned grid, hipStream_t st, const Params &P, double *scratch, size_t stride, double *rcond_dbg);

}  // namespace sbd
