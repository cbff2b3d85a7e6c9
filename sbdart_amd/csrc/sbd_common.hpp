// sbdart_amd -- shared device/host definitions for the batched DISORT engine (gfx950).
//
// Data layout in HBM (all fp64 unless noted; "slot" = work item inside the current
// chunk, "ms" = slot * nmode + azimuth mode):
//
//   inputs   dtauc/ssalb [nwork][L], pmom [nwork][L][nmom+1], wvnmlo/hi, fbeam, albedo [nwork]
//   sv       [slot][SV(L)]   per-solve vectors from the setup kernel (delta-M scaled
//                            optical depths, beam transmission, Planck terms, ...)
//   svi      [slot][SVI(L)]  int32: ncut, lyrcut, status, layru[L+1]
//   gc       [ms][L][n][n]   eigenvector blocks GC(iq,jq,lc), ROW-major in (iq,jq) so that
//                            the band assembly / flux evaluation read contiguous rows
//   kk       [ms][L][n]      eigenvalues, -k first (disort.f:3264-3269 ordering)
//   ek       [ms][L][nn]     exp(KK(iq)*dtau') iq<=nn  (STWJ scaling factors, disort.f:2846)
//   zz, zp0, zp1, ll [ms][L][n]   particular solutions and integration constants
//   ga, gb   [ms][L][n][n]   interface blocks of the boundary-value matrix, ready to load:
//                            ga(jq,iq) = +GC(jq,iq,lc)*[EK(n+1-iq) if iq>nn] (row jq of interface lc),
//                            gb(jq,iq) = -GC(jq,iq,lc)*[EK(iq) if iq<=nn]    (row jq of interface lc-1)
//   ufac     [ms][N][CW]     U factor of the band LU, row-major (row k holds U(k, k..k+2NCD))
//   gu       [ms][L][n][numu], zb/z0u/z1u [ms][L][numu]   user-angle interpolants (radiance)
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SBD_DEVICE __device__ __forceinline__

namespace sbd {

constexpr int SBD_NFLUX_ = 5;  // flux components per level (SBD_NFLUX of the C ABI)
constexpr int kMaxNlyr = 65;   // params.f: mxly
constexpr int kMaxNstr = 40;   // params.f: nstrms

// fp32-rounded literals of the reference (every un-suffixed Fortran literal is REAL*4)
#define SBD_F32(x) ((double)(x##f))

struct Tables {            // per-run constants, device pointers
    const double *cmu;     // [n]   +mu (1..nn) then -mu
    const double *cwt;     // [n]
    const double *ylmc;    // [nmode][n][n+1]   YLMC(l, iq), mirrored for iq>nn
    const double *ylm0;    // [nmode][n+1]      YLM0(l) at -umu0
    const double *ylmu;    // [nmode][numu][n+1]
    const double *cosmphi; // [nmode][nphi]     cos(m*(phi-phi0)*rpd), row 0 = 1
    const double *cosphi;  // [nphi]            cos((phi-phi0)*rpd), always filled (INTCOR's scattering angle)
    const double *zeros;   // [n][n] of 0.0 (the x_lc+1 block of the bottom-boundary rows, sbd_band4.hpp)
    const double *tags;    // [nn][nn]: rows 0..2 hold 1, 2, 3 -- marks of the fused band kernel's functional rows
    const double *gmu50;   // [50] SURFAC's azimuth / incidence quadrature: QGAUSN(25) and its mirror image (disort.f:3707-3716)
    const double *gwt50;   // [50]
    const double *temper;  // [L+1]
    const double *umu;     // [numu]
    const int32_t *level_out; // [nlev]
};

struct Params {
    int32_t L, n, nn, nmom, numu, nphi, nlev, nmode;
    int32_t onlyfl, usrang, all_levels;
    int32_t force_fallback;  // test hook: route every layer through the QR kernel
    int32_t *eighint;        // host-visible word: the list's length as the list-walking layer kernel last found it (this workspace)
    int32_t *eiglist;        // [1 + nslot*nmode*L] count + (item, mode, layer) indices left to the QR kernel
    // errmsg 2 on LINPACK's own estimate (sbd_refband.hpp): the systems (item x mode) of this pass that a cheap filter flagged
    // -- an item with a layer within 1e-12 of conservative scattering, SSALB = 1 included (setup_kernel marks its systems:
    // rcflag = 2), or a pivot ratio <= 1e-10 in the band LU -- are listed by the band kernels in rclist (count, then ms
    // indices; rcflag[ms] = 1: listed) for band_rcond_kernel at the end of the pass
    int32_t *rclist, *rcflag;
    int32_t *rchint;         // host-visible word: the list's length as band_rcond_kernel last found it (this workspace)
    int32_t nslot;          // work items in this chunk
    int32_t sv_stride, svi_stride;
    int32_t cw, ncd;        // band: CW = 2*ncd+1 columns kept per U row, ncd = 3nn-1
    int32_t ublock;         // U rows stored by layer block (sbd_band4.hpp) instead of diagonal-relative
    int32_t gconly;         // the band kernel scales GC itself (sbd_band4.hpp): the layer kernels write no ga/gb
    double umu0, rumu0, fisot, btemp, ttemp, temis;      // (rumu0 = 1 / umu0, formed once on the host)
    double pi, dither;
    Tables t;
    // chunk inputs (device pointers, already offset to the chunk)
    const double *dtauc, *ssalb, *pmom, *wvnmlo, *wvnmhi, *fbeam, *albedo;
    const uint8_t *plank;
    const int32_t *pmom_row;   // NULL: pmom holds one block of moments per work item; else the block of item `slot` is
                               // row pmom_row[slot] of pmom (moments per SPECTRAL POINT, shared by its k-terms)
    // workspace
    double *sv; int32_t *svi;
    double *gc, *kk, *ek, *zz, *zp0, *zp1, *ll, *ufac;
    double *yv;             // [ms][L*n] right-hand side / forward-eliminated RHS of the band system
    double *ga, *gb;        // matrix-ready interface blocks (see sbd_band.hpp), [ms][L][n][n] each (absent when gconly)
    double *bcb;            // [ms][n][n] bottom-boundary rows of sbd_band4.hpp (gconly)
    double *gcc;            // [ms][L][2][nn][nn] (gconly) GC's two independent quarters per layer: GC(iq+nn, jq+nn) = -GC(nn+1-iq,
                            //   nn+1-jq) = [0][iq-1][jq-1] and GC(nn+1-iq, jq+nn) = -GC(iq+nn, nn+1-jq) = [1][iq-1][jq-1]
                            //   (disort.f:3290-3312): what sbd_band4.hpp reads -- half the bytes of GC
    double *gu, *zb, *z0u, *z1u, *uum;
    // bidirectional surface (sbd_surface.hpp): model 1..3 (0 = Lambertian), its run parameters, per-item ocean
    // constants [nslot][4], and SURFAC's tables per (item or run, mode)
    int32_t ibdrf, brdf_shared, brdf_bad;   // brdf_bad: CHEKIN rejected the (wavelength-independent) model
    double bpar[8];
    const double *bitem;
    double *bdr, *bem, *rmu, *emu;
    // IBCND = 1 (ALBTRN): every work item is solved twice, lit isotropically from the top (even slots) and from the
    // bottom (odd slots), without beam, thermal source, surface or LYRCUT; slot_base = global index of the pass's first slot
    int32_t ibcnd, slot_base;
    int32_t *pivdbg;        // [ms][L*n] register index of each pivot row (band4_kernel<.., PIVDBG>), tests only
    // outputs (offset to the chunk)
    double *flux, *uu; int32_t *status;
};

// list system ms = slot * nmode + mazim for band_rcond_kernel (once: rcflag dedupes); any lane of any kernel of the pass
SBD_DEVICE void rcond_candidate(const Params &P, long long ms)
{
    if (atomicExch(&P.rcflag[ms], 1) != 1) P.rclist[1 + atomicAdd(&P.rclist[0], 1)] = (int32_t)ms;
}

// index of the [L][nmom+1] block of moments that belongs to work item `slot` of the pass
SBD_DEVICE size_t pmom_item(const Params &P, int slot) { return P.pmom_row ? (size_t)P.pmom_row[slot] : (size_t)slot; }

// ---- per-solve vector block (doubles) ----
// offsets inside sv[slot]
struct SV {
    int L;
    __host__ __device__ explicit SV(int L_) : L(L_) {}
    __host__ __device__ int ssalb()  const { return 0; }              // [L]  dithered
    __host__ __device__ int dtaucp() const { return L; }              // [L]
    __host__ __device__ int taucpr() const { return 2 * L; }          // [L+1]
    __host__ __device__ int oprim()  const { return 3 * L + 1; }      // [L]
    __host__ __device__ int flyr()   const { return 4 * L + 1; }      // [L]
    __host__ __device__ int expbea() const { return 5 * L + 1; }      // [L+1]
    __host__ __device__ int pkag()   const { return 6 * L + 2; }      // [L+1]
    __host__ __device__ int xr0()    const { return 7 * L + 3; }      // [L]
    __host__ __device__ int xr1()    const { return 8 * L + 3; }      // [L]
    __host__ __device__ int utau()   const { return 9 * L + 3; }      // [L+1]
    __host__ __device__ int utaupr() const { return 10 * L + 4; }     // [L+1]
    __host__ __device__ int bplank() const { return 11 * L + 5; }
    __host__ __device__ int tplank() const { return 11 * L + 6; }
    __host__ __device__ int size()   const { return 11 * L + 8; }
};
// svi[slot]: 0 ncut, 1 lyrcut, 2 status, 3.. layru[L+1]
#define SBD_SVI_NCUT 0
#define SBD_SVI_LYRCUT 1
#define SBD_SVI_STATUS 2
#define SBD_SVI_NAZ 3        /* highest azimuth mode of the item that can differ from zero (setup_kernel) */
#define SBD_SVI_LAYRU 4

// Lanes of one wave exchange data through LDS without a hardware barrier: LDS
// instructions of a wave execute in order, so a compiler+counter fence suffices.
SBD_DEVICE void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

SBD_DEVICE double dsign(double a, double b) { return (b >= 0.0) ? fabs(a) : -fabs(a); }

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <int N, int I = 0, class F>
SBD_DEVICE void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

}  // namespace sbd
