// sbdart_amd -- host side of the C ABI (include/sbdart_amd.h): per-run tables, HBM
// workspace, chunked kernel pipeline on one HIP stream, timing with HIP events.
//
// Pipeline per pass of work items (all on the caller's / engine's stream):
//   setup_kernel     (disort.f:482-571)   -> sv/svi
//   layer_kernel2    (disort.f:638-693)   -> kk, ek, ga/gb, zz, zp0/1 [, gc, gu, zb, z0u, z1u]
//   layer_kernel     (same, reference algorithm) for the layers layer_kernel2 listed
//   band_kernel      (disort.f:701-721)   -> U factor, eliminated right-hand side
//   backsolve_kernel (same + FLUXES)      -> ll, flux
//   usrint_kernel + azimuth_kernel (disort.f:745-825), radiance mode only -> uu
//   finish_kernel                         -> status
#include "../../include/sbdart_amd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen()ed when a fleet needs it
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "sbd_common.hpp"
#include "sbd_launch.hpp"
#include "sbd_band.hpp"     // BandLds / SolveLds / u_width (host-side sizes)
#include "sbd_layer.hpp"    // LayerLds
#include "sbd_layer2.hpp"   // Layer2Lds
#include "sbd_surface.hpp"  // bidirectional surfaces: surfac_kernel, host-side check of the model
#include "sbd_hosttables.hpp" // QGAUSN / LEPOLY on the host (also behind sbd_band_rcond_host)
#include "sbd_scat_types.hpp" // the scatterers' kernel (sbd_k_scat.hip on sbd_scat.hpp)
#include "sbd_gas_types.hpp" // the gas model's launch interface (its source, sbd_gas.hpp, is compiled without contraction in sbd_k_gas.hip)
static_assert(sbd::SBD_NFLUX_ == SBD_NFLUX, "flux component count");

namespace {
using sbd::hosttab::gauss01;
using sbd::hosttab::legendre_norm;
using sbd::hosttab::ref_pi;

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(SBD_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

__global__ void finish_kernel(sbd::Params P)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= P.nslot) return;
    P.status[slot] = P.svi[(size_t)slot * P.svi_stride + SBD_SVI_STATUS];
}

// stdout1's weighted sums (drt.f:964-1054), two fixed-order levels: 256-item segments summed in work-item order,
// then segments summed in order.  Both kernels stage their operands through LDS with coalesced loads and let one
// thread per output element add them up IN ORDER from there (the sums are the same bits as a serial loop over the
// items; read straight from HBM by ten threads that loop took 80 + 90 us of the 9.7 ms step).
constexpr int kAccTile = 256;
__global__ void __launch_bounds__(256) accum_partial_kernel(int nwork, int nel, const double *w, const double *x, double *partial)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];      // [kAccTile][ne] products of one tile of elements
    const int seg = blockIdx.x;
    const int i0 = seg * 256, cnt = ((i0 + 256 < nwork) ? 256 : nwork - i0);
    const int ET = 16;                                               // elements per LDS tile
    for (int e0 = 0; e0 < nel; e0 += ET) {
        const int ne = (nel - e0 < ET) ? nel - e0 : ET;
        for (int t = threadIdx.x; t < cnt * ne; t += blockDim.x) {
            const int i = t / ne, e = t % ne;
            sh[i * ET + e] = w[i0 + i] * x[(size_t)(i0 + i) * nel + e0 + e];
        }
        __syncthreads();
        if ((int)threadIdx.x < ne) {
            double acc = 0.0;
            for (int i = 0; i < cnt; ++i) acc = acc + sh[i * ET + threadIdx.x];
            partial[(size_t)seg * nel + e0 + threadIdx.x] = acc;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) accum_final_kernel(int nseg, int nel, const double *partial, double *acc)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];      // [kAccTile segments][ET]
    const int ET = 16, e0 = blockIdx.x * ET;
    const int ne = (nel - e0 < ET) ? nel - e0 : ET;
    double a = ((int)threadIdx.x < ne) ? acc[e0 + threadIdx.x] : 0.0;
    for (int s0 = 0; s0 < nseg; s0 += kAccTile) {
        const int ns = (nseg - s0 < kAccTile) ? nseg - s0 : kAccTile;
        for (int t = threadIdx.x; t < ns * ne; t += blockDim.x) {
            const int sg = t / ne, e = t % ne;
            sh[sg * ET + e] = partial[(size_t)(s0 + sg) * nel + e0 + e];
        }
        __syncthreads();
        if ((int)threadIdx.x < ne)
            for (int sg = 0; sg < ns; ++sg) a = a + sh[sg * ET + threadIdx.x];
        __syncthreads();
    }
    if ((int)threadIdx.x < ne) acc[e0 + threadIdx.x] = a;
}

// The compact form of a batch (sbd_mix_in, ABI v6) -> DISORT's arguments, on the device: what depthscl (taugas.f:7598-7603),
// GETMOM (disutil.f:2104-2209), taucloud / tauaero's accumulation (taucloud.f:103, 132; tauaero.f:1300, 1330) and normom
// (drt.f:1390-1395) do on the host for every (wavelength, k-term), in the reference's own association.  A block per
// work item, a thread per layer: DTAUC and SSALB of the item; the FIRST item of a spectral point (in this launch) also
// forms the point's block of moments, which its k-terms share (pmom_row).  Integer powers of g the way the reference's
// compiler forms GG**K (square-and-multiply from the low bit, compiler-rt's __powidf2), the Rayleigh 0.1 as the REAL*4
// literal it is there: the oracle's restatement (oracle/mix_restatement.py) is bit-equal.
__device__ __forceinline__ double powi_like_fortran(double a, int b)
{
#pragma clang fp contract(off)
    double r = 1.0;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}
struct MixFamilies { int32_t f[SBD_MIX_MAX_TERMS]; };
// point blocks: lay[(p - pbase)][4 + 3 nterm][L]; outputs: dtauc / ssalb [item], pmom [(p - pbase)], pmom_row = p - pbase
__global__ void __launch_bounds__(64) assemble_kernel(int w0, int nitem, int L, int nmom, int first_point_done, int pbase, int lay_base,
                                                      int nterm, MixFamilies fam, const int32_t *kterm, const double *gslots, int gas_p0,
                                                      const int32_t *point_of, const double *dtaug, const double *lay,
                                                      const double *plo, const double *phi_, const double *pfb, const double *pal,
                                                      const uint8_t *ppl, double *dtauc, double *ssalb, double *pmom,
                                                      int32_t *pmom_row, double *wvnmlo, double *wvnmhi, double *fbeam,
                                                      double *albedo, uint8_t *plank)
{
#pragma clang fp contract(off)      // (the host forms these products and sums one rounding at a time: so must this kernel)
    const int w = w0 + blockIdx.x;
    if (blockIdx.x >= nitem) return;
    const int p = point_of[w] - pbase;
    // the point's moments: by its first item in this launch -- unless the launch before already made them
    const bool first = (blockIdx.x == 0) ? (first_point_done == 0) : (point_of[w - 1] - pbase != p);
    const int nch = 4 + 3 * nterm;
    const double *blk = lay + (size_t)(point_of[w] - lay_base) * nch * L;       // (staged per call from pbase, or the gas call's resident copy)
    // the item's gas: from the host (dtaug), or what the gas kernel left on this device for (point, k-term)
    const double *gas = gslots ? gslots + ((size_t)(point_of[w] - gas_p0) * 3 + kterm[w]) * L : dtaug + (size_t)w * L;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const double dc = blk[l], da = blk[L + l], dr = blk[2 * L + l], scat = blk[3 * L + l];
        const double dt = ((gas[l] + dc) + da) + dr;                                       // taugas.f:7598
        dtauc[(size_t)w * L + l] = dt;
        ssalb[(size_t)w * L + l] = (dt > 2.2250738585072014e-308) ? scat / dt : 0.0;      // (tiny(1.d0): taugas.f:7599)
        if (first) {
            double *pm = pmom + ((size_t)p * L + l) * (nmom + 1);
            pm[0] = 1.0;
            for (int k = 1; k <= nmom; ++k) {
                double q = 0.0;
                for (int t = 0; t < nterm; ++t) {
                    const double *tc = blk + (size_t)(4 + 3 * t) * L;
                    const int f = fam.f[t];
                    const double pk = (f == 3) ? powi_like_fortran(tc[l], k) : ((f == 2 && k == 2) ? (double)0.1f : 0.0);
                    q = q + (pk * tc[L + l]) * tc[2 * L + l];
                }
                if (k == 2) q = q + (double)0.1f * dr;                                      // drt.f:1391
                pm[k] = (scat != 0.0) ? q / scat : q;                                       // drt.f:1392-1393
            }
        }
    }
    if (threadIdx.x == 0) {
        pmom_row[w] = p;
        wvnmlo[w] = plo[p]; wvnmhi[w] = phi_[p]; fbeam[w] = pfb[p]; albedo[w] = pal[p]; plank[w] = ppl[p];
    }
}

// IBCND = 1: every work item becomes two consecutive internal items with the same optical properties, no beam,
// no thermal source and a black surface (the surface's albedo enters ALBTRN's closing formulas only)
__global__ void ibcnd_expand_kernel(int nwork, int L, int npm, const double *dt, const double *ss, const double *pm,
                                    const double *lo, const double *hi, double *dt2, double *ss2, double *pm2,
                                    double *lo2, double *hi2, double *fb2, double *al2, uint8_t *pl2)
{
    const size_t tot = (size_t)nwork * 2 * npm;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
        const size_t s = t / npm, k = t % npm, i = s / 2;
        pm2[t] = pm[i * npm + k];
        if (k < (size_t)L) { dt2[s * L + k] = dt[i * L + k]; ss2[s * L + k] = ss[i * L + k]; }
        if (k == 0) { lo2[s] = lo[i]; hi2[s] = hi[i]; fb2[s] = 0.0; al2[s] = 0.0; pl2[s] = 0; }
    }
}
// ... and ALBTRN's results from the two solves (disort.f:6890-6990): the azimuthally averaged upward intensity at the
// top for illumination from the top is the albedo for beam incidence at that angle, for illumination from the bottom
// (USRINT's boundary term carries the directly transmitted part) the transmissivity; a reflecting surface is put
// back analytically with the spherical albedo / transmissivity of the medium = the second solve's hemispheric
// fluxes (SPALTR, disort.f:7319-7432, = FLUXES' sums over pi).
__global__ void ibcnd_combine_kernel(int nwork, int nout, int numu2, double pi, const double *albedo, const double *flux2,
                                     const double *uu2, const int32_t *st2, double *albtrn, double *flux, int32_t *status)
{
    const int i = blockIdx.x;
    if (i >= nwork) return;
    const double a = albedo[i];
    const double *f2 = flux2 + (size_t)(2 * i + 1) * SBD_NFLUX * 2;        // bottom-lit slot: [5][2 levels]
    const double sphtrn = f2[SBD_FLUP * 2 + 0] / pi, sphalb = f2[SBD_RFLDN * 2 + 1] / pi;
    for (int iu = threadIdx.x; iu < nout; iu += blockDim.x) {
        double alb = uu2[((size_t)(2 * i) * 2 + 0) * numu2 + (numu2 / 2 + iu)];       // [slot][1 azimuth][2 levels][numu2]
        double trn = uu2[((size_t)(2 * i + 1) * 2 + 0) * numu2 + (numu2 / 2 + iu)];
        if (a > 0.0) {
            alb = alb + (a / (1.0 - a * sphalb)) * sphtrn * trn;
            trn = trn + (a / (1.0 - a * sphalb)) * sphalb * trn;
        }
        albtrn[((size_t)i * 2 + 0) * nout + iu] = alb;
        albtrn[((size_t)i * 2 + 1) * nout + iu] = trn;
    }
    if (threadIdx.x == 0) {
        int st = st2[2 * i] | st2[2 * i + 1];
        if (a < 0.0 || a > 1.0) st |= 0x20;                               // CHEKIN (disort.f:5099-5103)
        status[i] = st;
        if (flux) for (int k = 0; k < SBD_NFLUX * 2; ++k) flux[(size_t)i * SBD_NFLUX * 2 + k] = 0.0;
    }
}

}  // namespace


struct sbd_engine {
    sbd_run_cfg cfg{};
    int n = 0, nn = 0, L = 0, nmode = 1, naz_run = 0, nlev = 0, G = 0, G2 = 0;
    int chunk = 0;
    size_t ws_bytes = 0;
    hipStream_t stream = nullptr;
    std::vector<double> h_cmu, h_cwt;
    // device tables
    double *d_tab = nullptr;      // one allocation for all double tables
    int32_t *d_level = nullptr;
    sbd::Tables tab{};
    // workspace (one allocation)
    char *d_ws = nullptr;
    sbd::Params P{};              // workspace pointers + constants; chunk fields filled per call
    sbd::Params P2{};             // the same over the second workspace (passes alternate between the two)
    hipStream_t aux = nullptr;    // second stream: odd passes run here, beside the even ones
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // host-pointer solves: ONE stream carries the H2D copies of all passes, back to back, each followed by
    // an event its pass waits for -- the inputs of pass i+1 cross PCIe while pass i computes (with the copies
    // on the passes' own two streams both streams copied, then both computed: nothing overlapped)
    hipStream_t copy = nullptr;
    std::vector<hipEvent_t> ev_h2d;
    // staging for solve_host / accumulate
    char *d_stage = nullptr;
    size_t stage_bytes = 0;
    // Outputs of a host-pointer solve land in a pinned buffer of the engine and move to the caller's arrays
    // after the stream has drained: a D2H copy into pageable memory is staged by the runtime and blocks the
    // enqueueing thread until the pass has finished -- which serialises the passes (H2D of pass i+1 would wait).
    char *h_pin = nullptr;
    size_t h_pin_bytes = 0;
    struct PendingOut { void *dst; const void *src; size_t bytes; };
    std::vector<PendingOut> pending_out;
    double *d_partial = nullptr;
    size_t partial_elems = 0;
    double *d_acc = nullptr;      // [5*nlev + nphi*nlev*numu] weighted sums of the last fleet solve
    double *d_red = nullptr;      // same size: RCCL reduce result (root)
    // timing
    bool timing = false;            // per-kernel HIP events, ONE stream, a synchronisation per pass (the kernels of a pass alone on the chip)
    bool timing_inplace = false;    // ... or the events recorded where the passes run -- two streams, a pass's kernels beside the other
                                    // pass's -- and read afterwards: the durations rocprofv3's kernel trace of the same command shows
    std::vector<hipEvent_t> ev_ip;  // [pass][kPhases + 1]
    int ip_npass = 0;
    bool ip_pending = false;
    static constexpr int kPhases = 5;   // setup, layer, band LU, back-substitution + fluxes, intensities
    hipEvent_t ev[kPhases + 1] = {};
    float ms_phase[kPhases] = {};
    bool have_times = false;
    int layer_lds = 0, band_lds = 0, solve_lds = 0, usr_lds = 0, layer2_lds = 0;
    bool solve_v1 = false;
    bool pivot_exact = false;       // sbd_run_cfg::pivot_exact / SBD_EXACT_PIVOT: ISAMAX's rule in band4_kernel
    int32_t *h_hint = nullptr;      // [4] pinned, device-visible: length of the fallback list of the workspace's last pass (-1: not known yet),
                                    // then [2..3] the same for band_rcond_kernel's list
    int32_t *d_eigflag = nullptr;
    bool use_layer2 = true;
    bool band_reg = false;
    bool band4 = false;             // four systems per wave, block form (sbd_band4.hpp), NSTR <= 16
    bool band1 = false;             // one system per wave, block form in registers (sbd_band1.hpp), 16 < NSTR <= 32
    bool band_rows = false;         // one system per wave, a row per lane (sbd_bandr.hpp), NSTR 34..40
    int32_t *d_surf_flag = nullptr; // shared surface tables: CHEKIN's verdict on the model (sbd_surface.hpp)
    double *d_surf = nullptr;       // shared surface tables (Hapke / Ross-Li: one set per run)
    bool brdf_bad = false;          // ... the model's flux albedo leaves [0,1]: every item gets SBD_ST_ERR_INPUT
    bool fused = false;             // band4 / band1, flux-only, levels = {top of layer 1, surface}: the band kernel carries FLUXES'
                                    // functionals through the elimination -- no U factor, no back-substitution kernel
    int ibcnd = 0, ib_nout = 0;     // IBCND = 1 (ALBTRN): results at ib_nout cosines; the engine proper runs the doubled batch
    std::vector<double> ib_umu;     // ... at -umu reversed | +umu
    char *d_ib = nullptr;           // ... its doubled inputs and internal outputs
    size_t ib_bytes = 0;
    bool quad = false;              // radiances at the quadrature angles (USRANG = false): CMPINT instead of TERPEV/TERPSO/USRINT
    bool corint = false;            // intensity corrections after the azimuth series (sbd_intcor.hpp)
    int32_t *d_pivdbg = nullptr;    // sbd_engine_debug_pivots: [2][chunk * nmode][L * n]
    // the gas depths sbd_fleet_gas_terms left on this device: [gas_np][3][L] for the points gas_p0 .. gas_p0 + gas_np - 1 of that call
    double *d_gas_slots = nullptr;
    int32_t gas_p0 = 0, gas_np = 0;
    double *d_gas_lay = nullptr;    // ... and those points' layer blocks [gas_np][gas_nch][L], kept for the solves that follow
    int32_t gas_nch = 0;
    // errmsg 2 on LINPACK's own estimate (sbd_refband.hpp): scratch of band_rcond_kernel, kRcBlocks blocks per workspace
    int rc_blocks = 8;              // blocks of band_rcond_kernel per workspace (scratch: <= 1 GB per workspace)
    double *d_rb = nullptr;
    size_t rb_stride = 0;           // doubles per block
    double *d_rcdbg = nullptr;      // [2][chunk * nmode] the estimates of the systems served (tests; NaN where none)
    int64_t gas_token = 0;          // generation number of d_gas_lay (sbd_fleet_gas_terms hands it to the caller; a solve names it
                                    // in sbd_mix_in::lay_token to read the resident blocks -- never inferred from pointers)
    std::vector<int32_t> gas_nk;    // [gas_np] number of k-terms of those points (a solve's kterm is checked against it)
    int64_t fallback_layers = 0;    // timing mode: (item, mode, layer) problems of the last solve left to the QR kernel
};

extern "C" {

int32_t sbd_abi_version(void) { return SBD_ABI_VERSION; }

const char *sbd_last_error(void) { return g_last_error.c_str(); }

const char *sbd_strerror(int code)
{
    switch (code) {
    case SBD_OK: return "ok";
    case SBD_E_INVALID: return "invalid argument or run configuration";
    case SBD_E_RETRY_NSTR: return "beam angle equals a quadrature angle: change NSTR (disort.f:2645-2650)";
    case SBD_E_NO_DEVICE: return "no usable HIP device";
    case SBD_E_HIP: return "HIP runtime error";
    case SBD_E_UNSUPPORTED: return "feature outside the hot-path scope (BRDF / IBCND=1)";
    case SBD_E_NOMEM: return "out of device memory";
    default: return "unknown error";
    }
}

void sbd_engine_destroy(sbd_engine *e)
{
    if (!e) return;
    if (e->d_tab) (void)hipFree(e->d_tab);
    if (e->d_level) (void)hipFree(e->d_level);
    if (e->d_ws) (void)hipFree(e->d_ws);
    if (e->d_stage) (void)hipFree(e->d_stage);
    if (e->d_gas_slots) (void)hipFree(e->d_gas_slots);
    if (e->d_gas_lay) (void)hipFree(e->d_gas_lay);
    if (e->d_rb) (void)hipFree(e->d_rb);
    if (e->d_rcdbg) (void)hipFree(e->d_rcdbg);
    for (hipEvent_t ev : e->ev_ip) (void)hipEventDestroy(ev);
    if (e->h_pin) (void)hipHostFree(e->h_pin);
    if (e->h_hint) (void)hipHostFree(e->h_hint);
    if (e->d_partial) (void)hipFree(e->d_partial);
    if (e->d_acc) (void)hipFree(e->d_acc);
    if (e->d_red) (void)hipFree(e->d_red);
    if (e->d_pivdbg) (void)hipFree(e->d_pivdbg);
    if (e->d_surf) (void)hipFree(e->d_surf);
    if (e->d_ib) (void)hipFree(e->d_ib);
    if (e->d_surf_flag) (void)hipFree(e->d_surf_flag);
    for (auto &x : e->ev)
        if (x) (void)hipEventDestroy(x);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->aux) (void)hipStreamDestroy(e->aux);
    for (auto &x : e->ev_h2d) if (x) (void)hipEventDestroy(x);
    if (e->copy) (void)hipStreamDestroy(e->copy);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int sbd_engine_create(const sbd_run_cfg *cfg, sbd_engine **out)
{
    if (!cfg || !out) return fail(SBD_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->abi_version != SBD_ABI_VERSION) return fail(SBD_E_INVALID, "abi_version mismatch");
    if (cfg->ibcnd == 1) {
        // IBCND = 1 (ALBTRN, disort.f:6718-7432): albedo and transmissivity of the whole medium.  The engine proper is
        // created for the equivalent run of the general case -- two internal items per work item (lit isotropically
        // from the top / from the bottom), no beam, no thermal source, black surface, no LYRCUT, one azimuth mode,
        // intensities at -umu reversed | +umu (SETDIS, disort.f:2672-2687) at the top level, fluxes at top and bottom
        // (ibcnd = 2 marks it) -- and ibcnd_combine_kernel closes with ALBTRN's formulas.
        const int n1 = cfg->nstr, nn1 = n1 / 2;
        if (n1 < 4 || n1 > SBD_MAX_NSTR || (n1 & 1)) return fail(SBD_E_INVALID, "NSTR must be even, 4..40");
        if (cfg->usrang && cfg->onlyfl)
            return fail(SBD_E_UNSUPPORTED, "IBCND = 1 with USRANG and ONLYFL: the reference doubles NSTR angles into UMU(MAXUMU) (disort.f:2655-2687)");
        std::vector<double> um;
        int nout = 0;
        if (cfg->usrang) {
            if (cfg->numu < 1 || 2 * cfg->numu > SBD_MAX_NSTR || !cfg->umu) return fail(SBD_E_INVALID, "IBCND = 1: NUMU/UMU");
            for (int i = 0; i < cfg->numu; ++i) {
                if (!(cfg->umu[i] > 0.0) || cfg->umu[i] > 1.0) return fail(SBD_E_INVALID, "IBCND = 1: UMU must be positive cosines");
                if (i && cfg->umu[i] < cfg->umu[i - 1]) return fail(SBD_E_INVALID, "UMU must ascend");
            }
            nout = cfg->numu;
            um.resize(2 * nout);
            for (int i = 0; i < nout; ++i) { um[nout + i] = cfg->umu[i]; um[i] = -cfg->umu[nout - 1 - i]; }
        } else {
            std::vector<double> c(nn1), wgt(nn1);
            gauss01(nn1, c.data(), wgt.data());
            nout = nn1;
            um.resize(n1);
            for (int i = 0; i < nn1; ++i) { um[i] = -c[nn1 - 1 - i]; um[nn1 + i] = c[i]; }
        }
        const double phi0 = 0.0;
        const int32_t lev[2] = {0, cfg->nlyr};
        sbd_run_cfg c2 = *cfg;
        c2.ibcnd = 2;
        c2.onlyfl = 0; c2.usrang = 1; c2.lamber = 1; c2.ibdrf = 0; c2.corint = 0;
        c2.numu = (int32_t)um.size(); c2.umu = um.data();
        c2.nphi = 1; c2.phi = &phi0; c2.phi0 = 0.0;
        c2.nlevel_out = 2; c2.level_out = lev;
        c2.fisot = 0.0; c2.temis = 0.0; c2.umu0 = 1.0;
        if (c2.max_batch > 0) c2.max_batch = (c2.max_batch < (1 << 29)) ? 2 * c2.max_batch : c2.max_batch;
        const int rc = sbd_engine_create(&c2, out);
        if (*out) {
            (*out)->ibcnd = 1;
            (*out)->ib_nout = nout;
            (*out)->ib_umu = um;
            (*out)->cfg.umu = nullptr; (*out)->cfg.phi = nullptr; (*out)->cfg.level_out = nullptr;   // (locals of this call)
        }
        return rc == SBD_E_RETRY_NSTR ? SBD_OK : rc;     // (no beam: its angle never meets a quadrature angle)
    }
    const int n = cfg->nstr, L = cfg->nlyr;
    // CHEKIN's per-run checks (disort.f:4926-5140)
    if (n < 4 || n > SBD_MAX_NSTR || (n & 1)) return fail(SBD_E_INVALID, "NSTR must be even, 4..40");
    if (L < 1 || L > SBD_MAX_NLYR) return fail(SBD_E_INVALID, "NLYR out of range");
    if (cfg->nmom < 0) return fail(SBD_E_INVALID, "NMOM < 0");
    if (!cfg->lamber && (cfg->ibdrf < 1 || cfg->ibdrf > 3)) return fail(SBD_E_INVALID, "lamber = 0 needs ibdrf = 1, 2 or 3");
    if (cfg->lamber && cfg->ibdrf != 0) return fail(SBD_E_INVALID, "ibdrf set with lamber = 1");
    if (!cfg->temper) return fail(SBD_E_INVALID, "temper is NULL");
    const bool rad = !cfg->onlyfl;
    const bool quad = rad && !cfg->usrang;     // intensities at the quadrature angles (CMPINT, disort.f:1658-1778)
    if (rad) {
        if (!quad && (cfg->numu < 1 || cfg->numu > SBD_MAX_NSTR || !cfg->umu)) return fail(SBD_E_INVALID, "NUMU/UMU");
        if (cfg->nphi < 1 || cfg->nphi > SBD_MAX_NSTR || !cfg->phi) return fail(SBD_E_INVALID, "NPHI/PHI");
        for (int i = 0; !quad && i < cfg->numu; ++i) {
            if (cfg->umu[i] < -1.0 || cfg->umu[i] > 1.0 || cfg->umu[i] == 0.0) return fail(SBD_E_INVALID, "UMU range");
            if (i && cfg->umu[i] < cfg->umu[i - 1]) return fail(SBD_E_INVALID, "UMU must ascend");
        }
        for (int j = 0; j < cfg->nphi; ++j)
            if (cfg->phi[j] < 0.0 || cfg->phi[j] > 360.0) return fail(SBD_E_INVALID, "PHI range");
        if (cfg->phi0 < 0.0 || cfg->phi0 > 360.0) return fail(SBD_E_INVALID, "PHI0 range");
    }
    if (cfg->fisot < 0.0) return fail(SBD_E_INVALID, "FISOT < 0");
    if (cfg->temis < 0.0 || cfg->temis > 1.0) return fail(SBD_E_INVALID, "TEMIS range");
    if (cfg->nlevel_out < 0 || cfg->nlevel_out > L + 1) return fail(SBD_E_INVALID, "nlevel_out");
    if (cfg->nlevel_out > 0) {
        if (!cfg->level_out) return fail(SBD_E_INVALID, "level_out is NULL");
        for (int i = 0; i < cfg->nlevel_out; ++i)
            if (cfg->level_out[i] < 0 || cfg->level_out[i] > L) return fail(SBD_E_INVALID, "level_out range");
    }

    int ndev = 0;
    // SBD_TIMING: where the creation's time goes (ms since its first statement, to stderr) -- the runtime's bring-up is the
    // first HIP call of a process, the kernels' code objects are loaded by the first hipFuncSetAttribute / launch of a family
    const bool t_on = getenv("SBD_TIMING") != nullptr && getenv("SBD_TIMING_CREATE") != nullptr;
    const auto t_c0 = std::chrono::steady_clock::now();
    std::string t_log;
    auto t_mark = [&](const char *what) {
        if (!t_on) return;
        char b[96];
        snprintf(b, sizeof(b), " %s %.1f", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c0).count());
        t_log += b;
    };
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SBD_E_NO_DEVICE, "hipGetDeviceCount");
    t_mark("hipGetDeviceCount");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(SBD_E_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device));

    sbd_engine *e = new (std::nothrow) sbd_engine;
    if (!e) return fail(SBD_E_NOMEM, "host allocation");
    e->cfg = *cfg;
    e->n = n;
    e->nn = n / 2;
    e->L = L;
    const int nn = e->nn;
    e->nlev = cfg->nlevel_out > 0 ? cfg->nlevel_out : L + 1;
    const int numu = rad ? (quad ? n : cfg->numu) : 0, nphi = rad ? cfg->nphi : 0;
    const bool rad_user = rad && !quad;        // TERPEV / TERPSO / USRINT: the user-angle machinery

    // quadrature (SETDIS, disort.f:2629-2638)
    e->h_cmu.resize(n);
    e->h_cwt.resize(n);
    gauss01(nn, e->h_cmu.data(), e->h_cwt.data());
    for (int i = 0; i < nn; ++i) { e->h_cmu[i + nn] = -e->h_cmu[i]; e->h_cwt[i + nn] = e->h_cwt[i]; }

    // USRANG = false: the output angles are the quadrature angles, downward first (SETDIS, disort.f:2655-2669)
    std::vector<double> qumu(n);
    for (int iu = 0; iu < nn; ++iu) { qumu[iu] = -e->h_cmu[nn - 1 - iu]; qumu[nn + iu] = e->h_cmu[iu]; }
    const double *umu_ptr = quad ? qumu.data() : cfg->umu;
    // azimuth modes (disort.f:577-586): per-run part of the NAZ rule
    int naz = n - 1;
    {
        const double e5 = (double)1.0e-5f;
        if (fabs(1.0 - cfg->umu0) < e5 || cfg->onlyfl || cfg->ibcnd == 2
            || (numu == 1 && fabs(1.0 - umu_ptr[0]) < e5) || (numu == 1 && fabs(1.0 + umu_ptr[0]) < e5)
            || (numu == 2 && fabs(1.0 + umu_ptr[0]) < e5 && fabs(1.0 - umu_ptr[1]) < e5))
            naz = 0;
    }
    e->naz_run = naz;
    e->nmode = naz + 1;
    const int nmode = e->nmode;

    // Legendre tables for every mode
    std::vector<double> ylmc((size_t)nmode * n * (n + 1), 0.0), ylm0((size_t)nmode * (n + 1), 0.0);
    std::vector<double> ylmu((size_t)nmode * (numu > 0 ? numu : 1) * (n + 1), 0.0);
    std::vector<double> cosm((size_t)nmode * (nphi > 0 ? nphi : 1), 1.0);
    {
        std::vector<double> yc((size_t)n * (n + 1), 0.0), y0(n + 1, 0.0), yu((size_t)(numu > 0 ? numu : 1) * (n + 1), 0.0);
        const double ang0 = -cfg->umu0;
        for (int m = 0; m < nmode; ++m) {
            legendre_norm(1, m, n, n - 1, &ang0, y0.data());
            if (numu > 0) legendre_norm(numu, m, n, n - 1, umu_ptr, yu.data());
            legendre_norm(nn, m, n, n - 1, e->h_cmu.data(), yc.data());
            double sgn = -1.0;   // mirror to -mu (disort.f:611-627)
            for (int l = m; l <= n - 1; ++l) {
                sgn = -sgn;
                for (int iq = nn; iq < n; ++iq) yc[(size_t)iq * (n + 1) + l] = sgn * yc[(size_t)(iq - nn) * (n + 1) + l];
            }
            memcpy(&ylmc[(size_t)m * n * (n + 1)], yc.data(), sizeof(double) * n * (n + 1));
            memcpy(&ylm0[(size_t)m * (n + 1)], y0.data(), sizeof(double) * (n + 1));
            if (numu > 0) memcpy(&ylmu[(size_t)m * numu * (n + 1)], yu.data(), sizeof(double) * numu * (n + 1));
        }
        const double rpd = ref_pi() / 180.0;   // disort.f:450
        for (int m = 0; m < nmode; ++m)
            for (int j = 0; j < nphi; ++j)
                cosm[(size_t)m * nphi + j] = (m == 0) ? 1.0 : cos((double)m * (rpd * (cfg->phi[j] - cfg->phi0)));
    }

    // upload tables
    const size_t ntab = (size_t)2 * n + ylmc.size() + ylm0.size() + ylmu.size() + cosm.size() + (L + 1) + (numu > 0 ? numu : 1);
    std::vector<double> htab;
    htab.reserve(ntab);
    auto push = [&](const double *p, size_t cnt) { size_t off = htab.size(); htab.insert(htab.end(), p, p + cnt); return off; };
    const size_t o_cmu = push(e->h_cmu.data(), n), o_cwt = push(e->h_cwt.data(), n);
    const size_t o_ylmc = push(ylmc.data(), ylmc.size()), o_ylm0 = push(ylm0.data(), ylm0.size());
    const size_t o_ylmu = push(ylmu.data(), ylmu.size()), o_cos = push(cosm.data(), cosm.size());
    // cos(phi - phi0) for INTCOR's scattering angle (disort.f:2188-2190): filled whatever the number of azimuth modes
    std::vector<double> cphi1((size_t)(nphi > 0 ? nphi : 1), 1.0);
    for (int j = 0; j < nphi; ++j) cphi1[j] = cos((ref_pi() / 180.0) * (cfg->phi[j] - cfg->phi0));
    const size_t o_cphi = push(cphi1.data(), cphi1.size());
    std::vector<double> g50(2 * sbd::kSurfGauss, 0.0);
    gauss01(sbd::kSurfGauss / 2, g50.data(), g50.data() + sbd::kSurfGauss);
    for (int k = 0; k < sbd::kSurfGauss / 2; ++k) {
        g50[k + sbd::kSurfGauss / 2] = -g50[k];
        g50[sbd::kSurfGauss + k + sbd::kSurfGauss / 2] = g50[sbd::kSurfGauss + k];
    }
    const size_t o_g50 = push(g50.data(), g50.size());
    const size_t o_temper = push(cfg->temper, L + 1);
    double zero = 0.0;
    const size_t o_umu = numu > 0 ? push(umu_ptr, numu) : push(&zero, 1);
    const std::vector<double> zblock((size_t)n * n, 0.0);
    const size_t o_zero = push(zblock.data(), zblock.size());
    std::vector<double> tagblock((size_t)nn * nn, 0.0);
    for (int k = 0; k < 3 && k < nn; ++k)
        for (int c = 0; c < nn; ++c) tagblock[(size_t)k * nn + c] = (double)(k + 1);
    const size_t o_tags = push(tagblock.data(), tagblock.size());
#define CREATE_TRY(expr)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            sbd_engine_destroy(e);                                                          \
            return fail(e_ == hipErrorOutOfMemory ? SBD_E_NOMEM : SBD_E_HIP,                \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                 \
        }                                                                                   \
    } while (0)
    t_mark("host-tables");
    CREATE_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    t_mark("first-stream");
    CREATE_TRY(hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&e->copy, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    CREATE_TRY(hipMalloc(&e->d_tab, htab.size() * sizeof(double)));
    CREATE_TRY(hipMemcpy(e->d_tab, htab.data(), htab.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int32_t> hlev(e->nlev);
    for (int i = 0; i < e->nlev; ++i) hlev[i] = cfg->nlevel_out > 0 ? cfg->level_out[i] : i;
    CREATE_TRY(hipMalloc(&e->d_level, sizeof(int32_t) * e->nlev));
    CREATE_TRY(hipMemcpy(e->d_level, hlev.data(), sizeof(int32_t) * e->nlev, hipMemcpyHostToDevice));
    t_mark("tables-up");
    e->tab.cmu = e->d_tab + o_cmu;
    e->tab.cwt = e->d_tab + o_cwt;
    e->tab.ylmc = e->d_tab + o_ylmc;
    e->tab.ylm0 = e->d_tab + o_ylm0;
    e->tab.ylmu = e->d_tab + o_ylmu;
    e->tab.cosmphi = e->d_tab + o_cos;
    e->tab.cosphi = e->d_tab + o_cphi;
    e->tab.gmu50 = e->d_tab + o_g50;
    e->tab.gwt50 = e->d_tab + o_g50 + sbd::kSurfGauss;
    e->tab.zeros = e->d_tab + o_zero;
    e->tab.tags = e->d_tab + o_tags;
    e->tab.temper = e->d_tab + o_temper;
    e->tab.umu = e->d_tab + o_umu;
    e->tab.level_out = e->d_level;

    // ---- workspace ----
    const int ncd = 3 * nn - 1, cw = 2 * ncd + 1;
    const sbd::SV sv(L);
    const int sv_stride = (sv.size() + 1) & ~1;
    const int svi_stride = (SBD_SVI_LAYRU + L + 1 + 3) & ~3;
    // (band4: the band kernel reads GC and scales it itself, no ga/gb blocks -- a third of the workspace)
    bool band4 = nn <= 8;
    bool band1 = nn >= 9 && nn <= 16;
    bool band_rows = sbd::has_band_rows(nn);
    if (const char *s = getenv("SBD_BAND_V1")) { band4 = band4 && atoi(s) == 0; band1 = band1 && atoi(s) == 0; band_rows = band_rows && atoi(s) == 0; }
    // fluxes at the top of the first layer and at the surface only (IOUT 1 / 10 with the default ZOUT): fused band kernel
    bool fused = (band4 || band1 || band_rows) && nn >= 3 && cfg->onlyfl && cfg->nlevel_out == 2 && cfg->level_out[0] == 0 && cfg->level_out[1] == L;
    if (const char *s = getenv("SBD_NO_FUSE")) fused = fused && atoi(s) == 0;
    e->pivot_exact = cfg->pivot_exact != 0;
    if (const char *s = getenv("SBD_EXACT_PIVOT")) e->pivot_exact = e->pivot_exact || atoi(s) != 0;
    if (const char *s = getenv("SBD_SOLVE_V1")) e->solve_v1 = atoi(s) != 0;   // (developer switch: the column-oriented back-substitution also for NSTR 18-32)
    const size_t nblk = (band4 || band1) ? 1 : 3;          // GC alone, or GC + the matrix-ready interface blocks ga / gb
    const size_t per_ms = sizeof(double) * (nblk * L * n * n + (band4 ? (size_t)n * n + (size_t)L * 2 * nn * nn : 0) + (band1 ? (size_t)2 * n * n + (size_t)L * 2 * nn * nn : 0) + (size_t)L * n * 6 + (size_t)L * nn + (fused ? 0 : (size_t)L * n * (2 * n))
                                            + (rad_user ? (size_t)L * n * numu + 3 * (size_t)L * numu : 0) + (rad ? (size_t)e->nlev * numu : 0));
    const bool brdf = !cfg->lamber, brdf_item = brdf && cfg->ibdrf == 1;      // (the ocean's tables follow the wavelength)
    const size_t numu1 = numu > 0 ? (size_t)numu : 1;     // (a flux-only run still carves one row of RMU / EMU per item)
    const size_t surf_per_ms = sizeof(double) * ((size_t)nn * (nn + 1) + nn + numu1 * (nn + 1) + numu1 + 4);
    const size_t per_slot = (per_ms + (brdf_item ? surf_per_ms : 0)) * nmode + sizeof(double) * sv_stride + sizeof(int32_t) * svi_stride;
    size_t budget = (size_t)64 << 30;   // of 288 GB: fewer, larger passes (launch tails cost ~7 % at 16k items)
    // A caller that says how large its batches are (a run of the Fortran host: one batch, one process) gets at most 9 GB
    // unless they are huge: the runtime hands out up to ~9 GB in 0.3 ms, but 16 GB cost 1.2 s, 47 GB 1.9 s and 64 GB
    // 3.3 s the first time in a process (tools/microbench/malloc_time.hip, profiles/r05_malloc_time.txt) -- more than the
    // whole run of anything below a few million solves, whose extra passes cost milliseconds.
    if (cfg->max_batch > 0 && cfg->max_batch < 2000000) budget = (size_t)9 << 30;
    if (const char *s = getenv("SBD_WORKSPACE_MB")) budget = (size_t)atoll(s) << 20;
    // Two workspaces of `chunk` items each: consecutive passes of a batch alternate between them on two
    // streams, so that the layer kernel of one pass (latency-bound arithmetic) runs beside the band LU /
    // back-substitution of the other (HBM streaming) instead of after it.
    // (measured on the 131 254-solve sweep, tools/bench_chunks.py, twice: 6 passes of 21 876 items 14.95 ms,
    //  4 passes of 32 814 15.3 ms, 10 passes of 13 126 15.05 ms -- more boundaries where a tail of one stream
    //  overlaps the other's kernels, until launch tails take it back)
    int chunk = 32768;
    if (const char *s = getenv("SBD_CHUNK")) chunk = atoi(s);
    if (cfg->max_batch > 0 && cfg->max_batch < chunk) chunk = cfg->max_batch;
    while (chunk > 1 && (size_t)2 * chunk * per_slot > budget) chunk /= 2;
    if (chunk < 1) chunk = 1;
    size_t flag_bytes = 0;
    for (;;) {   // a GPU that cannot spare the budget right now gets smaller passes instead of an error
        flag_bytes = sizeof(int32_t) * ((size_t)chunk * nmode * L + 4);   // count + entries
        const size_t rc_bytes = sizeof(int32_t) * (2 * (size_t)chunk * nmode + 4) + 512;   // rclist (count + entries) and rcflag
        e->ws_bytes = 2 * (((size_t)chunk * per_slot + flag_bytes + rc_bytes + 8192 + 255) & ~(size_t)255);
        const hipError_t me_ = hipMalloc(&e->d_ws, e->ws_bytes);
        if (me_ == hipSuccess) break;
        e->d_ws = nullptr;
        (void)hipGetLastError();
        if (me_ != hipErrorOutOfMemory || chunk <= 64) CREATE_TRY(me_);
        chunk /= 2;
    }
    e->chunk = chunk;
    t_mark("workspace");
    // the list-walking layer kernel tells the host how long it found the list (a word of pinned host memory per workspace,
    // written by the kernel itself: no copy command in the stream)
    if (hipHostMalloc((void **)&e->h_hint, 4 * sizeof(int32_t), hipHostMallocDefault) == hipSuccess) { e->h_hint[0] = -1; e->h_hint[1] = -1; e->h_hint[2] = -1; e->h_hint[3] = -1; }
    else { (void)hipGetLastError(); e->h_hint = nullptr; }
    {
        char *p = e->d_ws;
        auto take = [&](size_t bytes) { char *r = p; p += (bytes + 255) & ~(size_t)255; return r; };
        sbd::Params &P = e->P;
        const size_t nms = (size_t)chunk * nmode;
        e->d_eigflag = (int32_t *)take(flag_bytes);
        P.eiglist = e->d_eigflag;
        P.eighint = e->h_hint;
        P.rclist = (int32_t *)take(sizeof(int32_t) * (nms + 4));
        P.rcflag = (int32_t *)take(sizeof(int32_t) * nms);
        P.sv = (double *)take(sizeof(double) * (size_t)chunk * sv_stride);
        P.svi = (int32_t *)take(sizeof(int32_t) * (size_t)chunk * svi_stride);
        P.gc = (double *)take(sizeof(double) * nms * L * n * n);
        if (band4) {
            P.ga = P.gb = nullptr;
            P.bcb = (double *)take(sizeof(double) * nms * n * n);
            P.gcc = (double *)take(sizeof(double) * nms * L * 2 * nn * nn);
        } else if (band1) {      // (round 4: band1_kernel builds its interface rows from GC's quarters like band4_kernel)
            P.ga = P.gb = nullptr;
            P.bcb = (double *)take(sizeof(double) * nms * 2 * n * n);
            P.gcc = (double *)take(sizeof(double) * nms * L * 2 * nn * nn);
        } else {
            P.ga = (double *)take(sizeof(double) * nms * L * n * n);
            P.gb = (double *)take(sizeof(double) * nms * L * n * n);
            P.bcb = band1 ? (double *)take(sizeof(double) * nms * 2 * n * n) : nullptr;
            P.gcc = nullptr;
        }
        P.kk = (double *)take(sizeof(double) * nms * L * n);
        P.ek = (double *)take(sizeof(double) * nms * L * nn);
        P.zz = (double *)take(sizeof(double) * nms * L * n);
        P.zp0 = (double *)take(sizeof(double) * nms * L * n);
        P.zp1 = (double *)take(sizeof(double) * nms * L * n);
        P.ll = (double *)take(sizeof(double) * nms * L * n);
        P.yv = (double *)take(sizeof(double) * nms * L * n);
        P.ufac = fused ? nullptr : (double *)take(sizeof(double) * nms * L * n * (2 * n));   // sbd::u_width(n)
        if (rad_user) {
            P.gu = (double *)take(sizeof(double) * nms * L * n * numu);
            P.zb = (double *)take(sizeof(double) * nms * L * numu);
            P.z0u = (double *)take(sizeof(double) * nms * L * numu);
            P.z1u = (double *)take(sizeof(double) * nms * L * numu);
        }
        if (rad) P.uum = (double *)take(sizeof(double) * nms * e->nlev * numu);
        P.ibcnd = cfg->ibcnd == 2 ? 1 : 0;
        P.slot_base = 0;
        P.ibdrf = brdf ? cfg->ibdrf : 0;
        P.brdf_shared = (brdf && !brdf_item) ? 1 : 0;
        for (int k = 0; k < 8; ++k) P.bpar[k] = brdf ? cfg->bpar[k] : 0.0;
        P.bitem = nullptr;
        P.bdr = P.bem = P.rmu = P.emu = nullptr;
        if (brdf_item) {
            P.bdr = (double *)take(sizeof(double) * nms * nn * (nn + 1));
            P.bem = (double *)take(sizeof(double) * nms * nn);
            P.rmu = (double *)take(sizeof(double) * nms * (numu > 0 ? numu : 1) * (nn + 1));
            P.emu = (double *)take(sizeof(double) * nms * (numu > 0 ? numu : 1));
        }
        if ((size_t)(p - e->d_ws) > e->ws_bytes / 2) {
            sbd_engine_destroy(e);
            return fail(SBD_E_NOMEM, "workspace carve overflow");
        }
        P.L = L; P.n = n; P.nn = nn; P.nmom = cfg->nmom; P.numu = numu; P.nphi = nphi;
        P.nlev = e->nlev; P.nmode = nmode;
        P.onlyfl = cfg->onlyfl ? 1 : 0; P.usrang = cfg->usrang ? 1 : 0;
        P.all_levels = cfg->nlevel_out > 0 ? 0 : 1;
        P.sv_stride = sv_stride; P.svi_stride = svi_stride;
        P.cw = cw; P.ncd = ncd;
        P.umu0 = cfg->umu0; P.rumu0 = (cfg->umu0 != 0.0) ? 1.0 / cfg->umu0 : 0.0; P.fisot = cfg->fisot; P.btemp = cfg->btemp; P.ttemp = cfg->ttemp; P.temis = cfg->temis;
        P.pi = ref_pi();
        P.dither = 100.0 * 2.220446049250313e-16;   // disort.f:442-448
        P.t = e->tab;
        P.force_fallback = 0;
        if (const char *s = getenv("SBD_FORCE_EIG_FALLBACK")) P.force_fallback = atoi(s) != 0;
    }
    // the carve above rounds every array up to 256 B: re-check against the allocation
    // (ws_bytes has 4 KB slack per array count << 16)
    // ---- launch geometry ----
    int G = 4;
    while (G < n) G <<= 1;
    e->G = G;
    const sbd::LayerLds ll(n, nn);
    e->layer_lds = (int)sizeof(double) * ll.total * (64 / G);
    e->band_reg = false;                          // (round 1's register-window LU is gone: LDS window, sbd_band.hpp)
    e->band4 = band4;
    e->band1 = band1;
    e->band_rows = band_rows;
    e->fused = fused;
    e->corint = rad && cfg->corint != 0;
    e->quad = quad;
    e->P.ublock = (band4 || band_rows) ? 1 : 0;
    e->P.gconly = (band4 || band1) ? 1 : 0;
    const sbd::BandLds bl(n, nn, e->band_reg);
    e->band_lds = (int)sizeof(double) * bl.total;
    const sbd::SolveLds sl(n, nn, L);
    e->solve_lds = (int)sizeof(double) * sl.total;
    e->usr_lds = (int)(sizeof(double) * (nn + 2) + sizeof(int) * ((size_t)e->nlev * (numu > 0 ? numu : 1) + 2));   // dfu + the active-item list
    if (e->layer_lds > 160 * 1024 || e->band_lds > 160 * 1024 || e->solve_lds > 160 * 1024) {
        sbd_engine_destroy(e);
        return fail(SBD_E_UNSUPPORTED, "LDS footprint exceeds 160 KiB for this NSTR/NLYR");
    }
    t_mark("carve");
    CREATE_TRY(sbd::prepare_layer_v1(G, e->layer_lds));
    t_mark("prepare-layer-v1");
    CREATE_TRY(sbd::prepare_band_lds(nn, e->band_lds));
    CREATE_TRY(sbd::prepare_backsolve(nn, e->solve_lds));
    {
        const sbd::Layer2Lds l2(n, nn, rad_user, rad_user ? numu : 0);
        e->G2 = sbd::l2_group(nn);
        e->layer2_lds = (int)sizeof(double) * (l2.shared_total + l2.group_total * (64 / e->G2));
        if (const char *s = getenv("SBD_LAYER_V1")) e->use_layer2 = atoi(s) == 0;
        if (e->layer2_lds > 160 * 1024) e->use_layer2 = false;
        // the list is emptied by setup_kernel, filled by layer_kernel2 and walked by the QR kernel
        CREATE_TRY(hipMemset(e->d_eigflag, 0, flag_bytes));
        if (e->use_layer2) CREATE_TRY(sbd::prepare_layer2(nn, rad_user, e->layer2_lds));
    }
    t_mark("prepare-kernels");
    for (auto &x : e->ev) CREATE_TRY(hipEventCreate(&x));
    {   // band_rcond_kernel's scratch (both workspaces) and its report
        e->rb_stride = sbd::band_rcond_scratch_doubles(n, L);
        // one wave per listed system, serial in its band part: the concurrency is the number of blocks -- as many as 1 GB of
        // scratch per workspace holds, 8 .. 512 (1 MB per block at NSTR 16 x 33 layers, 6.5 MB at NSTR 40 x 65)
        e->rc_blocks = (int)std::max<size_t>(8, std::min<size_t>(512, ((size_t)1 << 30) / (sizeof(double) * e->rb_stride)));
        if (cfg->max_batch > 0) e->rc_blocks = (int)std::min<size_t>((size_t)e->rc_blocks, std::max<size_t>(8, (size_t)cfg->max_batch * e->nmode));
        CREATE_TRY(hipMalloc(&e->d_rb, sizeof(double) * 2 * (size_t)e->rc_blocks * e->rb_stride));
        CREATE_TRY(hipMalloc(&e->d_rcdbg, sizeof(double) * 2 * (size_t)e->chunk * e->nmode));
        CREATE_TRY(hipMemset(e->d_rcdbg, 0xFF, sizeof(double) * 2 * (size_t)e->chunk * e->nmode));
        CREATE_TRY(hipMemset(e->P.rclist, 0, sizeof(int32_t) * 4));
    }
    {   // second workspace (every field of P is final here): each workspace pointer moved by half the allocation
        e->P2 = e->P;
        const size_t half = e->ws_bytes / 2;
        auto mv = [&](auto *&ptr) { if (ptr) ptr = (std::remove_reference_t<decltype(ptr)>)((char *)ptr + half); };
        sbd::Params &Q = e->P2;
        mv(Q.eiglist); mv(Q.rclist); mv(Q.rcflag); mv(Q.sv); mv(Q.svi); mv(Q.gc); mv(Q.ga); mv(Q.gb); mv(Q.bcb); mv(Q.gcc); mv(Q.kk); mv(Q.ek); mv(Q.zz); mv(Q.zp0);
        mv(Q.zp1); mv(Q.ll); mv(Q.yv); mv(Q.ufac); mv(Q.gu); mv(Q.zb); mv(Q.z0u); mv(Q.z1u); mv(Q.uum);
        if (brdf_item) { mv(Q.bdr); mv(Q.bem); mv(Q.rmu); mv(Q.emu); }
        CREATE_TRY(hipMemset(Q.eiglist, 0, sizeof(int32_t) * ((size_t)e->chunk * e->nmode * e->L + 4)));
        CREATE_TRY(hipMemset(Q.rclist, 0, sizeof(int32_t) * 4));
        Q.eighint = e->h_hint ? e->h_hint + 1 : nullptr;
    }
    if (brdf && !brdf_item) {
        // Hapke / Ross-Li do not depend on the wavelength: SURFAC's tables and CHEKIN's test of the model are made
        // once, here, for every azimuth mode (the beam column included: it only ever meets FBEAM > 0)
        const size_t cnt = (size_t)nmode * ((size_t)nn * (nn + 1) + nn + (size_t)(numu > 0 ? numu : 1) * (nn + 2));
        CREATE_TRY(hipMalloc(&e->d_surf, sizeof(double) * cnt));
        CREATE_TRY(hipMemset(e->d_surf, 0, sizeof(double) * cnt));
        CREATE_TRY(hipMalloc(&e->d_surf_flag, sizeof(int32_t)));
        CREATE_TRY(hipMemset(e->d_surf_flag, 0, sizeof(int32_t)));
        double *q = e->d_surf;
        for (sbd::Params *PP : {&e->P, &e->P2}) {
            PP->bdr = q;
            PP->bem = q + (size_t)nmode * nn * (nn + 1);
            PP->rmu = PP->bem + (size_t)nmode * nn;
            PP->emu = PP->rmu + (size_t)nmode * (numu > 0 ? numu : 1) * (nn + 1);
        }
        sbd::Params Ps = e->P;
        Ps.nslot = 1;
        hipLaunchKernelGGL(sbd::surfac_kernel, dim3((unsigned)nmode), dim3(256), sizeof(double) * sbd::surf_lds_doubles(nn, numu),
                           e->stream, Ps, e->d_surf_flag);
        int32_t flag = 0;
        CREATE_TRY(hipMemcpyAsync(&flag, e->d_surf_flag, sizeof(flag), hipMemcpyDeviceToHost, e->stream));
        CREATE_TRY(hipStreamSynchronize(e->stream));
        e->brdf_bad = flag != 0;
    }
#undef CREATE_TRY

    // beam angle == quadrature angle (disort.f:2643-2650): whole-run property when a beam
    // is present; reported here so the host can pick NSTR-2 / NSTR+2 as drt.f:536-555 does.
    t_mark("end");
    if (t_on) fprintf(stderr, "sbdart_amd: engine create on device %d (ms since entry):%s\n", cfg->device, t_log.c_str());
    *out = e;
    if (cfg->umu0 > 0.0)
        for (int iq = 0; iq < nn; ++iq)
            if (fabs(cfg->umu0 - e->h_cmu[iq]) / cfg->umu0 < (double)1.0e-4f)
                return fail(SBD_E_RETRY_NSTR, "beam angle = computational angle; change NSTR");
    return SBD_OK;
}

int32_t sbd_engine_nlevel(const sbd_engine *e) { return e ? e->nlev : 0; }
size_t sbd_engine_workspace_bytes(const sbd_engine *e) { return e ? e->ws_bytes : 0; }
int32_t sbd_engine_chunk(const sbd_engine *e) { return e ? e->chunk : 0; }
void *sbd_engine_stream(sbd_engine *e) { return e ? (void *)e->stream : nullptr; }
void sbd_engine_enable_timing(sbd_engine *e, int on) { if (e) { e->timing = on == 1; e->timing_inplace = on == 2; if (!on) e->have_times = false; } }

int sbd_engine_quadrature(const sbd_engine *e, double *cmu, double *cwt)
{
    if (!e || !cmu || !cwt) return SBD_E_INVALID;
    for (int i = 0; i < e->nn; ++i) { cmu[i] = e->h_cmu[i]; cwt[i] = e->h_cwt[i]; }
    return SBD_OK;
}

// DREF(WVNMLO, WVNMHI, MU) (disort.f:5178-5284) on the host: the flux albedo of a bidirectional surface for incidence
// cosine mu, with the very model functions the device integrates (sbd_surface.hpp) and the same 50-point rule.  The
// reference's driver asks for it once per wavelength when ISALB is -7, -8 or -9 -- a LAMBERTIAN surface whose albedo
// is the model's flux albedo at the solar zenith angle (drt.f:478-484) -- so it is a host program's call, no GPU.
int sbd_surface_flux_albedo(int32_t ibdrf, const double *bpar, const double *bitem, double mu, double *albedo)
{
    if (!bpar || !albedo || ibdrf < 1 || ibdrf > 3) return fail(SBD_E_INVALID, "sbd_surface_flux_albedo: model 1..3, bpar, albedo");
    if (ibdrf == 1 && !bitem) return fail(SBD_E_INVALID, "sbd_surface_flux_albedo: the ocean needs bitem");
    if (!(fabs(mu) <= 1.0)) return fail(SBD_E_INVALID, "DREF--input argument error(s)");   // disort.f:5262: ABS(MU) > 1 only --
    // a cosine below zero (the sun under the horizon: drt.f hands cos(SZA) over as it is) goes through the model functions
    constexpr int NG = sbd::kSurfGauss;
    static double gmu[NG], gwt[NG];
    static std::once_flag once;
    std::call_once(once, [] {
        gauss01(NG / 2, gmu, gwt);
        for (int k = 0; k < NG / 2; ++k) { gmu[k + NG / 2] = -gmu[k]; gwt[k + NG / 2] = gwt[k]; }
    });
    sbd::BrdfModel M;
    M.ibdrf = ibdrf;
    for (int k = 0; k < 8; ++k) M.bp[k] = bpar[k];
    M.nr = bitem ? bitem[0] : 0.0; M.ni = bitem ? bitem[1] : 0.0; M.rsw = bitem ? bitem[2] : 0.0;
    const double pi = ref_pi();
    double d = 0.0;
    for (int jg = 0; jg < NG; ++jg) {
        double sum = 0.0;
        for (int k = 0; k < NG / 2; ++k) sum = sum + gwt[k] * gmu[k] * sbd::surf_bdref(M, gmu[k], mu, pi * gmu[jg]);
        d = d + gwt[jg] * sum;
    }
    *albedo = d;
    return SBD_OK;
}

int64_t sbd_engine_last_fallback_layers(sbd_engine *e) { return (e && e->have_times) ? e->fallback_layers : -1; }

double sbd_engine_last_ms(sbd_engine *e, int phase)
{
    if (e && e->ip_pending) {                 // in-place events of the last solve: read them now that they are asked for
        e->ip_pending = false;
        (void)hipSetDevice(e->cfg.device);
        bool ok = true;
        for (float &m : e->ms_phase) m = 0.f;
        for (int ip = 0; ip < e->ip_npass && ok; ++ip) {
            ok = hipEventSynchronize(e->ev_ip[(size_t)ip * (sbd_engine::kPhases + 1) + sbd_engine::kPhases]) == hipSuccess;
            for (int ph = 0; ph < sbd_engine::kPhases && ok; ++ph) {
                float ms = 0.f;
                ok = hipEventElapsedTime(&ms, e->ev_ip[(size_t)ip * (sbd_engine::kPhases + 1) + ph], e->ev_ip[(size_t)ip * (sbd_engine::kPhases + 1) + ph + 1]) == hipSuccess;
                e->ms_phase[ph] += ms;
            }
        }
        if (!ok) (void)hipGetLastError();
        e->fallback_layers = -1;
        e->have_times = ok;
    }
    if (!e || !e->have_times) return -1.0;
    if (phase < 0) {
        double t = 0.0;
        for (float m : e->ms_phase) t += (double)m;
        return t;
    }
    if (phase >= sbd_engine::kPhases) return -1.0;
    return (double)e->ms_phase[phase];
}

// tests only: the four-per-wave band LU records the register index of every pivot row (NSTR <= 16, stored-factor path)
int sbd_engine_debug_pivots(sbd_engine *e, int on)
{
    if (!e) return SBD_E_INVALID;
    if (!e->band4 || e->fused) return fail(SBD_E_INVALID, "pivot record: NSTR <= 16 and the stored-factor path only");
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (on && !e->d_pivdbg) {
        const size_t cnt = (size_t)2 * e->chunk * e->nmode * e->L * e->n;
        HIP_TRY(hipMalloc(&e->d_pivdbg, sizeof(int32_t) * cnt));
        HIP_TRY(hipMemset(e->d_pivdbg, 0xff, sizeof(int32_t) * cnt));
    } else if (!on && e->d_pivdbg) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(e->d_pivdbg);
        e->d_pivdbg = nullptr;
    }
    return SBD_OK;
}

long long sbd_engine_debug_copy(sbd_engine *e, int which, void *host_buf, size_t nbytes)
{
    if (!e || !host_buf) return SBD_E_INVALID;
    const size_t nms = (size_t)e->chunk * e->nmode, L = e->L, n = e->n, nn = e->nn;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (which) {
    case 0: src = e->P.gc; bytes = 8 * nms * L * n * n; break;
    case 1: src = e->P.kk; bytes = 8 * nms * L * n; break;
    case 2: src = e->P.ek; bytes = 8 * nms * L * nn; break;
    case 3: src = e->P.zz; bytes = 8 * nms * L * n; break;
    case 4: src = e->P.zp0; bytes = 8 * nms * L * n; break;
    case 5: src = e->P.zp1; bytes = 8 * nms * L * n; break;
    case 6: src = e->P.ll; bytes = 8 * nms * L * n; break;
    case 7: src = e->P.sv; bytes = 8 * (size_t)e->chunk * e->P.sv_stride; break;
    case 8: src = e->P.svi; bytes = 4 * (size_t)e->chunk * e->P.svi_stride; break;
    case 9: src = e->P.gu; bytes = 8 * nms * L * n * e->P.numu; break;
    case 10: src = e->P.zb; bytes = 8 * nms * L * e->P.numu; break;
    case 11: src = e->P.z0u; bytes = 8 * nms * L * e->P.numu; break;
    case 12: src = e->P.z1u; bytes = 8 * nms * L * e->P.numu; break;
    case 13: src = e->P.eiglist; bytes = 4 * 64; break;     // count + first entries of the fallback list (first workspace)
    case 14: src = e->P2.eiglist; bytes = 4 * 64; break;    // ... second workspace
    case 15: src = e->d_pivdbg; bytes = e->d_pivdbg ? 4 * nms * L * n : 0; break;   // pivot register indices (first workspace)
    case 16: src = e->d_rcdbg; bytes = 8 * nms; break;        // band_rcond_kernel's estimates [item x mode] (first workspace; NaN: not served)
    case 17: src = e->P.rclist; bytes = 4 * (nms + 1); break; // its list: count, then the ms indices
    default: return SBD_E_INVALID;
    }
    if (bytes > nbytes) bytes = nbytes;
    if (hipSetDevice(e->cfg.device) != hipSuccess) return SBD_E_HIP;
    if (hipDeviceSynchronize() != hipSuccess) return SBD_E_HIP;
    if (hipMemcpy(host_buf, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return SBD_E_HIP;
    return (long long)bytes;
}

// Host arrays behind a device-pointer solve (sbd_engine_solve_host, fleets): every pass copies its own
// slice in before its kernels and its outputs back after them, on the pass's stream -- the H2D of one
// pass then runs beside the kernels of the other instead of ahead of everything.
struct MixStage {               // device staging of a compact batch (sbd_mix_in): its items, and the point blocks pbase .. they refer to
    int32_t *point_of, *kterm;
    double *dtaug, *lay, *lo, *hi, *fb, *al;
    uint8_t *pl;
    int32_t pbase;              // first spectral point the call's items refer to: staged block q holds point pbase + q
    int32_t lay_base;           // ... and the point whose layer block is block 0 of `lay` (pbase, or the gas call's first point)
    bool lay_resident;          // the layer blocks are already on the device (sbd_fleet_gas_terms kept them)
};
struct HostSide {
    const sbd_batch_in *in;     // host inputs (NULL members never occur: checked by the callers)
    const sbd_batch_out *out;   // host outputs; flux / uu / status may each be NULL (not wanted)
    bool rows_sorted;           // in->pmom_row is non-decreasing: every pass needs one contiguous range of moment blocks
    const sbd_mix_in *mix = nullptr;   // the batch comes in compact form: `in` is unused, every pass assembles its slice
    MixStage ms = {};
};
static int solve_device_impl(sbd_engine *e, const sbd_batch_in *in, const sbd_batch_out *out, void *hip_stream, const HostSide *hs);

// A batch larger than the workspace goes through in EQUAL passes (no short tail pass whose kernels cost their full launch
// latency for a handful of items); consecutive passes alternate between the two workspaces and run on two streams: an even
// number, the kernels of pass i+1 beside those of pass i.
int32_t sbd_engine_pass_count(const sbd_engine *e, int32_t nwork)
{
    if (!e || nwork <= 0) return 0;
    int npass = (nwork + e->chunk - 1) / e->chunk;
    if (npass == 1 && nwork >= 16384) npass = 2;
    // NSTR 18-32, flux: the layer kernel (VALU) and band1_kernel (LDS pipe, dependent chains) of different passes fill each
    // other's gaps -- six passes of >= 1 365 items instead of one: 446 k -> 461 k points/s on 16 258 solves of NSTR 32 x 50
    // layers (tools/chunk_probe_cfgD.py, round 4)
    if (e->band1 && e->cfg.onlyfl && nwork >= 8192 && npass < 6) npass = 6;
    if (npass > 1 && (npass & 1)) ++npass;
    return npass;
}

// IBCND = 1: expand the batch into its top-lit / bottom-lit internal items, run the general pipeline on them, close with
// ALBTRN's formulas (device pointers in and out, everything on the caller's stream)
static int ibcnd_solve_device(sbd_engine *e, const sbd_batch_in *in, const sbd_batch_out *out, void *hip_stream)
{
    if (!in || !out) return fail(SBD_E_INVALID, "null argument");
    if (in->nwork < 0) return fail(SBD_E_INVALID, "nwork < 0");
    if (in->nwork == 0) return SBD_OK;
    if (!in->dtauc || !in->ssalb || !in->pmom || !in->wvnmlo || !in->wvnmhi || !in->albedo) return fail(SBD_E_INVALID, "null input array");
    if (!out->albtrn || !out->status) return fail(SBD_E_INVALID, "IBCND = 1: albtrn / status is NULL");
    if (in->pmom_row) return fail(SBD_E_INVALID, "IBCND = 1: one block of moments per item (pmom_row must be NULL)");
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    const size_t W = in->nwork, W2 = 2 * W;
    const int L = e->L, npm = L * (e->cfg.nmom + 1), numu2 = e->P.numu;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_lay = up(8 * W2 * L), b_pm = up(8 * W2 * npm), b_w = up(8 * W2), b_fl = up(8 * W2 * SBD_NFLUX * 2),
                 b_uu = up(8 * W2 * 2 * numu2), b_st = up(4 * W2), b_pl = up(W2);
    const size_t total = 2 * b_lay + b_pm + 4 * b_w + b_pl + b_fl + b_uu + b_st;
    if (total > e->ib_bytes) {
        if (e->d_ib) (void)hipFree(e->d_ib);
        e->d_ib = nullptr; e->ib_bytes = 0;
        if (hipMalloc(&e->d_ib, total) != hipSuccess) { (void)hipGetLastError(); return fail(SBD_E_NOMEM, "hipMalloc(IBCND staging)"); }
        e->ib_bytes = total;
    }
    char *p = e->d_ib;
    auto take = [&](size_t bytes) { char *r = p; p += bytes; return r; };
    double *dt2 = (double *)take(b_lay), *ss2 = (double *)take(b_lay), *pm2 = (double *)take(b_pm);
    double *lo2 = (double *)take(b_w), *hi2 = (double *)take(b_w), *fb2 = (double *)take(b_w), *al2 = (double *)take(b_w);
    uint8_t *pl2 = (uint8_t *)take(b_pl);
    double *flux2 = (double *)take(b_fl), *uu2 = (double *)take(b_uu);
    int32_t *st2 = (int32_t *)take(b_st);
    const size_t tot = W2 * npm;
    const unsigned grid = (unsigned)((tot + 255) / 256 < 65535 ? (tot + 255) / 256 : 65535);
    hipLaunchKernelGGL(ibcnd_expand_kernel, dim3(grid), dim3(256), 0, st, (int)W, L, npm, in->dtauc, in->ssalb, in->pmom,
                       in->wvnmlo, in->wvnmhi, dt2, ss2, pm2, lo2, hi2, fb2, al2, pl2);
    const sbd_batch_in in2 = {(int32_t)W2, dt2, ss2, pm2, lo2, hi2, fb2, al2, pl2, nullptr};
    const sbd_batch_out out2 = {flux2, uu2, st2, nullptr};
    const int rc = solve_device_impl(e, &in2, &out2, st, nullptr);
    if (rc != SBD_OK) return rc;
    hipLaunchKernelGGL(ibcnd_combine_kernel, dim3((unsigned)W), dim3(64), 0, st, (int)W, e->ib_nout, numu2, e->P.pi, in->albedo,
                       (const double *)flux2, (const double *)uu2, (const int32_t *)st2, out->albtrn, out->flux, out->status);
    HIP_TRY(hipGetLastError());
    return SBD_OK;
}

int sbd_engine_solve_device(sbd_engine *e, const sbd_batch_in *in, const sbd_batch_out *out, void *hip_stream)
{
    if (e && e->ibcnd) return ibcnd_solve_device(e, in, out, hip_stream);
    return solve_device_impl(e, in, out, hip_stream, nullptr);
}

static int solve_device_impl(sbd_engine *e, const sbd_batch_in *in, const sbd_batch_out *out, void *hip_stream, const HostSide *hs)
{
    if (!e || !in || !out) return fail(SBD_E_INVALID, "null argument");
    if (in->nwork < 0) return fail(SBD_E_INVALID, "nwork < 0");
    if (in->nwork == 0) return SBD_OK;
    if (!in->dtauc || !in->ssalb || !in->pmom || !in->wvnmlo || !in->wvnmhi || !in->fbeam || !in->albedo || !in->plank)
        return fail(SBD_E_INVALID, "null input array");
    if (e->P.ibdrf == 1 && !in->bitem) return fail(SBD_E_INVALID, "ocean surface (ibdrf = 1): bitem is NULL");
    if (!out->flux || !out->status) return fail(SBD_E_INVALID, "null output array");
    const bool rad = !e->cfg.onlyfl;
    if (rad && !out->uu) return fail(SBD_E_INVALID, "uu is NULL in radiance mode");
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    const int L = e->L, n = e->n, nmode = e->nmode, nlev = e->nlev;
    const bool timing = e->timing;
    const bool tip = e->timing_inplace && !timing;
    const bool dbg = getenv("SBD_DEBUG_SYNC") != nullptr;
#define SBD_DBG(tag) do { if (dbg) { hipError_t de_ = hipStreamSynchronize(st); fprintf(stderr, "[sbd] %s: %s (eigflag=%p partial=%p ws=%p..%p)\n", tag, hipGetErrorString(de_), (void*)e->d_eigflag, (void*)e->d_partial, (void*)e->d_ws, (void*)(e->d_ws + e->ws_bytes)); } } while (0)
    float acc_ms[sbd_engine::kPhases] = {};
    int64_t nfallback = 0;
    // a batch larger than the workspace goes through in EQUAL passes (no short tail pass whose
    // kernels cost their full launch latency for a handful of items)
    // With more than one pass, consecutive passes alternate between the two workspaces and run on two
    // streams (the caller's and the engine's auxiliary one): an even number of equal passes, the
    // kernels of pass i+1 beside those of pass i.  Per-kernel timing (enable_timing) keeps one stream.
    int npass = sbd_engine_pass_count(e, in->nwork);
    const bool fork = !timing && !dbg && npass > 1;
    const int per_pass = (in->nwork + npass - 1) / npass;
    // pass ip = items [pw0[ip], pw0[ip + 1]).  Inputs resident on the device: the equal passes above.  Inputs coming
    // from the host (hs): nothing can start before the first pass's inputs have crossed PCIe -- 1.15 ms for a sixth of
    // the bench's batch (rocprofv3 time line) -- and copying a pass takes 0.77 of computing it, so the passes start
    // at a quarter of the size and grow by 1.3 x: each copy still lands before the pass ahead of it has finished.
    std::vector<int> pw0;
    pw0.push_back(0);
    // (Not below 8192 items: a smaller pass leaves the band kernel less than one wave per SIMD slot.  Only with the
    //  moments shared per spectral point: with per-item moments the copies are the longer leg and more of them cost.)
    if (hs && fork && in->nwork >= 32768 && (hs->mix || hs->in->pmom_row)) {
        double sz = (0.25 * per_pass > 8192.0) ? 0.25 * per_pass : 8192.0, grow = 1.3;
        if (hs->mix) { sz = 0.5 * per_pass; grow = 2.0; }     // (compact form: a ninth of the bytes -- one half-size pass ahead)
        if (const char *s1 = getenv("SBD_HOST_FIRST_PASS")) sz = atof(s1);      // (developer knobs: tools/host_passes_probe.py)
        if (const char *s2 = getenv("SBD_HOST_PASS_GROWTH")) grow = atof(s2);
        while (sz < 0.95 * per_pass && pw0.back() + (int)sz < in->nwork) {
            pw0.push_back(pw0.back() + (int)sz);
            sz *= grow;
        }
        const int rest = in->nwork - pw0.back();
        const int m = (rest + per_pass - 1) / per_pass;
        for (int k = 1; k <= m; ++k) pw0.push_back(in->nwork - rest + (int)((long long)rest * k / m));
    } else {
        for (int w = per_pass; w < in->nwork; w += per_pass) pw0.push_back(w);
        pw0.push_back(in->nwork);
    }
    const int npass_run = (int)pw0.size() - 1;
    if (fork) {
        HIP_TRY(hipEventRecord(e->ev_fork, st));                 // what the caller queued before this call ...
        HIP_TRY(hipStreamWaitEvent(e->aux, e->ev_fork, 0));      // ... is also ahead of the auxiliary stream
    }
    hipStream_t st_main = st;
    // inputs of pass ip, host -> staging, on the copy stream (see sbd_engine::copy).  Pass 0 goes first, pass
    // i+1 right after the kernels of pass i have been queued: with pinned host arrays nothing here waits; with
    // pageable ones (the runtime stages those and returns when the copy is done) the calling thread copies
    // while the GPU computes the pass before
    auto copy_pass = [&](const int ip) -> int {
        if (!hs || ip >= npass_run) return SBD_OK;
        const int w0 = pw0[ip];
        const size_t npm = (size_t)L * (e->cfg.nmom + 1);
        const int ns = pw0[ip + 1] - w0;
        hipStream_t cs = e->copy;
        if (hs->mix) {
            // compact form: this pass's items' gas depths and the scatterers of the spectral points they belong to cross
            // PCIe, then DISORT's arguments are formed where they are needed (assemble_kernel, on the copy stream:
            // ordered behind the copies, ahead of the event the pass's kernels wait for)
            const sbd_mix_in *m = hs->mix;
            const MixStage &d = hs->ms;
            const int p0 = m->point_of[w0], p1 = m->point_of[w0 + ns - 1];
            const bool done = ip > 0 && m->point_of[w0 - 1] == p0;        // (its moments exist: the pass before made them)
            const int q0 = done ? p0 + 1 : p0, nq = p1 - q0 + 1;           // (global point indices)
            const size_t blk = (size_t)(4 + 3 * m->nterm) * L;
            HIP_TRY(hipMemcpyAsync(d.point_of + w0, m->point_of + w0, sizeof(int32_t) * ns, hipMemcpyHostToDevice, cs));
            if (m->dtaug) HIP_TRY(hipMemcpyAsync(d.dtaug + (size_t)w0 * L, m->dtaug + (size_t)w0 * L, sizeof(double) * ns * L, hipMemcpyHostToDevice, cs));
            else HIP_TRY(hipMemcpyAsync(d.kterm + w0, m->kterm + w0, sizeof(int32_t) * ns, hipMemcpyHostToDevice, cs));
            if (nq > 0) {
                const int s0 = q0 - d.pbase;                               // (staged block index)
                if (!d.lay_resident) HIP_TRY(hipMemcpyAsync(d.lay + (size_t)s0 * blk, m->lay + (size_t)q0 * blk, sizeof(double) * nq * blk, hipMemcpyHostToDevice, cs));
                const std::pair<double *, const double *> sc[4] = {{d.lo, m->wvnmlo}, {d.hi, m->wvnmhi}, {d.fb, m->fbeam}, {d.al, m->albedo}};
                for (const auto &a : sc)
                    HIP_TRY(hipMemcpyAsync(a.first + s0, a.second + q0, sizeof(double) * nq, hipMemcpyHostToDevice, cs));
                HIP_TRY(hipMemcpyAsync(d.pl + s0, m->plank + q0, (size_t)nq, hipMemcpyHostToDevice, cs));
            }
            MixFamilies fam;
            for (int t = 0; t < SBD_MIX_MAX_TERMS; ++t) fam.f[t] = m->family[t];
            hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)ns), dim3(64), 0, cs, w0, ns, L, e->cfg.nmom, done ? 1 : 0,
                               (int)d.pbase, (int)d.lay_base, (int)m->nterm, fam, (const int32_t *)d.kterm, (const double *)(m->dtaug ? nullptr : e->d_gas_slots), (int)e->gas_p0,
                               (const int32_t *)d.point_of, (const double *)d.dtaug, (const double *)d.lay, (const double *)d.lo, (const double *)d.hi,
                               (const double *)d.fb, (const double *)d.al, (const uint8_t *)d.pl,
                               (double *)in->dtauc, (double *)in->ssalb, (double *)in->pmom, (int32_t *)in->pmom_row,
                               (double *)in->wvnmlo, (double *)in->wvnmhi, (double *)in->fbeam, (double *)in->albedo, (uint8_t *)in->plank);
            HIP_TRY(hipGetLastError());
            if ((int)e->ev_h2d.size() <= ip) {
                hipEvent_t ev = nullptr;
                HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                e->ev_h2d.push_back(ev);
            }
            HIP_TRY(hipEventRecord(e->ev_h2d[ip], cs));
            return SBD_OK;
        }
        HIP_TRY(hipMemcpyAsync((void *)(in->dtauc + (size_t)w0 * L), hs->in->dtauc + (size_t)w0 * L, sizeof(double) * ns * L, hipMemcpyHostToDevice, cs));
        HIP_TRY(hipMemcpyAsync((void *)(in->ssalb + (size_t)w0 * L), hs->in->ssalb + (size_t)w0 * L, sizeof(double) * ns * L, hipMemcpyHostToDevice, cs));
        if (hs->in->pmom_row) {
            // moments per spectral point: the blocks this pass's items point at (rows non-decreasing: a contiguous range).
            // The first block of a pass may have gone over with the pass before, whose kernels may be READING it now:
            // it is not copied again (same bytes, but a DMA write racing a kernel's reads of the same lines)
            int32_t r0 = hs->rows_sorted ? hs->in->pmom_row[w0] : 0;
            const int32_t r1 = hs->rows_sorted ? hs->in->pmom_row[w0 + ns - 1] : hs->in->npmom - 1;
            if (hs->rows_sorted && ip > 0 && r0 == hs->in->pmom_row[w0 - 1]) ++r0;
            if ((hs->rows_sorted || ip == 0) && r1 >= r0)
                HIP_TRY(hipMemcpyAsync((void *)(in->pmom + (size_t)r0 * npm), hs->in->pmom + (size_t)r0 * npm, sizeof(double) * (size_t)(r1 - r0 + 1) * npm, hipMemcpyHostToDevice, cs));
            HIP_TRY(hipMemcpyAsync((void *)(in->pmom_row + w0), hs->in->pmom_row + w0, sizeof(int32_t) * ns, hipMemcpyHostToDevice, cs));
        } else {
            HIP_TRY(hipMemcpyAsync((void *)(in->pmom + (size_t)w0 * npm), hs->in->pmom + (size_t)w0 * npm, sizeof(double) * ns * npm, hipMemcpyHostToDevice, cs));
        }
        HIP_TRY(hipMemcpyAsync((void *)(in->wvnmlo + w0), hs->in->wvnmlo + w0, sizeof(double) * ns, hipMemcpyHostToDevice, cs));
        HIP_TRY(hipMemcpyAsync((void *)(in->wvnmhi + w0), hs->in->wvnmhi + w0, sizeof(double) * ns, hipMemcpyHostToDevice, cs));
        HIP_TRY(hipMemcpyAsync((void *)(in->fbeam + w0), hs->in->fbeam + w0, sizeof(double) * ns, hipMemcpyHostToDevice, cs));
        HIP_TRY(hipMemcpyAsync((void *)(in->albedo + w0), hs->in->albedo + w0, sizeof(double) * ns, hipMemcpyHostToDevice, cs));
        HIP_TRY(hipMemcpyAsync((void *)(in->plank + w0), hs->in->plank + w0, (size_t)ns, hipMemcpyHostToDevice, cs));
        if (in->bitem) HIP_TRY(hipMemcpyAsync((void *)(in->bitem + (size_t)w0 * 4), hs->in->bitem + (size_t)w0 * 4, sizeof(double) * 4 * ns, hipMemcpyHostToDevice, cs));
        if ((int)e->ev_h2d.size() <= ip) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->ev_h2d.push_back(ev);
        }
        HIP_TRY(hipEventRecord(e->ev_h2d[ip], cs));
        return SBD_OK;
    };
    if (hs) {
        HIP_TRY(hipEventRecord(e->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(e->copy, e->ev_fork, 0));
        const int rc0 = copy_pass(0);
        if (rc0 != SBD_OK) return rc0;
    }
    if (tip) {
        while (e->ev_ip.size() < (size_t)npass_run * (sbd_engine::kPhases + 1)) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            e->ev_ip.push_back(ev);
        }
        e->ip_npass = npass_run;
        e->ip_pending = true;
        e->have_times = false;
    }
    for (int ipass = 0; ipass < npass_run; ++ipass) {
        const int w0 = pw0[ipass], ns = pw0[ipass + 1] - w0;
        const bool second = (ipass & 1) != 0;
        sbd::Params P = second ? e->P2 : e->P;
        st = (fork && second) ? e->aux : st_main;
        int32_t *const eigflag = P.eiglist;
        if (hs) HIP_TRY(hipStreamWaitEvent(st, e->ev_h2d[ipass], 0));   // this pass's inputs have landed
        P.nslot = ns;
        P.dtauc = in->dtauc + (size_t)w0 * L;
        P.ssalb = in->ssalb + (size_t)w0 * L;
        P.pmom = in->pmom_row ? in->pmom : in->pmom + (size_t)w0 * L * (e->cfg.nmom + 1);   // (rows are global indices)
        P.pmom_row = in->pmom_row ? in->pmom_row + w0 : nullptr;
        P.wvnmlo = in->wvnmlo + w0; P.wvnmhi = in->wvnmhi + w0;
        P.fbeam = in->fbeam + w0; P.albedo = in->albedo + w0; P.plank = in->plank + w0;
        P.bitem = in->bitem ? in->bitem + (size_t)w0 * 4 : nullptr;
        P.slot_base = w0;
        P.brdf_bad = e->brdf_bad ? 1 : 0;
        P.flux = out->flux + (size_t)w0 * SBD_NFLUX * nlev;
        P.uu = rad ? out->uu + (size_t)w0 * e->P.nphi * nlev * e->P.numu : nullptr;
        P.status = out->status + w0;

        if (timing) HIP_TRY(hipEventRecord(e->ev[0], st)); else if (tip) HIP_TRY(hipEventRecord(e->ev_ip[(size_t)ipass * (sbd_engine::kPhases + 1) + 0], st));
        sbd::launch_setup((unsigned)ns, st, P);
        if (P.ibdrf && !P.brdf_shared)      // the ocean's SURFAC tables and CHEKIN's test of them, per item and mode
            hipLaunchKernelGGL(sbd::surfac_kernel, dim3((unsigned)((size_t)ns * nmode)), dim3(256),
                               sizeof(double) * sbd::surf_lds_doubles(e->nn, e->P.numu), st, P, (int32_t *)nullptr);
        SBD_DBG("setup");
        if (timing) HIP_TRY(hipEventRecord(e->ev[1], st)); else if (tip) HIP_TRY(hipEventRecord(e->ev_ip[(size_t)ipass * (sbd_engine::kPhases + 1) + 1], st));
        {
            const int gpb = 64 / e->G;
            const long long groups = (long long)ns * nmode * L;
            unsigned grid = (unsigned)((groups + gpb - 1) / gpb);
            int32_t *flt = nullptr;
            if (e->use_layer2) {
                const int gpb2 = 64 / e->G2;
                const unsigned g2 = (unsigned)(((size_t)ns * L + gpb2 - 1) / gpb2) * (unsigned)nmode;
                sbd::launch_layer2(e->nn, rad && !e->quad, g2, e->layer2_lds, st, P, eigflag);
                flt = eigflag;          // the QR kernel below only redoes the listed layers: a fixed grid walks the
                if (grid > 2048u) grid = 2048u;   // list (normally empty: every block reads the count and leaves).  With
                                                  // 256 blocks a batch with 1 % of its layers listed (thermal runs with
                                                  // conservative cloud layers) spent more time here than in the fast
                                                  // kernel: 5 632 layers in 5.5 rounds of 0.13 ms (tools/fallback_probe.py)
                SBD_DBG("layer2");
            }
            // (an empty list last time this workspace ran -- the kernel's own report, e->h_hint -- : 64 blocks instead of
            //  2 048 wait for room behind the other stream's kernels; should the list be long after all, they walk it, slower)
            if (flt && e->h_hint && !getenv("SBD_NO_HINT")) {
                const int32_t hint = ((volatile int32_t *)e->h_hint)[second ? 1 : 0];
                static const unsigned hint_grid = getenv("SBD_HINT_GRID") ? (unsigned)atoi(getenv("SBD_HINT_GRID")) : 8u;   // (A/B: 64 until round 6)
                if (hint == 0 && grid > hint_grid) grid = hint_grid;
            }
            sbd::launch_layer_v1(e->G, grid, e->layer_lds, st, P, flt);
        }
        SBD_DBG("layer(v1/fallback)");
        if (timing) HIP_TRY(hipEventRecord(e->ev[2], st)); else if (tip) HIP_TRY(hipEventRecord(e->ev_ip[(size_t)ipass * (sbd_engine::kPhases + 1) + 2], st));
        {
            const unsigned bgrid = (unsigned)((size_t)ns * nmode);
            if (e->band4 && e->d_pivdbg) {
                P.pivdbg = e->d_pivdbg + (second ? (size_t)e->chunk * nmode * L * n : 0);
                if (e->pivot_exact) sbd::launch_band4_exact(e->nn, (bgrid + 3) / 4, st, P, false, true);
                else sbd::launch_band4_pivdbg(e->nn, (bgrid + 3) / 4, st, P);
            } else if (e->band4 && e->pivot_exact) sbd::launch_band4_exact(e->nn, (bgrid + 3) / 4, st, P, e->fused, false);
            else if (e->band4) sbd::launch_band4(e->nn, (bgrid + 3) / 4, st, P, e->fused);
            else if (e->band1) sbd::launch_band1(e->nn, bgrid, st, P, e->fused);
            else if (e->band_rows) sbd::launch_band_rows(e->nn, bgrid, st, P, e->fused);
            else sbd::launch_band_lds(e->nn, bgrid, e->band_lds, st, P);
        }
        SBD_DBG("band");
        if (timing) HIP_TRY(hipEventRecord(e->ev[3], st)); else if (tip) HIP_TRY(hipEventRecord(e->ev_ip[(size_t)ipass * (sbd_engine::kPhases + 1) + 3], st));
        if (e->fused) { /* the band kernel has written the fluxes */ }
        else if (e->band4) sbd::launch_backsolve4(e->nn, (unsigned)(((size_t)ns * nmode + 3) / 4), st, P);
        else if (e->band1 && !e->solve_v1) sbd::launch_backsolve1(e->nn, (unsigned)((size_t)ns * nmode), st, P);
        else sbd::launch_backsolve(e->nn, (unsigned)((size_t)ns * nmode), e->solve_lds, st, P);
        SBD_DBG("backsolve");
        if (timing) HIP_TRY(hipEventRecord(e->ev[4], st)); else if (tip) HIP_TRY(hipEventRecord(e->ev_ip[(size_t)ipass * (sbd_engine::kPhases + 1) + 4], st));
        if (rad) {
            if (e->quad) sbd::launch_cmpint((unsigned)((size_t)ns * nmode), st, P);
            else sbd::launch_usrint((unsigned)((size_t)ns * nmode), e->usr_lds, st, P);
            const long long items = (long long)ns * nlev * e->P.numu;
            sbd::launch_azimuth((unsigned)((items + 255) / 256), st, P, e->naz_run);
            if (e->corint) sbd::launch_intcor((unsigned)ns, st, P, e->naz_run);
        }
        if (!e->fused)      // (the fused band kernel writes the status words itself)
            hipLaunchKernelGGL(finish_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, P);
        // errmsg 2: the systems this pass's layer / band kernels listed, on the reference's own band matrix and LINPACK's own
        // estimate (normally none: every block reads the count and leaves)
        // (an empty list the last time this workspace ran -- the kernel's own report, h_hint[2..3] -- : two blocks instead of
        //  up to 512 have to find a free slot behind the other stream's kernels; should the list be long after all they walk
        //  it, slower, and the next pass knows)
        unsigned rc_grid = (unsigned)e->rc_blocks;
        if (e->h_hint && !getenv("SBD_NO_HINT") && ((volatile int32_t *)e->h_hint)[2 + (second ? 1 : 0)] == 0) rc_grid = getenv("SBD_HINT_GRID_RC") ? (unsigned)atoi(getenv("SBD_HINT_GRID_RC")) : 2u;
        P.rchint = e->h_hint ? e->h_hint + 2 + (second ? 1 : 0) : nullptr;
        sbd::launch_band_rcond(rc_grid, st, P, e->d_rb + (second ? (size_t)e->rc_blocks * e->rb_stride : 0), e->rb_stride,
                               e->d_rcdbg + (second ? (size_t)e->chunk * nmode : 0));
        if (hs) {   // this pass's outputs, staging -> host
            const size_t nf = (size_t)SBD_NFLUX * nlev, nu = rad ? (size_t)e->P.nphi * nlev * e->P.numu : 0;
            if (hs->out->flux) HIP_TRY(hipMemcpyAsync(hs->out->flux + (size_t)w0 * nf, P.flux, sizeof(double) * ns * nf, hipMemcpyDeviceToHost, st));
            if (rad && hs->out->uu) HIP_TRY(hipMemcpyAsync(hs->out->uu + (size_t)w0 * nu, P.uu, sizeof(double) * ns * nu, hipMemcpyDeviceToHost, st));
            if (hs->out->status) HIP_TRY(hipMemcpyAsync(hs->out->status + w0, P.status, sizeof(int32_t) * ns, hipMemcpyDeviceToHost, st));
        }
        if (timing) HIP_TRY(hipEventRecord(e->ev[5], st)); else if (tip) HIP_TRY(hipEventRecord(e->ev_ip[(size_t)ipass * (sbd_engine::kPhases + 1) + 5], st));
        HIP_TRY(hipGetLastError());
        {
            const int rcn = copy_pass(ipass + 1);
            if (rcn != SBD_OK) return rcn;
        }
        if (timing) {
            HIP_TRY(hipEventSynchronize(e->ev[sbd_engine::kPhases]));
            if (e->use_layer2) {                         // layers the fast layer kernel handed to the reference-algorithm one
                int32_t cnt = 0;
                HIP_TRY(hipMemcpy(&cnt, eigflag, sizeof(cnt), hipMemcpyDeviceToHost));
                nfallback += cnt;
            }
            for (int ph = 0; ph < sbd_engine::kPhases; ++ph) {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, e->ev[ph], e->ev[ph + 1]));
                acc_ms[ph] += ms;
            }
        }
    }
    st = st_main;
    if (fork) {
        HIP_TRY(hipEventRecord(e->ev_join, e->aux));             // the caller's stream continues when both are done
        HIP_TRY(hipStreamWaitEvent(st, e->ev_join, 0));
    }
    if (timing) {
        for (int ph = 0; ph < sbd_engine::kPhases; ++ph) e->ms_phase[ph] = acc_ms[ph];
        e->fallback_layers = nfallback;
        e->have_times = true;
    }
    return SBD_OK;
}

static int ensure_stage(sbd_engine *e, size_t bytes)
{
    if (bytes <= e->stage_bytes) return SBD_OK;
    if (e->d_stage) (void)hipFree(e->d_stage);
    e->d_stage = nullptr;
    e->stage_bytes = 0;
    hipError_t err = hipMalloc(&e->d_stage, bytes);
    if (err != hipSuccess) return fail(err == hipErrorOutOfMemory ? SBD_E_NOMEM : SBD_E_HIP, "hipMalloc(stage)");
    e->stage_bytes = bytes;
    return SBD_OK;
}

// Host-pointer solve, enqueue part: H2D of the inputs, the kernel pipeline, optionally the weighted
// sums of the batch into e->d_acc (weight != NULL), D2H of the per-item outputs the caller asked
// for -- all on the engine's stream, no synchronisation.
// IBCND = 1 from host arrays: a plain staged call (H2D, ibcnd_solve_device, D2H) -- the mode is a diagnostic of the
// medium, not a throughput path
static int ibcnd_solve_host(sbd_engine *e, const sbd_batch_in *in, const sbd_batch_out *out)
{
    if (!out->albtrn || !out->status) return fail(SBD_E_INVALID, "IBCND = 1: albtrn / status is NULL");
    if (!in->dtauc || !in->ssalb || !in->pmom || !in->wvnmlo || !in->wvnmhi || !in->albedo)
        return fail(SBD_E_INVALID, "IBCND = 1: dtauc / ssalb / pmom / wvnmlo / wvnmhi / albedo is NULL");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const size_t W = in->nwork;
    const int L = e->L, npm = L * (e->cfg.nmom + 1), nout = e->ib_nout;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_lay = up(8 * W * L), b_pm = up(8 * W * npm), b_w = up(8 * W), b_at = up(8 * W * 2 * nout), b_fl = up(8 * W * SBD_NFLUX * 2),
                 b_st = up(4 * W);
    int rc = ensure_stage(e, 2 * b_lay + b_pm + 3 * b_w + b_at + b_fl + b_st);
    if (rc != SBD_OK) return rc;
    char *p = e->d_stage;
    auto take = [&](size_t bytes) { char *r = p; p += bytes; return r; };
    double *dt = (double *)take(b_lay), *ss = (double *)take(b_lay), *pm = (double *)take(b_pm);
    double *lo = (double *)take(b_w), *hi = (double *)take(b_w), *al = (double *)take(b_w);
    double *at = (double *)take(b_at), *fl = (double *)take(b_fl);
    int32_t *stt = (int32_t *)take(b_st);
    hipStream_t st = e->stream;
    HIP_TRY(hipMemcpyAsync(dt, in->dtauc, 8 * W * L, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ss, in->ssalb, 8 * W * L, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(pm, in->pmom, 8 * W * npm, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(lo, in->wvnmlo, 8 * W, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(hi, in->wvnmhi, 8 * W, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(al, in->albedo, 8 * W, hipMemcpyHostToDevice, st));
    const sbd_batch_in din = {in->nwork, dt, ss, pm, lo, hi, nullptr, al, nullptr, nullptr};
    const sbd_batch_out dout = {fl, nullptr, stt, at};
    rc = ibcnd_solve_device(e, &din, &dout, st);
    if (rc != SBD_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out->albtrn, at, 8 * W * 2 * nout, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->status, stt, 4 * W, hipMemcpyDeviceToHost, st));
    if (out->flux) HIP_TRY(hipMemcpyAsync(out->flux, fl, 8 * W * SBD_NFLUX * 2, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    e->pending_out.clear();
    return SBD_OK;
}

static int solve_host_enqueue(sbd_engine *e, const sbd_batch_in *in, const sbd_batch_out *out, const double *weight)
{
    if (e->ibcnd) return ibcnd_solve_host(e, in, out);
    HIP_TRY(hipSetDevice(e->cfg.device));
    const size_t W = in->nwork;
    const int L = e->L, nlev = e->nlev;
    const bool rad = !e->cfg.onlyfl;
    const bool shared = in->pmom_row != nullptr;
    if (shared && in->npmom < 1) return fail(SBD_E_INVALID, "pmom_row given but npmom < 1");
    const size_t b_lay = sizeof(double) * W * L, b_pm = sizeof(double) * (shared ? (size_t)in->npmom : W) * L * (e->cfg.nmom + 1), b_w = sizeof(double) * W;
    const size_t b_flux = sizeof(double) * W * SBD_NFLUX * nlev;
    const size_t b_uu = rad ? sizeof(double) * W * e->P.nphi * nlev * e->P.numu : 0;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const bool ocean = e->P.ibdrf == 1;
    if (ocean && !in->bitem) return fail(SBD_E_INVALID, "ocean surface (ibdrf = 1): bitem is NULL");
    const size_t total = 2 * up(b_lay) + up(b_pm) + 5 * up(b_w) + up(W) + up(b_flux) + up(b_uu) + up(sizeof(int32_t) * W)
                         + (ocean ? up(4 * b_w) : 0) + (shared ? up(sizeof(int32_t) * W) : 0);
    int rc = ensure_stage(e, total);
    if (rc != SBD_OK) return rc;
    char *p = e->d_stage;
    auto take = [&](size_t bytes) { char *r = p; p += up(bytes); return r; };
    double *d_dt = (double *)take(b_lay), *d_ss = (double *)take(b_lay), *d_pm = (double *)take(b_pm);
    double *d_lo = (double *)take(b_w), *d_hi = (double *)take(b_w), *d_fb = (double *)take(b_w), *d_al = (double *)take(b_w);
    double *d_wt = (double *)take(b_w);
    uint8_t *d_pl = (uint8_t *)take(W);
    double *d_flux = (double *)take(b_flux);
    double *d_uu = rad ? (double *)take(b_uu) : nullptr;
    int32_t *d_st = (int32_t *)take(sizeof(int32_t) * W);
    double *d_bi = ocean ? (double *)take(4 * b_w) : nullptr;
    int32_t *d_row = shared ? (int32_t *)take(sizeof(int32_t) * W) : nullptr;
    bool rows_sorted = true;
    if (shared)
        for (size_t i = 0; i < W; ++i) {
            const int32_t r = in->pmom_row[i];
            if (r < 0 || r >= in->npmom) return fail(SBD_E_INVALID, "pmom_row out of range");
            if (i && r < in->pmom_row[i - 1]) rows_sorted = false;
        }
    hipStream_t st = e->stream;
    if (weight) HIP_TRY(hipMemcpyAsync(d_wt, weight, b_w, hipMemcpyHostToDevice, st));
    sbd_batch_in din = {in->nwork, d_dt, d_ss, d_pm, d_lo, d_hi, d_fb, d_al, d_pl, d_bi, d_row, shared ? in->npmom : 0};
    sbd_batch_out dout = {d_flux, d_uu, d_st};
    // pinned landing area for the outputs the caller wants (see sbd_engine::h_pin)
    const size_t w_flux = out->flux ? up(b_flux) : 0, w_uu = (rad && out->uu) ? up(b_uu) : 0, w_st = out->status ? up(sizeof(int32_t) * W) : 0;
    if (w_flux + w_uu + w_st > e->h_pin_bytes) {
        if (e->h_pin) (void)hipHostFree(e->h_pin);
        e->h_pin = nullptr;
        e->h_pin_bytes = 0;
        if (hipHostMalloc(&e->h_pin, w_flux + w_uu + w_st, hipHostMallocDefault) != hipSuccess) return fail(SBD_E_NOMEM, "hipHostMalloc(outputs)");
        e->h_pin_bytes = w_flux + w_uu + w_st;
    }
    char *hp = e->h_pin;
    sbd_batch_out pout = {nullptr, nullptr, nullptr};
    e->pending_out.clear();
    if (w_flux) { pout.flux = (double *)hp; e->pending_out.push_back({out->flux, hp, b_flux}); hp += w_flux; }
    if (w_uu) { pout.uu = (double *)hp; e->pending_out.push_back({out->uu, hp, b_uu}); hp += w_uu; }
    if (w_st) { pout.status = (int32_t *)hp; e->pending_out.push_back({out->status, hp, sizeof(int32_t) * W}); hp += w_st; }
    const HostSide hs = {in, &pout, rows_sorted};
    rc = solve_device_impl(e, &din, &dout, st, &hs);   // (each pass stages its slice in and out on its own stream)
    if (rc != SBD_OK) return rc;
    if (weight) {
        const size_t nel_f = (size_t)SBD_NFLUX * nlev, nel_u = rad ? (size_t)e->P.nphi * nlev * e->P.numu : 0;
        if (!e->d_acc) {
            HIP_TRY(hipMalloc(&e->d_acc, sizeof(double) * (nel_f + nel_u)));
            HIP_TRY(hipMalloc(&e->d_red, sizeof(double) * (nel_f + nel_u)));
        }
        HIP_TRY(hipMemsetAsync(e->d_acc, 0, sizeof(double) * (nel_f + nel_u), st));
        rc = sbd_engine_accumulate_device(e, in->nwork, d_wt, d_flux, d_uu, e->d_acc, rad ? e->d_acc + nel_f : nullptr, st);
        if (rc != SBD_OK) return rc;
    }
    return SBD_OK;
}

// The compact form (sbd_mix_in) through the same pipeline: staging for the compact arrays AND for the arguments the
// assemble kernel makes of them (the latter never exist on the host), then solve_device_impl with hs.mix set.
static int solve_mix_host_enqueue(sbd_engine *e, const sbd_mix_in *m, const sbd_batch_out *out, const double *weight)
{
    if (e->ibcnd) return fail(SBD_E_UNSUPPORTED, "compact batches: not with IBCND = 1");
    if (e->P.ibdrf == 1) return fail(SBD_E_UNSUPPORTED, "compact batches: not with the ocean surface (per-item constants)");
    if (!m->point_of || (!m->lay && m->lay_token == 0) || !m->wvnmlo || !m->wvnmhi
        || !m->fbeam || !m->albedo || !m->plank) return fail(SBD_E_INVALID, "compact batch: null input array");
    if (!m->dtaug && !m->kterm) return fail(SBD_E_INVALID, "compact batch: neither dtaug nor kterm");
    if (!m->dtaug && !e->d_gas_slots) return fail(SBD_E_INVALID, "compact batch: dtaug is NULL and no sbd_fleet_gas_terms call left gas depths on this device");
    if (m->npoint < 1) return fail(SBD_E_INVALID, "compact batch: npoint < 1");
    if (m->nterm < 0 || m->nterm > SBD_MIX_MAX_TERMS) return fail(SBD_E_INVALID, "compact batch: nterm outside 0..SBD_MIX_MAX_TERMS");
    for (int t = 0; t < m->nterm; ++t)
        if (m->family[t] < 1 || m->family[t] > 3) return fail(SBD_E_UNSUPPORTED, "compact batch: phase-function family of a term is not 1, 2 or 3 (tabulated families: arrays form)");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const size_t W = m->nwork;
    for (size_t i = 0; i < W; ++i) {
        const int32_t p = m->point_of[i];
        if (p < 0 || p >= m->npoint || (i && p < m->point_of[i - 1]))
            return fail(SBD_E_INVALID, "compact batch: point_of must be non-decreasing and inside 0..npoint-1");
    }
    const int32_t pbase = m->point_of[0];
    const size_t NP = (size_t)(m->point_of[W - 1] - pbase + 1);      // the point blocks this call's items refer to
    if (!m->dtaug) {
        if (pbase < e->gas_p0 || m->point_of[W - 1] >= e->gas_p0 + e->gas_np)
            return fail(SBD_E_INVALID, "compact batch: an item's point is not among the points whose gas depths this device holds");
        for (size_t i = 0; i < W; ++i)
            if (m->kterm[i] < 0 || m->kterm[i] >= e->gas_nk[(size_t)(m->point_of[i] - e->gas_p0)])
                return fail(SBD_E_INVALID, "compact batch: kterm outside 0..nk-1 of its point (nk as sbd_fleet_gas_terms returned it)");
    }
    const int L = e->L, nlev = e->nlev;
    const bool rad = !e->cfg.onlyfl;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_lay = sizeof(double) * W * L, b_pm = sizeof(double) * NP * L * (e->cfg.nmom + 1), b_w = sizeof(double) * W;
    const size_t b_blk = sizeof(double) * NP * (size_t)(4 + 3 * m->nterm) * L, b_p = sizeof(double) * NP;
    const size_t b_flux = sizeof(double) * W * SBD_NFLUX * nlev;
    const size_t b_uu = rad ? sizeof(double) * W * e->P.nphi * nlev * e->P.numu : 0;
    const size_t total = 3 * up(b_lay) + up(b_pm) + 5 * up(b_w) + up(W) + up(b_flux) + up(b_uu) + 4 * up(sizeof(int32_t) * W)
                         + up(b_blk) + 4 * up(b_p) + up(NP);
    int rc = ensure_stage(e, total);
    if (rc != SBD_OK) return rc;
    char *p = e->d_stage;
    auto take = [&](size_t bytes) { char *r = p; p += up(bytes); return r; };
    double *d_dt = (double *)take(b_lay), *d_ss = (double *)take(b_lay), *d_pm = (double *)take(b_pm);
    double *d_lo = (double *)take(b_w), *d_hi = (double *)take(b_w), *d_fb = (double *)take(b_w), *d_al = (double *)take(b_w);
    double *d_wt = (double *)take(b_w);
    uint8_t *d_pl = (uint8_t *)take(W);
    double *d_flux = (double *)take(b_flux);
    double *d_uu = rad ? (double *)take(b_uu) : nullptr;
    int32_t *d_st = (int32_t *)take(sizeof(int32_t) * W);
    int32_t *d_row = (int32_t *)take(sizeof(int32_t) * W);
    HostSide hs = {nullptr, nullptr, true};
    hs.mix = m;
    hs.ms.pbase = pbase;
    hs.ms.point_of = (int32_t *)take(sizeof(int32_t) * W);
    hs.ms.kterm = (int32_t *)take(sizeof(int32_t) * W);
    hs.ms.dtaug = (double *)take(b_lay);
    hs.ms.lay = (double *)take(b_blk);
    hs.ms.lay_base = pbase;
    hs.ms.lay_resident = false;
    if (m->lay_token != 0) {
        // the caller names the device copy sbd_fleet_gas_terms left (explicit residency, ABI v7)
        if (m->dtaug) return fail(SBD_E_INVALID, "compact batch: lay_token goes with dtaug == NULL (the gas call's points)");
        if (!e->d_gas_lay || m->lay_token != e->gas_token)
            return fail(SBD_E_INVALID, "compact batch: lay_token is not the one the last sbd_fleet_gas_terms call returned");
        if (e->gas_nch != 4 + 3 * m->nterm)
            return fail(SBD_E_INVALID, "compact batch: the resident layer blocks have another channel count than 4 + 3 nterm");
        hs.ms.lay = e->d_gas_lay;                   // the very blocks the gas call copied: no second trip over PCIe
        hs.ms.lay_base = e->gas_p0;
        hs.ms.lay_resident = true;
    } else if (!m->lay) {
        return fail(SBD_E_INVALID, "compact batch: lay is NULL and lay_token is 0");
    }
    hs.ms.lo = (double *)take(b_p); hs.ms.hi = (double *)take(b_p); hs.ms.fb = (double *)take(b_p); hs.ms.al = (double *)take(b_p);
    hs.ms.pl = (uint8_t *)take(NP);
    hipStream_t st = e->stream;
    if (weight) HIP_TRY(hipMemcpyAsync(d_wt, weight, b_w, hipMemcpyHostToDevice, st));
    sbd_batch_in din = {m->nwork, d_dt, d_ss, d_pm, d_lo, d_hi, d_fb, d_al, d_pl, nullptr, d_row, (int32_t)NP};
    sbd_batch_out dout = {d_flux, d_uu, d_st};
    const size_t w_flux = out->flux ? up(b_flux) : 0, w_uu = (rad && out->uu) ? up(b_uu) : 0, w_st = out->status ? up(sizeof(int32_t) * W) : 0;
    if (w_flux + w_uu + w_st > e->h_pin_bytes) {
        if (e->h_pin) (void)hipHostFree(e->h_pin);
        e->h_pin = nullptr;
        e->h_pin_bytes = 0;
        if (hipHostMalloc(&e->h_pin, w_flux + w_uu + w_st, hipHostMallocDefault) != hipSuccess) return fail(SBD_E_NOMEM, "hipHostMalloc(outputs)");
        e->h_pin_bytes = w_flux + w_uu + w_st;
    }
    char *hp = e->h_pin;
    sbd_batch_out pout = {nullptr, nullptr, nullptr};
    e->pending_out.clear();
    if (w_flux) { pout.flux = (double *)hp; e->pending_out.push_back({out->flux, hp, b_flux}); hp += w_flux; }
    if (w_uu) { pout.uu = (double *)hp; e->pending_out.push_back({out->uu, hp, b_uu}); hp += w_uu; }
    if (w_st) { pout.status = (int32_t *)hp; e->pending_out.push_back({out->status, hp, sizeof(int32_t) * W}); hp += w_st; }
    hs.out = &pout;
    rc = solve_device_impl(e, &din, &dout, st, &hs);
    if (rc != SBD_OK) return rc;
    if (weight) {
        const size_t nel_f = (size_t)SBD_NFLUX * nlev, nel_u = rad ? (size_t)e->P.nphi * nlev * e->P.numu : 0;
        if (!e->d_acc) {
            HIP_TRY(hipMalloc(&e->d_acc, sizeof(double) * (nel_f + nel_u)));
            HIP_TRY(hipMalloc(&e->d_red, sizeof(double) * (nel_f + nel_u)));
        }
        HIP_TRY(hipMemsetAsync(e->d_acc, 0, sizeof(double) * (nel_f + nel_u), st));
        rc = sbd_engine_accumulate_device(e, m->nwork, d_wt, d_flux, d_uu, e->d_acc, rad ? e->d_acc + nel_f : nullptr, st);
        if (rc != SBD_OK) return rc;
    }
    return SBD_OK;
}

// after the engine's stream has drained: the outputs of the last host-pointer solve, pinned buffer -> caller
static void deliver_host_outputs(sbd_engine *e)
{
    for (const auto &po : e->pending_out) memcpy(po.dst, po.src, po.bytes);
    e->pending_out.clear();
}

int sbd_engine_solve_host(sbd_engine *e, const sbd_batch_in *in, const sbd_batch_out *out)
{
    if (!e || !in || !out) return fail(SBD_E_INVALID, "null argument");
    if (in->nwork <= 0) return in->nwork == 0 ? SBD_OK : fail(SBD_E_INVALID, "nwork < 0");
    if ((!out->flux && !e->ibcnd) || !out->status) return fail(SBD_E_INVALID, "null output array");
    int rc = solve_host_enqueue(e, in, out, nullptr);
    if (rc != SBD_OK) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    deliver_host_outputs(e);
    return SBD_OK;
}

int sbd_engine_accumulate_device(sbd_engine *e, int32_t nwork, const double *weight, const double *flux,
                                 const double *uu, double *acc_flux, double *acc_uu, void *hip_stream)
{
    if (!e || nwork < 0 || !weight || !flux || !acc_flux) return fail(SBD_E_INVALID, "null argument");
    if (nwork == 0) return SBD_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    const int nseg = (nwork + 255) / 256;
    const int nel_f = SBD_NFLUX * e->nlev;
    const int nel_u = (uu && acc_uu) ? e->P.nphi * e->nlev * e->P.numu : 0;
    const size_t need = (size_t)nseg * (nel_f > nel_u ? nel_f : nel_u);
    if (need > e->partial_elems) {
        if (e->d_partial) (void)hipFree(e->d_partial);
        e->d_partial = nullptr;
        e->partial_elems = 0;
        HIP_TRY(hipMalloc(&e->d_partial, need * sizeof(double)));
        e->partial_elems = need;
    }
    const size_t acc_lds = sizeof(double) * kAccTile * 16;
    hipLaunchKernelGGL(accum_partial_kernel, dim3(nseg), dim3(256), acc_lds, st, (int)nwork, nel_f, weight, flux, e->d_partial);
    hipLaunchKernelGGL(accum_final_kernel, dim3((nel_f + 15) / 16), dim3(256), acc_lds, st, nseg, nel_f, (const double *)e->d_partial, acc_flux);
    if (nel_u > 0) {
        hipLaunchKernelGGL(accum_partial_kernel, dim3(nseg), dim3(256), acc_lds, st, (int)nwork, nel_u, weight, uu, e->d_partial);
        hipLaunchKernelGGL(accum_final_kernel, dim3((nel_u + 15) / 16), dim3(256), acc_lds, st, nseg, nel_u, (const double *)e->d_partial, acc_uu);
    }
    HIP_TRY(hipGetLastError());
    return SBD_OK;
}

int sbd_engine_accumulate_host(sbd_engine *e, int32_t nwork, const double *weight, const double *flux,
                               const double *uu, double *acc_flux, double *acc_uu)
{
    if (!e || nwork < 0 || !weight || !flux || !acc_flux) return fail(SBD_E_INVALID, "null argument");
    if (nwork == 0) return SBD_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const size_t W = nwork;
    const int nel_f = SBD_NFLUX * e->nlev;
    const int nel_u = (uu && acc_uu) ? e->P.nphi * e->nlev * e->P.numu : 0;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t total = up(8 * W) + up(8 * W * nel_f) + up(8 * W * nel_u) + up(8 * (size_t)nel_f) + up(8 * (size_t)nel_u);
    int rc = ensure_stage(e, total);
    if (rc != SBD_OK) return rc;
    char *p = e->d_stage;
    auto take = [&](size_t bytes) { char *r = p; p += up(bytes); return r; };
    double *d_w = (double *)take(8 * W), *d_f = (double *)take(8 * W * nel_f);
    double *d_u = nel_u ? (double *)take(8 * W * nel_u) : nullptr;
    double *d_af = (double *)take(8 * (size_t)nel_f), *d_au = nel_u ? (double *)take(8 * (size_t)nel_u) : nullptr;
    hipStream_t st = e->stream;
    HIP_TRY(hipMemcpyAsync(d_w, weight, 8 * W, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_f, flux, 8 * W * nel_f, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_af, acc_flux, 8 * (size_t)nel_f, hipMemcpyHostToDevice, st));
    if (nel_u) {
        HIP_TRY(hipMemcpyAsync(d_u, uu, 8 * W * nel_u, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_au, acc_uu, 8 * (size_t)nel_u, hipMemcpyHostToDevice, st));
    }
    rc = sbd_engine_accumulate_device(e, nwork, d_w, d_f, d_u, d_af, d_au, st);
    if (rc != SBD_OK) return rc;
    HIP_TRY(hipMemcpyAsync(acc_flux, d_af, 8 * (size_t)nel_f, hipMemcpyDeviceToHost, st));
    if (nel_u) HIP_TRY(hipMemcpyAsync(acc_uu, d_au, 8 * (size_t)nel_u, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return SBD_OK;
}

// ===================== several GPUs from one process =====================
// A fleet is one engine per device.  Work items are independent (drt.f:425-561), so a batch is cut
// into contiguous shards (sbd_shard_range), every device solves its shard with no data-path
// exchange, and the only collective is the sum of the weighted accumulator blocks (stdout1's
// spectral sums, drt.f:1047-1054): one ncclReduce(sum, double) over xGMI when the devices are
// distinct, a host-side sum in device order otherwise (the same engine twice: test configurations).
// RCCL is loaded on demand (a fleet over several distinct devices), never at link time: a process that
// drives one GPU -- or that hosts another HIP/RCCL user such as PyTorch -- does not get a second copy
// of the collective library mapped into it.
struct RcclApi {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    bool ok = false;
};
static const RcclApi &rccl_api()
{
    static RcclApi api = [] {
        RcclApi a;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
        a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
        a.Reduce = (decltype(a.Reduce))dlsym(h, "ncclReduce");
        a.ok = a.CommInitAll && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Reduce;
        return a;
    }();
    return api;
}

struct sbd_fleet {
    std::vector<sbd_engine *> eng;
    std::vector<ncclComm_t> comm;     // empty: host-side sum
    std::vector<double> hacc;         // [ndev][nel] staging of the host-side sum
    std::vector<double> t_enq;        // [ndev][2] host clock (s since the call began) around each device's enqueue
    int retry_nstr = 0;
    int pinned_last = 0;              // arrays of the last call that were page-locked for its duration
    std::vector<std::vector<int32_t>> shard_rows;   // [ndev] a shard's moment-block rows counted from its first block (kept until the next call)
};

void sbd_shard_range(int32_t nwork, int32_t nshard, int32_t rank, int32_t *lo, int32_t *hi)
{
    // contiguous blocks of ceil/floor(nwork/nshard) items, the first nwork % nshard shards one longer
    if (nshard < 1) nshard = 1;
    const int32_t base = nwork / nshard, extra = nwork % nshard;
    const int32_t l = rank * base + (rank < extra ? rank : extra);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < extra ? 1 : 0);
}

// ... for a batch in compact form: the same balanced item boundaries, each moved UP to the next item that starts a
// spectral point (point_of non-decreasing), so that the k-terms of a point stay on one device -- which then forms the
// point's moments once -- and every shard refers to a contiguous range of point blocks disjoint from its neighbours'.
void sbd_shard_range_points(int32_t nwork, const int32_t *point_of, int32_t nshard, int32_t rank, int32_t *lo, int32_t *hi)
{
    auto snap = [&](int32_t i) {
        if (!point_of) return i;
        while (i > 0 && i < nwork && point_of[i] == point_of[i - 1]) ++i;
        return i;
    };
    int32_t l = 0, h = 0;
    sbd_shard_range(nwork, nshard, rank, &l, &h);
    if (lo) *lo = snap(l);
    if (hi) *hi = snap(h);
}

void sbd_fleet_destroy(sbd_fleet *f)
{
    if (!f) return;
    for (auto c : f->comm) (void)rccl_api().CommDestroy(c);
    for (auto e : f->eng) sbd_engine_destroy(e);
    delete f;
}

int32_t sbd_fleet_size(const sbd_fleet *f) { return f ? (int32_t)f->eng.size() : 0; }
sbd_engine *sbd_fleet_engine(sbd_fleet *f, int32_t i) { return (f && i >= 0 && i < (int32_t)f->eng.size()) ? f->eng[i] : nullptr; }
int32_t sbd_fleet_uses_rccl(const sbd_fleet *f) { return f && !f->comm.empty(); }

int sbd_fleet_create(const sbd_run_cfg *cfg, int32_t ndev, const int32_t *devices, sbd_fleet **out)
{
    if (!cfg || !out) return fail(SBD_E_INVALID, "null argument");
    *out = nullptr;
    int nvis = 0;
    if (hipGetDeviceCount(&nvis) != hipSuccess || nvis <= 0) return fail(SBD_E_NO_DEVICE, "hipGetDeviceCount");
    std::vector<int> dev;
    if (ndev <= 0 || !devices) {
        for (int d = 0; d < nvis; ++d) dev.push_back(d);     // every visible device
    } else {
        for (int i = 0; i < ndev; ++i) dev.push_back(devices[i]);
    }
    sbd_fleet *f = new (std::nothrow) sbd_fleet;
    if (!f) return fail(SBD_E_NOMEM, "host allocation");
    int rc_all = SBD_OK;
    for (int d : dev) {
        sbd_run_cfg c = *cfg;
        c.device = d;
        sbd_engine *e = nullptr;
        const int rc = sbd_engine_create(&c, &e);
        if (rc == SBD_E_RETRY_NSTR) { rc_all = rc; f->retry_nstr = 1; }
        else if (rc != SBD_OK) { sbd_fleet_destroy(f); return rc; }
        f->eng.push_back(e);
    }
    // (SBD_FLEET_RCCL=1: a communicator also for a fleet of ONE device -- nranks = 1 is legal -- so that the
    //  collective path, dlopen to teardown, runs under test on a one-GPU box)
    const char *force = getenv("SBD_FLEET_RCCL");
    bool distinct = dev.size() > 1 || (force && atoi(force) != 0);
    for (size_t i = 0; i < dev.size(); ++i)
        for (size_t j = i + 1; j < dev.size(); ++j)
            if (dev[i] == dev[j]) distinct = false;
    if (distinct && rccl_api().ok) {
        f->comm.resize(dev.size());
        if (rccl_api().CommInitAll(f->comm.data(), (int)dev.size(), dev.data()) != ncclSuccess) {
            f->comm.clear();            // no RCCL path on this system: fall back to the host-side sum
        }
    }
    *out = f;
    return rc_all;
}

// after every busy device's shard has been enqueued: the one collective of the path -- the sum of the accumulator blocks
// onto device 0 over xGMI (RCCL), or on the host in device order --, then the per-item outputs pinned buffer -> caller
static int fleet_finish(sbd_fleet *f, const std::vector<int> &busy, const bool weight, double *acc_flux, double *acc_uu)
{
    const int nd = (int)f->eng.size();
    sbd_engine *e0 = f->eng[0];
    const int nlev = e0->nlev;
    const bool rad = !e0->cfg.onlyfl;
    const size_t nel_f = (size_t)SBD_NFLUX * nlev, nel_u = rad ? (size_t)e0->P.nphi * nlev * e0->P.numu : 0, nel = nel_f + nel_u;
    if (weight) {
        if (!f->comm.empty() && (int)busy.size() == nd) {
            // the one collective of the path: sum of the accumulator blocks onto device 0 over xGMI
            if (rccl_api().GroupStart() != ncclSuccess) return fail(SBD_E_HIP, "ncclGroupStart");
            for (int r = 0; r < nd; ++r) {
                HIP_TRY(hipSetDevice(f->eng[r]->cfg.device));
                if (rccl_api().Reduce(f->eng[r]->d_acc, f->eng[r]->d_red, nel, ncclDouble, ncclSum, 0, f->comm[r], f->eng[r]->stream) != ncclSuccess)
                    return fail(SBD_E_HIP, "ncclReduce");
            }
            if (rccl_api().GroupEnd() != ncclSuccess) return fail(SBD_E_HIP, "ncclGroupEnd");
            f->hacc.assign(nel, 0.0);
            HIP_TRY(hipSetDevice(e0->cfg.device));
            HIP_TRY(hipMemcpyAsync(f->hacc.data(), e0->d_red, sizeof(double) * nel, hipMemcpyDeviceToHost, e0->stream));
            for (int r : busy) { HIP_TRY(hipSetDevice(f->eng[r]->cfg.device)); HIP_TRY(hipStreamSynchronize(f->eng[r]->stream)); deliver_host_outputs(f->eng[r]); }
            for (size_t i = 0; i < nel_f; ++i) acc_flux[i] += f->hacc[i];
            if (acc_uu) for (size_t i = 0; i < nel_u; ++i) acc_uu[i] += f->hacc[nel_f + i];
        } else {
            f->hacc.assign((size_t)nd * nel, 0.0);
            for (int r : busy) {
                HIP_TRY(hipSetDevice(f->eng[r]->cfg.device));
                HIP_TRY(hipMemcpyAsync(f->hacc.data() + (size_t)r * nel, f->eng[r]->d_acc, sizeof(double) * nel, hipMemcpyDeviceToHost, f->eng[r]->stream));
            }
            for (int r : busy) { HIP_TRY(hipSetDevice(f->eng[r]->cfg.device)); HIP_TRY(hipStreamSynchronize(f->eng[r]->stream)); deliver_host_outputs(f->eng[r]); }
            for (int r : busy) {   // fixed order: device 0's block first
                const double *h = f->hacc.data() + (size_t)r * nel;
                for (size_t i = 0; i < nel_f; ++i) acc_flux[i] += h[i];
                if (acc_uu) for (size_t i = 0; i < nel_u; ++i) acc_uu[i] += h[nel_f + i];
            }
        }
    } else {
        for (int r : busy) { HIP_TRY(hipSetDevice(f->eng[r]->cfg.device)); HIP_TRY(hipStreamSynchronize(f->eng[r]->stream)); deliver_host_outputs(f->eng[r]); }
    }
    return SBD_OK;
}

int sbd_fleet_solve_host(sbd_fleet *f, const sbd_batch_in *in, const sbd_batch_out *out,
                         const double *weight, double *acc_flux, double *acc_uu)
{
    if (!f || !in || !out || f->eng.empty()) return fail(SBD_E_INVALID, "null argument");
    if (in->nwork < 0) return fail(SBD_E_INVALID, "nwork < 0");
    if (!out->status) return fail(SBD_E_INVALID, "status is NULL");
    if (weight && !acc_flux) return fail(SBD_E_INVALID, "acc_flux is NULL");
    const int nd = (int)f->eng.size();
    sbd_engine *e0 = f->eng[0];
    if (e0->ibcnd) weight = nullptr;                 // (albedo / transmissivity of the medium: nothing to integrate)
    const int L = e0->L, nlev = e0->nlev, nmom1 = e0->cfg.nmom + 1;
    const bool rad = !e0->cfg.onlyfl;
    const size_t nel_f = (size_t)SBD_NFLUX * nlev;
    const size_t uu_item = rad ? (size_t)e0->P.nphi * nlev * e0->P.numu : 0;
    std::vector<int> busy;
    for (int r = 0; r < nd; ++r) {
        int32_t lo, hi;
        sbd_shard_range(in->nwork, nd, r, &lo, &hi);
        if (hi > lo) busy.push_back(r);
    }
    // The caller's arrays are usually pageable (a Fortran ALLOCATE): a hipMemcpyAsync from them is staged by the
    // runtime and returns when the copy is done, so ONE enqueueing thread would feed the devices one after another
    // (0.67 GB per device and step at the bench's size, far more than the 10 ms of compute).  Two measures:
    //  * every device's shard is enqueued from a host thread of its own -- the devices' copies and kernels overlap
    //    whatever the memory is;
    //  * for a fleet of several devices (or SBD_PIN_INPUTS=1) the three large input arrays are page-locked for the
    //    duration of the call (hipHostRegister, portable): the copies become DMA at PCIe speed from any device.
    struct PinGuard {                                   // (unregistered on every way out of the call)
        std::vector<void *> p;
        ~PinGuard() { for (void *x : p) (void)hipHostUnregister(x); }
    } pins;
    std::vector<void *> &pinned = pins.p;
    {
        const char *pe = getenv("SBD_PIN_INPUTS");
        const size_t big = sizeof(double) * (size_t)(in->pmom_row ? in->npmom : in->nwork) * L * nmom1;
        const bool pin = pe ? atoi(pe) != 0 : (busy.size() > 1 && big >= ((size_t)32 << 20));
        if (pin) {
            const std::pair<const void *, size_t> arr[3] = {{in->dtauc, sizeof(double) * (size_t)in->nwork * L},
                                                            {in->ssalb, sizeof(double) * (size_t)in->nwork * L}, {in->pmom, big}};
            for (const auto &a : arr) {
                if (hipHostRegister((void *)a.first, a.second, hipHostRegisterPortable) == hipSuccess) pinned.push_back((void *)a.first);
                else (void)hipGetLastError();          // (already registered -- PyTorch pinned memory -- or not registrable)
            }
        }
        f->pinned_last = (int)pinned.size();
    }
    f->t_enq.assign((size_t)nd * 2, 0.0);
    f->shard_rows.resize((size_t)nd);
    {
        const auto t_begin = std::chrono::steady_clock::now();
        std::vector<int> rcs(nd, SBD_OK);
        std::vector<std::string> errs(nd);
        auto enqueue = [&](const int r) {
            int32_t lo, hi;
            sbd_shard_range(in->nwork, nd, r, &lo, &hi);
            sbd_batch_in si = {hi - lo, in->dtauc + (size_t)lo * L, in->ssalb + (size_t)lo * L, in->pmom + (size_t)lo * L * nmom1,
                               in->wvnmlo + lo, in->wvnmhi + lo, in->fbeam + lo, in->albedo + lo, in->plank + lo,
                               in->bitem ? in->bitem + (size_t)lo * 4 : nullptr,
                               in->pmom_row ? in->pmom_row + lo : nullptr, in->npmom};
            if (in->pmom_row) {
                // moments per spectral point.  Items in wavelength order (rows non-decreasing inside the shard): the
                // shard is handed ITS blocks only, rows counted from its first one -- its device stages r1 - r0 + 1
                // blocks, not all npmom of them (round 3 sized every device's staging area for the whole list).
                // Unsorted rows: the whole block list, global indices.
                si.pmom = in->pmom;
                bool sorted = hi > lo;
                for (int32_t i = lo + 1; i < hi && sorted; ++i) sorted = in->pmom_row[i] >= in->pmom_row[i - 1];
                const int32_t r0 = sorted ? in->pmom_row[lo] : 0, r1 = sorted ? in->pmom_row[hi - 1] : -1;
                if (sorted && nd > 1 && r0 >= 0 && r1 < in->npmom) {
                    std::vector<int32_t> &rows = f->shard_rows[r];
                    rows.resize((size_t)(hi - lo));
                    for (int32_t i = lo; i < hi; ++i) rows[(size_t)(i - lo)] = in->pmom_row[i] - r0;
                    si.pmom = in->pmom + (size_t)r0 * L * nmom1;
                    si.pmom_row = rows.data();
                    si.npmom = r1 - r0 + 1;
                }
            }
            sbd_batch_out so = {out->flux ? out->flux + (size_t)lo * nel_f : nullptr,
                                (rad && out->uu) ? out->uu + (size_t)lo * uu_item : nullptr, out->status + lo,
                                (e0->ibcnd && out->albtrn) ? out->albtrn + (size_t)lo * 2 * e0->ib_nout : nullptr};
            f->t_enq[2 * r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
            rcs[r] = solve_host_enqueue(f->eng[r], &si, &so, weight ? weight + lo : nullptr);
            if (rcs[r] != SBD_OK) errs[r] = g_last_error;          // (thread-local: carried to the caller below)
            f->t_enq[2 * r + 1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        };
        if (busy.size() > 1) {
            std::vector<std::thread> th;
            for (int r : busy) th.emplace_back(enqueue, r);
            for (auto &t : th) t.join();
        } else if (!busy.empty()) {
            enqueue(busy[0]);
        }
        for (int r : busy)
            if (rcs[r] != SBD_OK) {
                for (int q : busy) { (void)hipSetDevice(f->eng[q]->cfg.device); (void)hipStreamSynchronize(f->eng[q]->stream); }
                return fail(rcs[r], errs[r]);
            }
    }
    return fleet_finish(f, busy, weight != nullptr, acc_flux, acc_uu);
}

int sbd_fleet_solve_mix_host(sbd_fleet *f, const sbd_mix_in *in, const sbd_batch_out *out,
                             const double *weight, double *acc_flux, double *acc_uu)
{
    if (!f || !in || !out || f->eng.empty()) return fail(SBD_E_INVALID, "null argument");
    if (in->nwork <= 0) return in->nwork == 0 ? SBD_OK : fail(SBD_E_INVALID, "nwork < 0");
    if (!out->status) return fail(SBD_E_INVALID, "status is NULL");
    if (weight && !acc_flux) return fail(SBD_E_INVALID, "acc_flux is NULL");
    if (!in->point_of) return fail(SBD_E_INVALID, "compact batch: null input array");
    const int nd = (int)f->eng.size();
    sbd_engine *e0 = f->eng[0];
    const int L = e0->L, nlev = e0->nlev;
    const bool rad = !e0->cfg.onlyfl;
    const size_t nel_f = (size_t)SBD_NFLUX * nlev;
    const size_t uu_item = rad ? (size_t)e0->P.nphi * nlev * e0->P.numu : 0;
    for (int32_t i = 1; i < in->nwork; ++i)          // (the cut rule below needs it; every shard checks its own range again)
        if (in->point_of[i] < in->point_of[i - 1]) return fail(SBD_E_INVALID, "compact batch: point_of must be non-decreasing and inside 0..npoint-1");
    // shards cut between spectral points (sbd_shard_range_points): neighbours share no point block
    std::vector<int> busy;
    std::vector<int32_t> slo(nd), shi(nd);
    for (int r = 0; r < nd; ++r) {
        if (in->dtaug) {
            sbd_shard_range_points(in->nwork, in->point_of, nd, r, &slo[r], &shi[r]);
        } else {
            // gas depths resident on the devices (sbd_fleet_gas_terms): an item goes where its point's depths are
            const int32_t *b = in->point_of, *e_ = in->point_of + in->nwork;
            slo[r] = (int32_t)(std::lower_bound(b, e_, f->eng[r]->gas_p0) - b);
            shi[r] = (int32_t)(std::lower_bound(b, e_, f->eng[r]->gas_p0 + f->eng[r]->gas_np) - b);
            if (!f->eng[r]->d_gas_slots) shi[r] = slo[r];
        }
        if (shi[r] > slo[r]) busy.push_back(r);
    }
    if (!in->dtaug) {
        int64_t covered = 0;
        for (int r : busy) covered += shi[r] - slo[r];
        if (covered != in->nwork) return fail(SBD_E_INVALID, "compact batch: dtaug is NULL and some item's point was not in the last sbd_fleet_gas_terms call");
    }
    struct PinGuard {
        std::vector<void *> p;
        ~PinGuard() { for (void *x : p) (void)hipHostUnregister(x); }
    } pins;
    {
        const char *pe = getenv("SBD_PIN_INPUTS");
        const size_t big = sizeof(double) * (size_t)in->nwork * L;
        const bool pin = pe ? atoi(pe) != 0 : (busy.size() > 1 && big >= ((size_t)32 << 20));
        if (pin && in->dtaug && in->lay) {
            const std::pair<const void *, size_t> arr[2] = {{in->dtaug, big},
                                                            {in->lay, sizeof(double) * (size_t)in->npoint * (4 + 3 * (in->nterm > 0 ? in->nterm : 0)) * L}};
            for (const auto &a : arr) {
                if (hipHostRegister((void *)a.first, a.second, hipHostRegisterPortable) == hipSuccess) pins.p.push_back((void *)a.first);
                else (void)hipGetLastError();
            }
        }
        f->pinned_last = (int)pins.p.size();
    }
    f->t_enq.assign((size_t)nd * 2, 0.0);
    {
        const auto t_begin = std::chrono::steady_clock::now();
        std::vector<int> rcs(nd, SBD_OK);
        std::vector<std::string> errs(nd);
        auto enqueue = [&](const int r) {
            const int32_t lo = slo[r], hi = shi[r];
            sbd_mix_in si = *in;                                     // (point blocks by their GLOBAL index: the engine stages the range its items refer to)
            si.nwork = hi - lo;
            si.point_of = in->point_of + lo;
            si.dtaug = in->dtaug ? in->dtaug + (size_t)lo * L : nullptr;
            si.kterm = in->kterm ? in->kterm + lo : nullptr;
            sbd_batch_out so = {out->flux ? out->flux + (size_t)lo * nel_f : nullptr,
                                (rad && out->uu) ? out->uu + (size_t)lo * uu_item : nullptr, out->status + lo, nullptr};
            f->t_enq[2 * r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
            rcs[r] = solve_mix_host_enqueue(f->eng[r], &si, &so, weight ? weight + lo : nullptr);
            if (rcs[r] != SBD_OK) errs[r] = g_last_error;
            f->t_enq[2 * r + 1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        };
        if (busy.size() > 1) {
            std::vector<std::thread> th;
            for (int r : busy) th.emplace_back(enqueue, r);
            for (auto &t : th) t.join();
        } else if (!busy.empty()) {
            enqueue(busy[0]);
        }
        for (int r : busy)
            if (rcs[r] != SBD_OK) {
                for (int q : busy) { (void)hipSetDevice(f->eng[q]->cfg.device); (void)hipStreamSynchronize(f->eng[q]->stream); }
                return fail(rcs[r], errs[r]);
            }
    }
    return fleet_finish(f, busy, weight != nullptr, acc_flux, acc_uu);
}

// The gas part of the band model for the points of a run, on the fleet's devices (sbd_gas.hpp through sbd_k_gas.hip):
// points sharded by sbd_shard_range, every device keeps its points' gas depths for the compact-form solves that follow.
static int fleet_point_terms_impl(sbd_fleet *f, const sbd_gas_model *g, const sbd_scat_model *sm, int32_t npoint, const double *wl,
                                  const double *lay, int32_t nch, int32_t *nk, double *wt, int32_t *failed, double *dtaug_out,
                                  double *lay_out, int64_t *lay_token)
{
    static std::atomic<int64_t> generation{0};
    if (lay_token) *lay_token = 0;
    if (!f || f->eng.empty() || !g || !wl || (!lay && !sm) || !nk || !wt) return fail(SBD_E_INVALID, "null argument");
    if (sm && (!sm->z || !sm->p || !sm->t || sm->nz != f->eng[0]->L)) return fail(SBD_E_INVALID, "scatter model: null profile, or nz differs from the fleet's NLYR");
    if (!g->uu || !g->z || !g->tables) return fail(SBD_E_INVALID, "gas model: null array");
    if (npoint <= 0) return npoint == 0 ? SBD_OK : fail(SBD_E_INVALID, "npoint < 0");
    sbd_engine *e0 = f->eng[0];
    const int L = e0->L;
    if (g->nz != L) return fail(SBD_E_INVALID, "gas model: nz differs from the fleet's NLYR");
    if (nch < 3) return fail(SBD_E_INVALID, "gas model: the layer blocks need the cloud, aerosol and Rayleigh channels");
    if (g->kdist < 0 || g->kdist > 3) return fail(SBD_E_INVALID, "gas model: KDIST outside 0..3");
    sbd::GasTablesPacked pk;
    std::string perr;
    if (!pk.parse(g->tables, g->tables_bytes, perr)) return fail(SBD_E_INVALID, perr);
    const int nd = (int)f->eng.size();
    std::vector<int> rcs(nd, SBD_OK);
    std::vector<std::string> errs(nd);
    auto run = [&](const int r) {
        sbd_engine *e = f->eng[r];
        int32_t lo, hi;
        sbd_shard_range(npoint, nd, r, &lo, &hi);
        const int np = hi - lo;
        auto bad = [&](hipError_t he, const char *what) {
            if (he == hipSuccess) return false;
            rcs[r] = SBD_E_HIP;
            errs[r] = std::string(what) + ": " + hipGetErrorString(he);
            (void)hipGetLastError();
            return true;
        };
        if (bad(hipSetDevice(e->cfg.device), "hipSetDevice")) return;
        if (e->d_gas_slots) { (void)hipFree(e->d_gas_slots); e->d_gas_slots = nullptr; }
        if (e->d_gas_lay) { (void)hipFree(e->d_gas_lay); e->d_gas_lay = nullptr; }
        e->gas_p0 = lo; e->gas_np = 0; e->gas_nch = 0; e->gas_token = 0; e->gas_nk.clear();
        if (np <= 0) return;
        const size_t npad = ((size_t)np + 63) & ~(size_t)63;
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t b_td = up(8 * pk.d.size()), b_ti = up(4 * pk.i.size()), b_uu = up(8 * (size_t)L * SBD_GAS_SLOTS), b_z = up(8 * (size_t)L);
        const size_t b_wl = up(8 * (size_t)np), b_lay = up(8 * (size_t)np * nch * L), b_ws = up(8 * (size_t)16 * L * npad);
        const size_t b_nk = up(4 * (size_t)np), b_wt = up(8 * (size_t)np * 3);
        const size_t naw = (sm && sm->iaer != 0) ? (size_t)sm->aer_nwl : 0;
        const size_t b_sm = sm ? up(8 * (4 * (size_t)L + 4 * naw)) : 0;      // z, p, t, column, the boundary-layer spectrum
        const bool timing = getenv("SBD_TIMING") != nullptr;
        const auto t_0 = std::chrono::steady_clock::now();
        auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count(); };
        double t_alloc = 0, t_h2d = 0, t_kernel = 0;
        char *tmp = nullptr;
        if (bad(hipMalloc(&tmp, b_td + b_ti + b_uu + b_z + b_wl + b_ws + 2 * b_nk + b_wt + b_sm), "hipMalloc(gas work area)")) return;
        if (bad(hipMalloc(&e->d_gas_slots, 8 * (size_t)np * 3 * L), "hipMalloc(gas depths)")) { (void)hipFree(tmp); return; }
        if (bad(hipMalloc(&e->d_gas_lay, b_lay), "hipMalloc(layer blocks)")) { (void)hipFree(tmp); (void)hipFree(e->d_gas_slots); e->d_gas_slots = nullptr; return; }
        char *q = tmp;
        auto take = [&](size_t b) { char *x = q; q += b; return x; };
        double *d_td = (double *)take(b_td);
        int32_t *d_ti = (int32_t *)take(b_ti);
        double *d_uu = (double *)take(b_uu), *d_z = (double *)take(b_z), *d_wl = (double *)take(b_wl), *d_lay = e->d_gas_lay;
        double *d_ws = (double *)take(b_ws);
        int32_t *d_nk = (int32_t *)take(b_nk), *d_fail = (int32_t *)take(b_nk);
        double *d_wt = (double *)take(b_wt);
        double *d_sm = sm ? (double *)take(b_sm) : nullptr;
        hipStream_t st = e->stream;
        bool err = false;
        t_alloc = since();
        err = err || bad(hipMemcpyAsync(d_td, pk.d.data(), 8 * pk.d.size(), hipMemcpyHostToDevice, st), "H2D tables");
        err = err || bad(hipMemcpyAsync(d_ti, pk.i.data(), 4 * pk.i.size(), hipMemcpyHostToDevice, st), "H2D tables");
        err = err || bad(hipMemcpyAsync(d_uu, g->uu, 8 * (size_t)L * SBD_GAS_SLOTS, hipMemcpyHostToDevice, st), "H2D uu");
        err = err || bad(hipMemcpyAsync(d_z, g->z, 8 * (size_t)L, hipMemcpyHostToDevice, st), "H2D z");
        err = err || bad(hipMemcpyAsync(d_wl, wl + lo, 8 * (size_t)np, hipMemcpyHostToDevice, st), "H2D wl");
        if (!sm) err = err || bad(hipMemcpyAsync(d_lay, lay + (size_t)lo * nch * L, 8 * (size_t)np * nch * L, hipMemcpyHostToDevice, st), "H2D lay");
        err = err || bad(hipMemsetAsync(e->d_gas_slots, 0, 8 * (size_t)np * 3 * L, st), "memset");
        if (!err) {
            sbd::GasRun R;
            std::string verr;
            if (!pk.view(d_td, d_ti, R.T, verr)) { rcs[r] = SBD_E_INVALID; errs[r] = "gas tables: " + verr; err = true; }
            R.uu = d_uu; R.z = d_z; R.nz = L; R.kdist = g->kdist;
            R.amu0_first = g->amu0_first; R.amu0_rest = g->amu0_rest; R.xo4 = g->xo4; R.re_earth = sbd::kReEarth;
            if (sm && !err) {
                // the layer blocks are MADE here (scatter_kernel on sbd_scat.hpp): the model's small arrays go over, the
                // blocks are born where the gas kernel and the solves read them
                sbd_scat_model dsm = *sm;
                double *q2 = d_sm;
                auto put = [&](const double *src, size_t cnt) -> const double * {
                    double *dst = q2;
                    q2 += cnt;
                    if (cnt) err = err || bad(hipMemcpyAsync(dst, src, 8 * cnt, hipMemcpyHostToDevice, st), "H2D scatter model");
                    return dst;
                };
                dsm.z = put(sm->z, L); dsm.p = put(sm->p, L); dsm.t = put(sm->t, L);
                dsm.aer_column = (sm->iaer != 0) ? put(sm->aer_column, L) : nullptr;
                if (naw) { dsm.aer_wl = put(sm->aer_wl, naw); dsm.aer_ext = put(sm->aer_ext, naw); dsm.aer_absb = put(sm->aer_absb, naw); dsm.aer_asym = put(sm->aer_asym, naw); }
                std::string serr;
                if (!err && !sbd::launch_scatter_abi(st, &dsm, pk, d_td, np, d_wl, d_lay, nch, serr)) { rcs[r] = SBD_E_INVALID; errs[r] = serr; err = true; }
                err = err || bad(hipGetLastError(), "scatter_kernel");
                if (lay_out && !err) err = err || bad(hipMemcpyAsync(lay_out + (size_t)lo * nch * L, d_lay, 8 * (size_t)np * nch * L, hipMemcpyDeviceToHost, st), "D2H layer blocks");
            }
            if (timing) { (void)hipStreamSynchronize(st); t_h2d = since(); }
            if (!err) sbd::launch_gas(st, R, np, lo == 0 ? 1 : 0, d_wl, d_lay, nch, d_ws, npad, d_nk, d_wt, d_fail, e->d_gas_slots);
            err = err || bad(hipGetLastError(), "gas_kernel");
            if (timing) { (void)hipStreamSynchronize(st); t_kernel = since(); }
            err = err || bad(hipMemcpyAsync(nk + lo, d_nk, 4 * (size_t)np, hipMemcpyDeviceToHost, st), "D2H nk");
            err = err || bad(hipMemcpyAsync(wt + (size_t)lo * 3, d_wt, 8 * (size_t)np * 3, hipMemcpyDeviceToHost, st), "D2H wt");
            if (failed) err = err || bad(hipMemcpyAsync(failed + lo, d_fail, 4 * (size_t)np, hipMemcpyDeviceToHost, st), "D2H fail");
            if (dtaug_out) err = err || bad(hipMemcpyAsync(dtaug_out + (size_t)lo * 3 * L, e->d_gas_slots, 8 * (size_t)np * 3 * L, hipMemcpyDeviceToHost, st), "D2H depths");
        }
        (void)bad(hipStreamSynchronize(st), "gas_kernel (sync)");
        const double t_d2h = since();
        (void)hipFree(tmp);
        if (timing) fprintf(stderr, "sbdart_amd: gas terms on device %d: %d points; alloc %.2f ms, H2D %.2f ms, gas_kernel %.2f ms, D2H %.2f ms, free %.2f ms\n",
                            e->cfg.device, np, t_alloc, t_h2d - t_alloc, t_kernel - t_h2d, t_d2h - t_kernel, since() - t_d2h);
        if (rcs[r] == SBD_OK) { e->gas_np = np; e->gas_nch = nch; e->gas_nk.assign(nk + lo, nk + lo + np); }
        else {
            if (e->d_gas_slots) { (void)hipFree(e->d_gas_slots); e->d_gas_slots = nullptr; }
            if (e->d_gas_lay) { (void)hipFree(e->d_gas_lay); e->d_gas_lay = nullptr; }
        }
    };
    if (nd > 1) {
        std::vector<std::thread> th;
        for (int r = 0; r < nd; ++r) th.emplace_back(run, r);
        for (auto &t : th) t.join();
    } else {
        run(0);
    }
    for (int r = 0; r < nd; ++r)
        if (rcs[r] != SBD_OK) return fail(rcs[r], errs[r]);
    const int64_t token = ++generation;               // (never 0; unique in the process: a token of another fleet cannot match)
    for (int r = 0; r < nd; ++r) f->eng[r]->gas_token = token;
    if (lay_token) *lay_token = token;
    return SBD_OK;
}

int sbd_fleet_gas_terms(sbd_fleet *f, const sbd_gas_model *g, int32_t npoint, const double *wl, const double *lay,
                        int32_t nch, int32_t *nk, double *wt, int32_t *failed, double *dtaug_out, int64_t *lay_token)
{
    if (!lay) return fail(SBD_E_INVALID, "null argument");
    return fleet_point_terms_impl(f, g, nullptr, npoint, wl, lay, nch, nk, wt, failed, dtaug_out, nullptr, lay_token);
}

int sbd_fleet_point_terms(sbd_fleet *f, const sbd_gas_model *g, const sbd_scat_model *sm, int32_t npoint, const double *wl,
                          int32_t nch, int32_t *nk, double *wt, int32_t *failed, double *dtaug_out, double *lay_out, int64_t *lay_token)
{
    if (!sm) return fail(SBD_E_INVALID, "null scatter model");
    return fleet_point_terms_impl(f, g, sm, npoint, wl, nullptr, nch, nk, wt, failed, dtaug_out, lay_out, lay_token);
}

// host clock around device i's enqueue in the last sbd_fleet_solve_host (seconds since that call began), and how
// many of the caller's arrays were page-locked for it: introspection for the multi-device tests
int sbd_fleet_last_enqueue(const sbd_fleet *f, int32_t i, double *t_begin, double *t_end, int32_t *npinned)
{
    if (!f || i < 0 || (size_t)(2 * i + 1) >= f->t_enq.size()) return SBD_E_INVALID;
    if (t_begin) *t_begin = f->t_enq[2 * i];
    if (t_end) *t_end = f->t_enq[2 * i + 1];
    if (npinned) *npinned = f->pinned_last;
    return SBD_OK;
}

// page-locked host memory for the batch arrays of a host program (the Fortran host allocates its batch here, so
// that every H2D of sbd_fleet_solve_host is a DMA from the start and nothing has to be registered per call)
int sbd_host_alloc(size_t bytes, void **out)
{
    if (!out) return SBD_E_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return fail(SBD_E_NOMEM, "hipHostMalloc");
    }
    return SBD_OK;
}
void sbd_host_free(void *p) { if (p) (void)hipHostFree(p); }

}  // extern "C"
