// errmsg 2 on LINPACK's own estimate: the reference's formulation of a flagged boundary-value system (sbd_refband.hpp) as a
// kernel -- one wave per listed system, a lane per layer for the eigenproblems, then SETMTX + SGBCO -- and as plain host
// code behind sbd_band_rcond_host: the SAME source on both sides (the host side is the pin: bit-equal to the oracle's RCOND
// with the host's exp, tests/test_refband_host.py).
#include "../../include/sbdart_amd.h"
#include "sbd_common.hpp"
#include "sbd_surface.hpp"
#include "sbd_launch.hpp"
#include "sbd_hosttables.hpp"
#include "sbd_refband.hpp"

#include <vector>

namespace sbd {

// grid: any number of single-wave blocks; block b serves list entries b, b + gridDim.x, ...  scratch: per block
// refband::SystemScratch(n, L).total + 64 * refband::layer_work_doubles(n) doubles.
// rcond_dbg (tests; may be NULL): [nslot * nmode] the estimate of every system served (untouched otherwise).
__global__ void __launch_bounds__(64) band_rcond_kernel(Params P, double *scratch, size_t stride, double *rcond_dbg)
{
    const int lane = threadIdx.x;
    const int L = P.L, n = P.n, nmode = P.nmode, nmom = P.nmom;
    const int count = P.rclist[0];
    const refband::SystemScratch o(n, L);
    const size_t lw = refband::layer_work_doubles(n);
    double *s = scratch + (size_t)blockIdx.x * stride;
    double *work = s + o.total + (size_t)lane * lw;
    __shared__ int s_bad;
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
        const long long ms = P.rclist[1 + it];
        const int slot = (int)(ms / nmode), mazim = (int)(ms % nmode);
        int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
        const int st0 = svi[SBD_SVI_STATUS];
        if ((st0 & (0x20 | 0x10 | 0x08)) || mazim > svi[SBD_SVI_NAZ]) continue;      // (no system was factored for this entry)
        const int ncut = svi[SBD_SVI_NCUT];
        const bool lyrcut = svi[SBD_SVI_LYRCUT] != 0;
        const double *dtauc = P.dtauc + (size_t)slot * L, *ssalb = P.ssalb + (size_t)slot * L;
        const double *pmom = P.pmom + pmom_item(P, slot) * L * (nmom + 1);
        const double *ylmc = P.t.ylmc + (size_t)mazim * n * (n + 1);
        if (lane == 0) s_bad = 0;
        __syncthreads();
        for (int lc = lane + 1; lc <= L; lc += 64) {
            double w = ssalb[lc - 1];
            if (w == 1.0) w = 1.0 - P.dither;                                        // disort.f:486
            double dt = dtauc[lc - 1];
            if (dt < 0.0) dt = 0.0;                                                  // CHEKIN, disort.f:4944
            const double *pm = pmom + (size_t)(lc - 1) * (nmom + 1);
            if (lc <= ncut) {
                const int ier = refband::soleig_layer(n, mazim, dt, w, pm, nmom, P.t.cmu, P.t.cwt, ylmc, work, s + o.gc + (size_t)(lc - 1) * n * n,
                                                      s + o.kk + (size_t)(lc - 1) * n, s + o.dtaucp + (lc - 1));
                if (ier != 0) s_bad = 1;
            } else {
#pragma clang fp contract(off)
                const double f = (n <= nmom) ? pm[n] : 0.0;
                s[o.dtaucp + (lc - 1)] = (1.0 - f * w) * dt;
            }
        }
        __threadfence_block();
        __syncthreads();
        if (lane == 0 && !s_bad) {
            const bool brdf = P.ibdrf != 0;
            const double *bdr = brdf ? surf_bdr(P, surf_index(P, slot, mazim)) : nullptr;
            const double rcond = refband::band_rcond_from_layers(n, L, ncut, lyrcut, !brdf, mazim, P.albedo[slot], bdr, P.t.cmu, P.t.cwt, s);
            if (rcond_dbg) rcond_dbg[ms] = rcond;
            if (1.0 + rcond == 1.0) {                                                // disort.f:3609
                atomicOr(&svi[SBD_SVI_STATUS], 0x01);
                atomicOr(&P.status[slot], 0x01);
            }
        }
        __syncthreads();
    }
}

size_t band_rcond_scratch_doubles(int n, int L)
{
    const refband::SystemScratch o(n, L);
    return (o.total + 64 * refband::layer_work_doubles(n) + 31) & ~(size_t)31;
}

void launch_band_rcond(unsigned grid, hipStream_t st, const Params &P, double *scratch, size_t stride, double *rcond_dbg)
{
    hipLaunchKernelGGL(band_rcond_kernel, dim3(grid), dim3(64), 0, st, P, scratch, stride, rcond_dbg);
}

}  // namespace sbd

// The same arithmetic on the HOST (no GPU involved): LINPACK's reciprocal condition estimate of the boundary-value system
// of azimuth mode `mazim` for one work item over a Lambertian surface, exactly as SOLVE0 sees it (disort.f:3607).
extern "C" int sbd_band_rcond_host(int32_t nlyr, int32_t nstr, int32_t nmom, int32_t mazim, int32_t plank, double albedo,
                                   const double *dtauc_in, const double *ssalb_in, const double *pmom, double *rcond)
{
    using namespace sbd;
    if (!dtauc_in || !ssalb_in || !pmom || !rcond || nlyr < 1 || nlyr > SBD_MAX_NLYR || nstr < 4 || nstr > SBD_MAX_NSTR || (nstr & 1)
        || mazim < 0 || mazim >= nstr || nmom < 0) return SBD_E_INVALID;
    const int n = nstr, nn = n / 2, L = nlyr;
    std::vector<double> cmu(n), cwt(n);
    hosttab::gauss01(nn, cmu.data(), cwt.data());
    for (int i = 0; i < nn; ++i) { cmu[i + nn] = -cmu[i]; cwt[i + nn] = cwt[i]; }
    std::vector<double> yc((size_t)n * (n + 1), 0.0);
    for (int m = 0; m <= mazim; ++m) {                     // (LEPOLY's recurrence in m needs order m - 1 in place)
        hosttab::legendre_norm(nn, m, n, n - 1, cmu.data(), yc.data());
        double sgn = -1.0;                                 // mirror to -mu (disort.f:611-627)
        for (int l = m; l <= n - 1; ++l) {
            sgn = -sgn;
            for (int iq = nn; iq < n; ++iq) yc[(size_t)iq * (n + 1) + l] = sgn * yc[(size_t)(iq - nn) * (n + 1) + l];
        }
    }
    const double dither = 100.0 * 2.220446049250313e-16;
    std::vector<double> dt(L), w(L);
    double abstau = 0.0;
    int ncut = L;
    for (int lc = 0; lc < L; ++lc) {                       // SETDIS's cut-off (disort.f:2557-2605)
        w[lc] = (ssalb_in[lc] == 1.0) ? 1.0 - dither : ssalb_in[lc];
        dt[lc] = dtauc_in[lc] < 0.0 ? 0.0 : dtauc_in[lc];
        if (abstau < 10.0) ncut = lc + 1;
        abstau = abstau + (1.0 - w[lc]) * dt[lc];
    }
    const bool lyrcut = abstau >= 10.0 && !plank && L > 1;
    if (!lyrcut) ncut = L;
    const refband::SystemScratch o(n, L);
    std::vector<double> s(o.total + refband::layer_work_doubles(n), 0.0);
    double *work = s.data() + o.total;
    for (int lc = 1; lc <= L; ++lc) {
        const double *pm = pmom + (size_t)(lc - 1) * (nmom + 1);
        if (lc <= ncut) {
            const int ier = refband::soleig_layer(n, mazim, dt[lc - 1], w[lc - 1], pm, nmom, cmu.data(), cwt.data(), yc.data(), work,
                                                  s.data() + o.gc + (size_t)(lc - 1) * n * n, s.data() + o.kk + (size_t)(lc - 1) * n,
                                                  s.data() + o.dtaucp + (lc - 1));
            if (ier != 0) return SBD_E_UNSUPPORTED;        // (ASYMTX did not converge: the reference stops, disort.f:3254-3261)
        } else {
            const double f = (n <= nmom) ? pm[n] : 0.0;
            s[o.dtaucp + (lc - 1)] = (1.0 - f * w[lc - 1]) * dt[lc - 1];
        }
    }
    *rcond = refband::band_rcond_from_layers(n, L, ncut, lyrcut, true, mazim, albedo, nullptr, cmu.data(), cwt.data(), s.data());
    return SBD_OK;
}
