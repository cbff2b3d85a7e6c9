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

#include <cstdlib>
#include <vector>

namespace sbd {


// ---------------------------------------------------------------------------------------------------------------------
// SGBCO once more, for the 64 lanes of a wave: refband::sgbco (sbd_refband.hpp) serves one lane, and a band system of
// NSTR 32 x 50 layers costs it a second.  Every operation here is the one of the serial code, on the same operands, in
// the same association -- the factors and the estimate are the serial code's bit for bit (the fixtures demand it, and
// SBD_RCOND_SERIAL=1 runs the serial code for the comparison): element-wise work (SSCAL, SAXPY, the zeroing of fill-in,
// the products of SDOT, the terms of SASUM) is spread over the lanes; every SUM is formed by all lanes alike, term by
// term in the serial order, from values fetched lane by lane (v_readlane); ISAMAX is a maximum over the lanes and the
// first lane that holds it.  abd lives in global memory (a fence + wave barrier where lanes read what others wrote),
// z in LDS.
// ---------------------------------------------------------------------------------------------------------------------
namespace wave {

SBD_DEVICE void gsync() { __threadfence_block(); __builtin_amdgcn_wave_barrier(); }
SBD_DEVICE double lane_val(double v, int r) { return __shfl(v, r); }        // r uniform: v_readlane

// s + t(0) + t(1) + ... + t(cnt-1), t(r) = lane r's `term`, added in that order (cnt <= 64, uniform)
SBD_DEVICE double add_in_order(double s, double term, int cnt)
{
#pragma clang fp contract(off)
    for (int r = 0; r < cnt; ++r) s = s + lane_val(term, r);
    return s;
}

// SASUM(n, z) with z in LDS: left to right
SBD_DEVICE double sasum_lds(int n, const double *z, int lane)
{
    double s = 0.0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int cnt = (n - i0 < 64) ? n - i0 : 64;
        const double t = (lane < cnt) ? fabs(z[i0 + lane]) : 0.0;
        s = add_in_order(s, t, cnt);
    }
    return s;
}
SBD_DEVICE void sscal_lds(int n, double sa, double *z, int lane)
{
    for (int i = lane; i < n; i += 64) z[i] = sa * z[i];
    wave_lds_sync();
}

SBD_DEVICE double sgbco(double *abd, int lda, int n, int ml, int mu, int32_t *ipvt, double *z, int lane)
{
#pragma clang fp contract(off)
#define ABD(i, j) abd[((size_t)(j) - 1) * lda + ((i) - 1)]
    const int m = ml + mu + 1;
    // ---- ANORM: the largest SASUM of a column; a lane sums its columns top to bottom ----
    double anorm = 0.0;
    for (int j = lane + 1; j <= n; j += 64) {
        const int inc = (j - 1 < mu) ? j - 1 : mu;                   // increments of l before column j: columns j' <= mu
        const int dec = (j - (n - ml) > 0) ? j - (n - ml) : 0;       // decrements: columns j' >= n - ml
        const int l = ml + 1 + inc - dec;
        const int is = (ml + 1 + mu - (j - 1) > ml + 1) ? ml + 1 + mu - (j - 1) : ml + 1;
        double sj = 0.0;
        for (int i = 0; i < l; ++i) sj = sj + fabs(ABD(is + i, j));
        if (sj > anorm) anorm = sj;
    }
    {
        const double mine = anorm;                                    // (this lane's columns)
        anorm = lane_val(mine, 0);
        for (int r = 1; r < 64; ++r) { const double o = lane_val(mine, r); if (o > anorm) anorm = o; }
    }
    // (the serial code takes the columns in order with `if (s > anorm)`: a maximum that skips NaN, like this one)

    // ---- SGBFA ----
    {
        const int j0 = mu + 2, j1 = ((n < m) ? n : m) - 1;
        for (int jz = j0; jz <= j1; ++jz) {
            const int i0 = m + 1 - jz;
            for (int i = i0 + lane; i <= ml; i += 64) ABD(i, jz) = 0.0;
        }
        int jz = j1 + 1;                                              // the column iteration k = 1 zeroes first
        if (jz <= n)
            for (int i = 1 + lane; i <= ml; i += 64) ABD(i, jz) = 0.0;
        gsync();
        int ju = 0;
        for (int k = 1; k <= n - 1; ++k) {
            const int kp1 = k + 1;
            const int lm = (ml < n - k) ? ml : n - k;
            // ISAMAX over ABD(m .. m+lm, k): lane r holds element r
            const double v = (lane <= lm) ? ABD(m + lane, k) : 0.0;
            const double xm = fabs(v);
            double key = (lane <= lm && xm > 0.0) ? xm : -1.0;       // (zero and NaN never win: `smax < xmag` is false for them)
            double mx = key;
            for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(mx, d); mx = (o > mx) ? o : mx; }
            const unsigned long long hit = __ballot(key == mx && mx > 0.0);
            const int idx = hit ? (int)__builtin_ctzll(hit) + 1 : 0;  // 1-based, 0 when nothing is > 0
            const int l = idx + m - 1;
            const int ipv = l + k - m;
            if (lane == 0) ipvt[k - 1] = ipv;
            const double piv = idx ? lane_val(v, idx - 1) : ABD(m - 1, k);
            if (piv != 0.0) {                                         // (a zero pivot: INFO = k, nothing is eliminated)
                const double diag = lane_val(v, 0);
                const double t = -1.0 / piv;
                // the interchange, then SSCAL of the lm elements below the diagonal: lane r <-> row m + r
                double x = 0.0;
                if (lane >= 1 && lane <= lm) {
                    const double val = (m + lane == l) ? diag : v;
                    x = t * val;
                    ABD(m + lane, k) = x;
                }
                if (lane == 0 && l != m) ABD(m, k) = piv;
                { const int c = mu + ipv; ju = (ju > c) ? ju : c; ju = (ju < n) ? ju : n; }
                int mm = m, ll = l;
                for (int j = kp1; j <= ju; ++j) {
                    ll = ll - 1;
                    mm = mm - 1;
                    const double tj = ABD(ll, j);
                    double am = 0.0;
                    if (ll != mm) {
                        am = ABD(mm, j);
                        if (lane == 0) ABD(mm, j) = tj;
                    }
                    if (lane >= 1 && lane <= lm) {                    // SAXPY(lm, tj, x, ABD(mm+1.., j)); row ll takes the interchanged value
                        double y = (ll != mm && mm + lane == ll) ? am : ABD(mm + lane, j);
                        if (tj != 0.0) y = y + tj * x;               // (SAXPY leaves y alone when sa == 0)
                        if (tj != 0.0 || (ll != mm && mm + lane == ll)) ABD(mm + lane, j) = y;
                    }
                }
            }
            jz = jz + 1;                                              // the fill-in column of the NEXT iteration
            if (jz <= n)
                for (int i = 1 + lane; i <= ml; i += 64) ABD(i, jz) = 0.0;
            gsync();
        }
        if (lane == 0) ipvt[n - 1] = n;
        gsync();
    }

    // ---- the estimate: z in LDS ----
    double ek = 1.0, s, sm, t, wk, wkm, ynorm;
    for (int j = lane; j < n; j += 64) z[j] = 0.0;
    wave_lds_sync();
    int ju = 0;
    for (int k = 1; k <= n; ++k) {                                    // solve trans(U) w = e
        const double dk = ABD(m, k);
        double zk = z[k - 1];
        if (zk != 0.0) ek = refband::dsign(ek, -zk);
        if (fabs(ek - zk) > fabs(dk)) {
            s = fabs(dk) / fabs(ek - zk);
            sscal_lds(n, s, z, lane);
            ek = s * ek;
            zk = z[k - 1];
        }
        wk = ek - zk;
        wkm = -ek - zk;
        s = fabs(wk);
        sm = fabs(wkm);
        if (dk != 0.0) { wk = wk / dk; wkm = wkm / dk; }
        else { wk = 1.0; wkm = 1.0; }
        const int kp1 = k + 1;
        { const int c = mu + ipvt[k - 1]; ju = (ju > c) ? ju : c; ju = (ju < n) ? ju : n; }
        if (kp1 <= ju) {
            const int cnt_all = ju - kp1 + 1;
            for (int r0 = 0; r0 < cnt_all; r0 += 64) {               // j = kp1 + r0 + lane, element ABD(m - (j - k), j)
                const int cnt = (cnt_all - r0 < 64) ? cnt_all - r0 : 64;
                const int j = kp1 + r0 + lane;
                double t1 = 0.0, t2 = 0.0;
                if (lane < cnt) {
                    const double a = ABD(m - (j - k), j);
                    const double zj = z[j - 1];
                    t1 = fabs(zj + wkm * a);
                    const double zn = zj + wk * a;
                    z[j - 1] = zn;
                    t2 = fabs(zn);
                }
                // sm = sm + |z(j) + wkm a|, s = s + |z(j)| -- j ascending
                for (int r = 0; r < cnt; ++r) { sm = sm + lane_val(t1, r); s = s + lane_val(t2, r); }
            }
            wave_lds_sync();
            if (s < sm) {
                t = wkm - wk;
                wk = wkm;
                for (int r0 = 0; r0 < cnt_all; r0 += 64) {
                    const int j = kp1 + r0 + lane;
                    if (r0 + lane < cnt_all) z[j - 1] = z[j - 1] + t * ABD(m - (j - k), j);
                }
            }
        }
        if (lane == 0) z[k - 1] = wk;
        wave_lds_sync();
    }
    s = 1.0 / sasum_lds(n, z, lane);
    sscal_lds(n, s, z, lane);
    for (int kb = 1; kb <= n; ++kb) {                                 // solve trans(L) y = w
        const int k = n + 1 - kb;
        const int lm = (ml < n - k) ? ml : n - k;
        if (k < n) {
            const double p = (lane < lm) ? ABD(m + 1 + lane, k) * z[k + lane] : 0.0;
            const double dot = add_in_order(0.0, p, lm);
            if (lane == 0) z[k - 1] = z[k - 1] + dot;
            wave_lds_sync();
        }
        const double zk = z[k - 1];
        if (fabs(zk) > 1.0) { s = 1.0 / fabs(zk); sscal_lds(n, s, z, lane); }
        const int lp = ipvt[k - 1];
        if (lane == 0) { const double a = z[lp - 1]; z[lp - 1] = z[k - 1]; z[k - 1] = a; }
        wave_lds_sync();
    }
    s = 1.0 / sasum_lds(n, z, lane);
    sscal_lds(n, s, z, lane);
    ynorm = 1.0;
    for (int k = 1; k <= n; ++k) {                                    // solve L v = y
        const int lp = ipvt[k - 1];
        t = z[lp - 1];
        wave_lds_sync();
        if (lane == 0) { z[lp - 1] = z[k - 1]; z[k - 1] = t; }
        wave_lds_sync();
        const int lm = (ml < n - k) ? ml : n - k;
        if (k < n && t != 0.0) {
            if (lane < lm) z[k + lane] = z[k + lane] + t * ABD(m + 1 + lane, k);
            wave_lds_sync();
        }
        const double zk = z[k - 1];
        if (fabs(zk) > 1.0) { s = 1.0 / fabs(zk); sscal_lds(n, s, z, lane); ynorm = s * ynorm; }
    }
    s = 1.0 / sasum_lds(n, z, lane);
    sscal_lds(n, s, z, lane);
    ynorm = s * ynorm;
    for (int kb = 1; kb <= n; ++kb) {                                 // solve U z = v
        const int k = n + 1 - kb;
        const double dk = ABD(m, k);
        double zk = z[k - 1];
        if (fabs(zk) > fabs(dk)) {
            s = fabs(dk) / fabs(zk);
            sscal_lds(n, s, z, lane);
            ynorm = s * ynorm;
            zk = z[k - 1];
        }
        if (dk != 0.0) zk = zk / dk;
        if (dk == 0.0) zk = 1.0;
        wave_lds_sync();
        if (lane == 0) z[k - 1] = zk;
        const int lm = ((k < m) ? k : m) - 1;
        const int la = m - lm, lz = k - lm;
        t = -zk;
        if (t != 0.0)
            for (int r = lane; r < lm; r += 64) z[lz - 1 + r] = z[lz - 1 + r] + t * ABD(la + r, k);
        wave_lds_sync();
    }
    s = 1.0 / sasum_lds(n, z, lane);
    sscal_lds(n, s, z, lane);
    ynorm = s * ynorm;
    return (anorm != 0.0) ? ynorm / anorm : 0.0;
#undef ABD
}

}  // namespace wave

// grid: any number of single-wave blocks; block b serves list entries b, b + gridDim.x, ...  scratch: per block
// refband::SystemScratch(n, L).total + 64 * refband::layer_work_doubles(n) doubles.
// rcond_dbg (tests; may be NULL): [nslot * nmode] the estimate of every system served (untouched otherwise).
__global__ void __launch_bounds__(64, 4) band_rcond_kernel(Params P, double *scratch, size_t stride, double *rcond_dbg, int serial)
{
    extern __shared__ __attribute__((aligned(16))) double zlds[];      // SGBCO's z of the wave form: n NLYR doubles
    const int lane = threadIdx.x;
    const int L = P.L, n = P.n, nmode = P.nmode, nmom = P.nmom;
    const int count = P.rclist[0];
    if (P.rchint && blockIdx.x == 0 && lane == 0) *P.rchint = count;     // (the host sizes the next pass's grid by it)
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
        if (!s_bad) {
            const bool brdf = P.ibdrf != 0;
            const double *bdr = brdf ? surf_bdr(P, surf_index(P, slot, mazim)) : nullptr;
            double rcond = 0.0;
            if (serial) {                                                            // (SBD_RCOND_SERIAL=1: the one-lane code, for comparison)
                if (lane == 0) rcond = refband::band_rcond_from_layers(n, L, ncut, lyrcut, !brdf, mazim, P.albedo[slot], bdr, P.t.cmu, P.t.cwt, s);
            } else {
                // TAUCPR and SETMTX by lane 0 (a few thousand stores), the factorisation and the estimate by the wave
                if (lane == 0) {
#pragma clang fp contract(off)
                    double *taucpr = s + o.taucpr;
                    taucpr[0] = 0.0;
                    for (int lc = 1; lc <= L; ++lc) taucpr[lc] = taucpr[lc - 1] + s[o.dtaucp + lc - 1];
                    refband::setmtx(n, ncut, lyrcut, !brdf, (mazim == 0) ? 1.0 : 0.0, P.albedo[slot], bdr, P.t.cmu, P.t.cwt, s + o.gc, s + o.kk,
                                    s + o.dtaucp, taucpr, s + o.cband, o.lda, s + o.wk);
                }
                __threadfence_block();
                __syncthreads();
                const int ncd = 3 * (n / 2) - 1;
                rcond = wave::sgbco(s + o.cband, o.lda, n * ncut, ncd, ncd, (int32_t *)(s + o.ipvt), zlds, lane);
            }
            if (lane == 0) {
                if (rcond_dbg) rcond_dbg[ms] = rcond;
                if (1.0 + rcond == 1.0) {                                            // disort.f:3609
                    atomicOr(&svi[SBD_SVI_STATUS], 0x01);
                    atomicOr(&P.status[slot], 0x01);
                }
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
    static const int serial = getenv("SBD_RCOND_SERIAL") ? atoi(getenv("SBD_RCOND_SERIAL")) : 0;
    hipLaunchKernelGGL(band_rcond_kernel, dim3(grid), dim3(64), sizeof(double) * (size_t)P.n * P.L, st, P, scratch, stride, rcond_dbg, serial);
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
