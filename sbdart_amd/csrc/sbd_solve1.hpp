// Back-substitution + fluxes for 16 < NSTR <= 32 (the U factor band1_kernel leaves: one row per unknown, 2 NSTR doubles,
// relative to the diagonal): one wave per (work item, azimuth mode), ROW oriented.
//
// Same job as backsolve_kernel (sbd_solve.hpp): SGBSL's second loop (disutil.f:1038-1050), LL(j, lc)
// (disort.f:3624-3633) and, for mode 0, FLUXES (disort.f:1780-2042).  That kernel walks U by columns: blocks of eight
// columns are transposed through an LDS stage, the running right-hand side rides a ring of registers -- 242 of them at
// NSTR 32, two waves per SIMD, a chain of readlane -> divide -> FMA per column, 2.5 TB/s.  Here a row of U is ONE
// coalesced load (lane 63 - d holds U(k, k + d)), rows are fetched eight ahead, the solved unknowns x(k+1 .. k+63) wait in
// the lanes that will meet them (a one-lane shift of the wave per row, wave_shl:1), the row's products are summed on the
// DPP network (row_shr 1, 2, 4, 8, row_bcast15, row_bcast31: the total lands in lane 63, where the diagonal and the
// right-hand side are), lane 63 divides.  No LDS in the solve, ~60 registers.
// (The sums of a row run over the lanes instead of SGBSL's column-by-column updates: the unknowns agree with the
//  column-oriented kernel to rounding, the parity gates are the same.)
#pragma once
#include "sbd_band.hpp"

namespace sbd {

template <int CTRL, int ROWMASK>
SBD_DEVICE double dpp_add_step(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, true);
    return v + __hiloint2double(hi, lo);
}
// sum over the wave, total in lane 63 (the other lanes hold partial sums)
SBD_DEVICE double wave_sum_to_lane63(double v)
{
    v = dpp_add_step<0x111, 0xF>(v);     // row_shr:1
    v = dpp_add_step<0x112, 0xF>(v);     // row_shr:2
    v = dpp_add_step<0x114, 0xF>(v);     // row_shr:4
    v = dpp_add_step<0x118, 0xF>(v);     // row_shr:8   -> lane 15 of every row: the row's sum
    v = dpp_add_step<0x142, 0xA>(v);     // row_bcast15 -> lanes of rows 1, 3 += lane 15 of rows 0, 2
    v = dpp_add_step<0x143, 0xC>(v);     // row_bcast31 -> lanes of rows 2, 3 += lane 31
    return v;
}
SBD_DEVICE double wave_shift_down1(double v)    // lane l <- lane l + 1, lane 63 <- 0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xF, 0xF, true);   // wave_shl:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int NN>
__global__ void __launch_bounds__(64) backsolve1_kernel(Params P)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];   // FLUXES' staging only: [16][n] twice
    const int lane = threadIdx.x;
    const int nmode = P.nmode;
    // (blocks in mode-major order, see sbd_band1.hpp)
    const int mazim = (int)(blockIdx.x / (unsigned)P.nslot);
    const int slot = (int)(blockIdx.x % (unsigned)P.nslot);
    if (mazim >= nmode) return;
    const long long ms = (long long)slot * nmode + mazim;
    constexpr int n = 2 * NN, nn = NN, UW = u_width(n);
    static_assert(UW <= 64, "backsolve1_kernel: a row of U must fit the wave");
    const int L = P.L;
    const int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const int st0 = svi[SBD_SVI_STATUS];
    const double fbeam = P.fbeam[slot];
    // same early exits as the LU kernel (which zeroed the fluxes of a dead item)
    if ((st0 & (0x20 | 0x10 | 0x08)) != 0) return;
    if (mazim > svi[SBD_SVI_NAZ]) return;
    const int nlev = P.nlev;
    double *flux = P.flux + (size_t)slot * SBD_NFLUX_ * nlev;
    const int ncut = svi[SBD_SVI_NCUT];
    const bool lyrcut = svi[SBD_SVI_LYRCUT] != 0;
    const SV o(L);
    const double *sv = P.sv + (size_t)slot * P.sv_stride;
    const double *taucpr = sv + o.taucpr();
    const double umu0 = P.umu0;
    const double *cmu = P.t.cmu, *cwt = P.t.cwt;
    const bool beam = fbeam > 0.0;
    const double *yv = P.yv + (size_t)ms * L * n;
    const double *gc = P.gc + (size_t)ms * L * n * n;
    const double *kk = P.kk + (size_t)ms * L * n;
    const double *zz = P.zz + (size_t)ms * L * n;
    const double *zp0 = P.zp0 + (size_t)(ms - mazim) * L * n;
    const double *zp1 = P.zp1 + (size_t)(ms - mazim) * L * n;
    const double *ufac = P.ufac + (size_t)ms * (size_t)(L * n) * UW;
    double *ll = P.ll + (size_t)ms * L * n;
    const int N = ncut * n;
#define GC(i, j, lc) gc[((size_t)((lc) - 1) * n + ((i) - 1)) * n + ((j) - 1)]
#define KK(i, lc) kk[((lc) - 1) * n + ((i) - 1)]
#define ZZ(i, lc) zz[((lc) - 1) * n + ((i) - 1)]
#define ZP0(i, lc) zp0[((lc) - 1) * n + ((i) - 1)]
#define ZP1(i, lc) zp1[((lc) - 1) * n + ((i) - 1)]

    // ---- back-substitution by rows.  Row k (1-based) of layer lc = (k-1)/n + 1, J = (k-1) % n: U(k, k + d) for
    //      d = 0 .. 2n-1-J (the columns of x_lc from the diagonal on, then all of x_lc+1; the last layer has no x_lc+1:
    //      d <= n-1-J), at ufac[(k-1) UW + d] (sbd_band1.hpp); the words beyond were never written ----
    {
        const int d = 63 - lane;                  // this lane's offset from the diagonal
        const bool inrow = d < UW;
        constexpr int D = 8;                      // rows in flight
        double cu[D], cy[D], nu[D], ny[D];
        auto fetch = [&](const int k0, double (&u)[D], double (&y)[D]) {     // rows k0, k0-1, .. k0-D+1
#pragma unroll
            for (int t = 0; t < D; ++t) {
                const int kr = (k0 - t >= 1) ? k0 - t : 1;                   // (below the first row: a valid address)
                u[t] = inrow ? ufac[(size_t)(kr - 1) * UW + d] : 0.0;
                y[t] = yv[kr - 1];
            }
        };
        double xw = 0.0;                          // x(k + d) of the row in hand, 0 where there is none yet
        int J = n - 1;
        bool tail = true;                         // rows of the last layer
        fetch(N, cu, cy);
        for (int k = N; k >= 1; k -= D) {
            fetch(k - D, nu, ny);
#pragma unroll
            for (int t = 0; t < D; ++t) {
                if (k - t >= 1) {
                    const int dmax = (tail ? n - 1 : 2 * n - 1) - J;
                    const double u = cu[t];
                    const double p = (d >= 1 && d <= dmax) ? u * xw : 0.0;
                    const double s = wave_sum_to_lane63(p);
                    // b(k) / U(k,k) in lane 63: refined reciprocal, one residual correction (as sbd_solve.hpp)
                    double r = __builtin_amdgcn_rcp(u);
                    r = r * (2.0 - u * r);
                    r = r * (2.0 - u * r);
                    const double bk = cy[t] - s;
                    const double q0 = bk * r;
                    const double xk = q0 + (bk - q0 * u) * r;
                    xw = (lane == 63) ? xk : xw;
                    if (J == 0) {                 // a layer is complete: x of its n rows sit in lanes 63 .. 64-n
                        if (d < n) ll[(k - t - 1) + d] = xw;
                        J = n - 1;
                        tail = false;
                    } else --J;
                    xw = wave_shift_down1(xw);
                }
            }
#pragma unroll
            for (int t = 0; t < D; ++t) { cu[t] = nu[t]; cy[t] = ny[t]; }
        }
    }

    // ---- FLUXES (mode 0) ----
    if (mazim != 0) return;
    __threadfence_block();                        // LL is read back by other lanes of this wave
    {
        double *win = smem;
        const double *b = ll;
        const int32_t *layru = svi + SBD_SVI_LAYRU;
        const double *utau = sv + o.utau(), *utaupr = sv + o.utaupr(), *ssalbv = sv + o.ssalb();
        const double *xr0 = sv + o.xr0(), *xr1 = sv + o.xr1();
        double *efac = win;                 // [16][n]
        double *u0c = win + 16 * n;         // [16][n]
        const double pi = P.pi;
        for (int lev0 = 0; lev0 < nlev; lev0 += 16) {
            const int nb = (nlev - lev0 < 16) ? nlev - lev0 : 16;
            wave_lds_sync();
            // E(jq, lev) = exp(-KK(jq,lyu) * (utaupr - taucpr(lyu or lyu-1)))
            for (int e = lane; e < nb * n; e += 64) {
                const int li = e / n, jq = e % n + 1;
                const int lev = P.all_levels ? lev0 + li : P.t.level_out[lev0 + li];
                const int lyu = layru[lev];
                double val = 0.0;
                if (!(lyrcut && lyu > ncut)) {
                    const double up = utaupr[lev];
                    const double ref = (jq <= nn) ? taucpr[lyu] : taucpr[lyu - 1];
                    val = exp(-KK(jq, lyu) * (up - ref));
                }
                efac[li * n + jq - 1] = val;
            }
            wave_lds_sync();
            for (int e = lane; e < nb * n; e += 64) {
                const int li = e / n, iq = e % n + 1;
                const int lev = P.all_levels ? lev0 + li : P.t.level_out[lev0 + li];
                const int lyu = layru[lev];
                double val = 0.0;
                if (!(lyrcut && lyu > ncut)) {
                    double zint = 0.0;
                    const double *grow = &GC(iq, 1, lyu);
                    const double *llv = b + (lyu - 1) * n;
#pragma unroll 4
                    for (int jq = 1; jq <= n; ++jq) zint = zint + grow[jq - 1] * llv[jq - 1] * efac[li * n + jq - 1];
                    val = zint;
                    if (beam) val = zint + ZZ(iq, lyu) * exp(-utaupr[lev] / umu0);
                    val = val + ZP0(iq, lyu) + ZP1(iq, lyu) * utaupr[lev];
                }
                u0c[li * n + iq - 1] = val;
            }
            wave_lds_sync();
            if (lane < nb) {   // one lane per level: sums in the reference's order
                const int li = lane;
                const int lev = P.all_levels ? lev0 + li : P.t.level_out[lev0 + li];
                const int lyu = layru[lev];
                double rfldir = 0.0, rfldn = 0.0, flup = 0.0, dfdt = 0.0, uavg = 0.0;
                if (!(lyrcut && lyu > ncut)) {
                    double dirint = 0.0, fldir = 0.0, fldn = 0.0;
                    if (beam) {
                        const double fact = exp(-utaupr[lev] / umu0);
                        dirint = fbeam * fact;
                        fldir = umu0 * (fbeam * fact);
                        rfldir = umu0 * fbeam * exp(-utau[lev] / umu0);
                    }
#pragma unroll 2
                    for (int iq = 1; iq <= nn; ++iq) {
                        const double u = u0c[li * n + iq - 1];
                        uavg = uavg + cwt[nn - iq] * u;
                        fldn = fldn + cwt[nn - iq] * cmu[nn - iq] * u;
                    }
#pragma unroll 2
                    for (int iq = nn + 1; iq <= n; ++iq) {
                        const double u = u0c[li * n + iq - 1];
                        uavg = uavg + cwt[iq - nn - 1] * u;
                        flup = flup + cwt[iq - nn - 1] * cmu[iq - nn - 1] * u;
                    }
                    flup = 2.0 * pi * flup;
                    fldn = 2.0 * pi * fldn;
                    const double fdntot = fldn + fldir;
                    rfldn = fdntot - rfldir;
                    uavg = (2.0 * pi * uavg + dirint) / (4.0 * pi);
                    const double plsorc = xr0[lyu - 1] + xr1[lyu - 1] * utaupr[lev];
                    dfdt = (1.0 - ssalbv[lyu - 1]) * 4.0 * pi * (uavg - plsorc);
                }
                const int ol = lev0 + li;
                flux[0 * nlev + ol] = rfldir;
                flux[1 * nlev + ol] = rfldn;
                flux[2 * nlev + ol] = flup;
                flux[3 * nlev + ol] = dfdt;
                flux[4 * nlev + ol] = uavg;
            }
        }
    }
#undef GC
#undef KK
#undef ZZ
#undef ZP0
#undef ZP1
}

}  // namespace sbd
