// Back-substitution + fluxes for NSTR <= 16, FOUR systems per wave: the partner of sbd_band4.hpp.
//
// Same job as sbd_solve.hpp (SGBSL's second loop, disutil.f:1038-1050, on the U factor and the
// forward-eliminated right-hand side; LL(j, lc), disort.f:3624-3633; for mode 0 FLUXES at the
// requested levels, disort.f:1780-2042) on sbd_band4.hpp's layer-block layout of U: row J of
// layer lc holds U(k, .) for the columns of x_lc in words 0..n-1 and of x_lc+1 in words n..2n-1.
//
// A system owns a row of 16 lanes, lane q <-> unknown q of the current layer.  Layers run from
// the last to the first; inside a layer rows J = n-1 .. 0:
//     x_J = ( y_J - sum_c U1(J,c) x_lc+1(c) - sum_{c>J} U0(J,c) x_lc(c) ) / U0(J,J)
// -- each lane multiplies the words it loaded by the unknowns it holds, the 16 products are summed
// by a DPP butterfly (quad_perm, row_half_mirror, row_mirror), lane J divides.  No LDS.  The two
// row words of the NEXT layer are fetched into a row's registers as soon as the row is used
// (rows are consumed in the same order in every layer), so a layer's worth of elimination covers
// the HBM latency; nothing but U (and B) streams through, once, in full lines.
// Fluxes: when the layer of an output level has been solved, U0C = GC (LL * exp) + particular
// solutions is formed with GC's row in registers and LL broadcast by the DP-ALU DPP FMA.
#pragma once
#include "sbd_common.hpp"
#include "sbd_band.hpp"
#include "sbd_band4.hpp"

namespace sbd {

template <int NN>
__global__ void __launch_bounds__(64, 2) backsolve4_kernel(Params P)
{
    constexpr int n = 2 * NN, nn = NN, UW = u_width(n);
    static_assert(n <= 16, "backsolve4_kernel: a layer's unknowns must fit a row of 16 lanes");
    const int lane = threadIdx.x, q = lane & 15;
    const int nmode = P.nmode, L = P.L;
    // (blocks in mode-major order: the items of mode 0 first.  Item-major, the modes an item does not need -- no beam, no
    //  moment left, SBD_SVI_NAZ -- left their live blocks on a few of the eight XCDs: block b goes to XCD b mod 8)
    const long long bid = (long long)blockIdx.x * 4 + (lane >> 4);
    if (bid >= (long long)P.nslot * nmode) return;
    const int mazim = (int)((unsigned)bid / (unsigned)P.nslot);
    const int slot = (int)((unsigned)bid % (unsigned)P.nslot);
    const long long ms = (long long)slot * nmode + mazim;
    const int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const int st0 = svi[SBD_SVI_STATUS];
    const double fbeam = P.fbeam[slot];
    // same early exits as the LU kernel (which zeroed the fluxes of a dead item)
    if ((st0 & (0x20 | 0x10 | 0x08)) != 0) return;
    if (mazim > svi[SBD_SVI_NAZ]) return;
    const int nlev = P.nlev;
    const int ncut = svi[SBD_SVI_NCUT];
    const bool lyrcut = svi[SBD_SVI_LYRCUT] != 0;
    const SV o(L);
    const double *sv = P.sv + (size_t)slot * P.sv_stride;
    const double *taucpr = sv + o.taucpr();
    const double umu0 = P.umu0;
    const bool beam = fbeam > 0.0;
    const bool col = q < n;
    const int cq = col ? q : 0;
    const double *yv = P.yv + (size_t)ms * L * n;
    const double *gc = P.gc + (size_t)ms * L * n * n;
    const double *kk = P.kk + (size_t)ms * L * n;
    const double *zz = P.zz + (size_t)ms * L * n;
    const double *zp0 = P.zp0 + (size_t)(ms - mazim) * L * n;
    const double *zp1 = P.zp1 + (size_t)(ms - mazim) * L * n;
    const double *ufac = P.ufac + (size_t)ms * (size_t)(L * n) * UW;
    double *ll = P.ll + (size_t)ms * L * n;
    double *flux = P.flux + (size_t)slot * SBD_NFLUX_ * nlev;
    const int32_t *layru = svi + SBD_SVI_LAYRU;
    const double *utau = sv + o.utau(), *utaupr = sv + o.utaupr(), *ssalbv = sv + o.ssalb();
    const double *xr0 = sv + o.xr0(), *xr1 = sv + o.xr1();
    // quadrature weights of this lane's stream: positions iq <= nn look down (-mu), the others up
    // (FLUXES pairs position iq with CWT/CMU(nn+1-iq) resp. (iq-nn), disort.f:1964-1966, 1992-1994)
    const int iqw = col ? ((q < nn) ? nn - 1 - q : q - nn) : 0;
    const double wq = col ? P.t.cwt[iqw] : 0.0, wmq = col ? P.t.cwt[iqw] * P.t.cmu[iqw] : 0.0;
    const double pi = P.pi;

    // rows of the layer being solved: u0[J] = U0(J, q), u1[J] = U1(J, q)
    double u0[n], u1[n];
    // (16-column layers: sbd_band4.hpp's 28-line block -- U1's rows, U0's rows 0..7, the live halves of U0's
    //  rows 8..15 two to a line; otherwise rows of 2n words)
    auto load_row = [&](int lc, auto jj, double &w0, double &w1) {
        constexpr int J = decltype(jj)::value;
        const double *blk = ufac + (size_t)(lc - 1) * n * UW;
        if constexpr (n == 16) {
            w1 = blk[J * 16 + cq];
            if constexpr (J < 8) w0 = blk[(16 + J) * 16 + cq];
            else if (q >= 8) w0 = blk[(24 + (J - 8) / 2) * 16 + (J & 1) * 8 + (cq - 8)];   // (lanes 0..7: columns long finished)
        } else {
            const double *p = blk + (size_t)J * UW + cq;
            w0 = p[0];                              // (raw words: lanes left of the diagonal hold multipliers and
            w1 = p[n];                              //  are masked where they are used, not where they are loaded)
        }
    };
#pragma unroll
    for (int J = 0; J < n; ++J) { u0[J] = 1.0; u1[J] = 0.0; }
    static_for<n>([&](auto jj) { load_row(ncut, jj, u0[decltype(jj)::value], u1[decltype(jj)::value]); });
    double yq = yv[(ncut - 1) * n + cq];
    double xn = 0.0;                                 // x of the layer below (none below the last)
    int ol_scan = nlev - 1;                          // all levels: the lowest level not yet evaluated
    for (int lc = ncut; lc >= 1; --lc) {
        const int lcp = (lc > 1) ? lc - 1 : 1;       // the layer whose rows are fetched meanwhile
        double xq = 0.0;
        double yqn = 0.0;
        static_for<n>([&](auto jj) {
            constexpr int J = n - 1 - decltype(jj)::value;
            // -1/pivot for lane J (v_rcp + two Newton steps); every lane computes one, lane J's is used
            const double d = u0[J];
            double r = __builtin_amdgcn_rcp(d);
            r = r * (2.0 - d * r);
            r = r * (2.0 - d * r);
            double p = u1[J] * xn;
            p = (q > J) ? p + u0[J] * xq : p;
            if (n < 16) p = col ? p : 0.0;
            const double tot = row_sum16(p);
            // b(k)/U(k,k): product with the refined reciprocal plus one residual correction
            const double bk = yq - tot;
            const double q0 = bk * r;
            const double xk = q0 + (bk - q0 * d) * r;
            xq = (q == J) ? xk : xq;
            // this row is done: fetch the same row of the layer above into its registers
            load_row(lcp, std::integral_constant<int, J>{}, u0[J], u1[J]);   // (no use of the values here: the loads stay in flight)
            if constexpr (J == n - 1) yqn = yv[(lcp - 1) * n + cq];
        });
        if (col) ll[(lc - 1) * n + q] = xq;           // LL(j, lc) = B((lc-1)*n + j) (disort.f:3624-3633)
        // ---- FLUXES (mode 0) at the output levels that lie in this layer ----
        if (mazim == 0) {
            // (all levels: LAYRU does not decrease with the level, so the levels of layer lc are the next ones down from
            //  where layer lc+1 stopped -- scanning every level for every layer cost as much as the solve itself:
            //  NLYR x (NLYR+1) tests per system)
            int ol_lo = 0, ol_hi = nlev - 1;
            if (P.all_levels) {
                while (ol_scan >= 0 && layru[ol_scan] > lc) --ol_scan;
                ol_hi = ol_scan;
                while (ol_scan >= 0 && layru[ol_scan] == lc) --ol_scan;
                ol_lo = ol_scan + 1;
            }
            for (int ol = ol_lo; ol <= ol_hi; ++ol) {
                const int lev = P.all_levels ? ol : P.t.level_out[ol];
                if (layru[lev] != lc) continue;
                // U0C(iq) = sum_j GC(iq,j) LL(j) E(j) + ZZ(iq) e^{-tau'/mu0} + ZPLK0 + ZPLK1 tau' (disort.f:1945-1960),
                // E(j) = exp(-KK(j) (utaupr - taucpr(lc or lc-1)))
                const double up = utaupr[lev];
                const double ref = (q < nn) ? taucpr[lc] : taucpr[lc - 1];
                const double ev = col ? xq * exp(-kk[(lc - 1) * n + cq] * (up - ref)) : 0.0;   // lane j: LL(j) E(j)
                double u0c = 0.0;
                const double *grow = gc + ((size_t)(lc - 1) * n + cq) * n;                      // GC(iq, ., lc), iq = q+1
                static_for<n>([&](auto jj) {
                    constexpr int j = decltype(jj)::value;
                    u0c = fmac_lane_bcast<j>(u0c, ev, grow[j]);
                });
                if (beam) u0c = u0c + zz[(lc - 1) * n + cq] * exp(-up / umu0);
                u0c = u0c + zp0[(lc - 1) * n + cq] + zp1[(lc - 1) * n + cq] * up;
                if (!col) u0c = 0.0;
                const double uavg_s = row_sum16(wq * u0c);
                const double fldn_s = row_sum16((q < nn) ? wmq * u0c : 0.0);
                const double flup_s = row_sum16((q >= nn) ? wmq * u0c : 0.0);
                double rfldir = 0.0, rfldn = 0.0, flup = 0.0, dfdt = 0.0, uavg = 0.0;
                {
                    double dirint = 0.0, fldir = 0.0;
                    if (beam) {
                        const double fact = exp(-up / umu0);
                        dirint = fbeam * fact;
                        fldir = umu0 * (fbeam * fact);
                        rfldir = umu0 * fbeam * exp(-utau[lev] / umu0);
                    }
                    flup = 2.0 * pi * flup_s;
                    const double fldn = 2.0 * pi * fldn_s;
                    const double fdntot = fldn + fldir;
                    rfldn = fdntot - rfldir;
                    uavg = (2.0 * pi * uavg_s + dirint) / (4.0 * pi);
                    const double plsorc = xr0[lc - 1] + xr1[lc - 1] * up;
                    dfdt = (1.0 - ssalbv[lc - 1]) * 4.0 * pi * (uavg - plsorc);
                }
                if (q == 0) {
                    flux[0 * nlev + ol] = rfldir;
                    flux[1 * nlev + ol] = rfldn;
                    flux[2 * nlev + ol] = flup;
                    flux[3 * nlev + ol] = dfdt;
                    flux[4 * nlev + ol] = uavg;
                }
            }
        }
        xn = xq;
        yq = yqn;
    }
    // levels below a cut-off layer (LYRCUT, disort.f:1907-1916) stay zero
    if (mazim == 0 && lyrcut) {
        for (int ol = q; ol < nlev; ol += 16) {
            const int lev = P.all_levels ? ol : P.t.level_out[ol];
            if (layru[lev] > ncut)
                for (int c = 0; c < SBD_NFLUX_; ++c) flux[c * nlev + ol] = 0.0;
        }
    }
}

}  // namespace sbd
