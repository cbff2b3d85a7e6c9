// Back-substitution + fluxes: one wave per (work item, azimuth mode) finishes what the LU
// kernel (sbd_band.hpp) started -- SGBSL's second loop (disutil.f:1038-1050) on the U factor
// and the forward-eliminated right-hand side left in HBM, LL(j, lc) (disort.f:3624-3633),
// and, for mode 0, FLUXES at the requested levels (disort.f:1780-2042).  A kernel of its own
// so that its registers (a block of U in flight) do not cost the LU kernel occupancy.
#pragma once
#include "sbd_band.hpp"

namespace sbd {

template <int NN>
__global__ void __launch_bounds__(64) backsolve_kernel(Params P)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int nmode = P.nmode;
    // (blocks in mode-major order: the items of mode 0 first.  Item-major, the modes an item does not need -- no beam, no
    //  moment left, SBD_SVI_NAZ -- left their live blocks on a few of the eight XCDs: block b goes to XCD b mod 8)
    const int mazim = (int)(blockIdx.x / (unsigned)P.nslot);
    const int slot = (int)(blockIdx.x % (unsigned)P.nslot);
    if (mazim >= nmode) return;
    const long long ms = (long long)slot * nmode + mazim;
    constexpr int n = 2 * NN, nn = NN;
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

    const SolveLds lds(n, nn, L);
    constexpr int UW = u_width(n), UB = UW - 1;          // stored width / upper bandwidth of U
    double *win = smem + lds.stage;
    double *b = smem + lds.x;                         // right-hand side -> solution vector
    const double *yv = P.yv + (size_t)ms * L * n;
    const double *gc = P.gc + (size_t)ms * L * n * n;
    const double *kk = P.kk + (size_t)ms * L * n;
    const double *zz = P.zz + (size_t)ms * L * n;
    const double *zp0 = P.zp0 + (size_t)(ms - mazim) * L * n;
    const double *zp1 = P.zp1 + (size_t)(ms - mazim) * L * n;
    const double *ufac = P.ufac + (size_t)ms * (size_t)(L * n) * UW;
    const int N = ncut * n;
#define GC(i, j, lc) gc[((size_t)((lc) - 1) * n + ((i) - 1)) * n + ((j) - 1)]
#define KK(i, lc) kk[((lc) - 1) * n + ((i) - 1)]
#define ZZ(i, lc) zz[((lc) - 1) * n + ((i) - 1)]
#define ZP0(i, lc) zp0[((lc) - 1) * n + ((i) - 1)]
#define ZP1(i, lc) zp1[((lc) - 1) * n + ((i) - 1)]

    // forward-eliminated RHS (written by the LU kernel) into LDS
#pragma unroll 4
    for (int i = lane; i < N; i += 64) b[i] = yv[i];
    wave_lds_sync();

    // ---- back-substitution, column oriented (SGBSL second loop, disutil.f:1038-1050).
    //      U is row-major in HBM (ufac[i][j-i]); blocks of 16 columns are transposed through
    //      an LDS stage: stage[r][c] = U(i0+r, k0+c), rows i0 = k0-(2n-1) .. k1.  The running
    //      right-hand side lives in registers on a ring of RS rows (row i <-> lane i%64, slot
    //      (i%RS)/64), so the only serial chain per column is readlane -> divide -> FMA; the
    //      stage entries and the pivot reciprocals of a block are fetched ahead of it. ----
    {
        constexpr int BC = kBackBlock, SP = BC + 1, NR = UB + BC;
        constexpr int NB = (NR + 63) / 64, RS = 64 * NB;
        double *stage = win;                              // NR rows + one all-zero row
        double bval[NB];
        int rowi[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) bval[q] = 0.0;
        if (lane < SP) stage[NR * SP + lane] = 0.0;
        int lo_loaded = N + 1;
        // stage[r][c] of a block comes from 64/BC rows x BC columns per load instruction
        // (contiguous segments of U's rows); the loads of block k1-BC are in flight while block k1 is solved
        constexpr int RPL = 64 / BC;                       // rows per load instruction
        constexpr int NLD = (NR + RPL - 1) / RPL;
        const int rr = lane / BC, cc = lane % BC;
        auto load_block = [&](int k1, double (&v)[NLD]) {
            const int k0 = (k1 - BC + 1 > 1) ? k1 - BC + 1 : 1;
            const int i0 = k0 - UB;
            const int j = k0 + cc;
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const int r = it * RPL + rr;
                const int i = i0 + r;
                v[it] = 0.0;
                // (row i is the ((i-1) % n)-th row of its layer: its support ends with the next layer's
                //  columns, UB - (i-1) % n places right of the diagonal; sbd_band4.hpp stores no more)
                // U(i, j): diagonal-relative (sbd_band.hpp) or, from sbd_band4.hpp, relative to the first
                // column of row i's layer (block-aligned rows: every store of that kernel is a full line)
                if (r < NR && i >= 1 && i <= k1 && j <= k1 && j >= i && j - i <= UB - (i - 1) % n)
                    v[it] = ufac[(size_t)(i - 1) * UW + (P.ublock ? j - 1 - ((i - 1) / n) * n : j - i)];
            }
        };
        double cur[NLD];
        load_block(N, cur);
        for (int k1 = N; k1 >= 1; k1 -= BC) {
            const int k0 = (k1 - BC + 1 > 1) ? k1 - BC + 1 : 1;
            const int i0 = k0 - UB;                   // may be <= 0: rows < 1 hold zeros
            wave_lds_sync();
#pragma unroll
            for (int it = 0; it < NLD; ++it)
                if (it * RPL + rr < NR) stage[(it * RPL + rr) * SP + cc] = cur[it];
            if (k1 - BC >= 1) load_block(k1 - BC, cur);
            // rows of this lane during the block, and the rows that enter the ring with it
            const int lo = (i0 > 1) ? i0 : 1;
            unsigned ubase[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int d = (((k1 - lane - 64 * q) % RS) + RS) % RS;
                rowi[q] = k1 - d;
                if (rowi[q] >= lo && rowi[q] < lo_loaded) bval[q] = b[rowi[q] - 1];
                const int r = rowi[q] - i0;
                ubase[q] = lds_addr(stage + ((r >= 0 && r < NR) ? r : NR) * SP);
            }
            lo_loaded = lo;
            wave_lds_sync();
            double u[NB][BC], dg[BC], rd[BC];
            const unsigned sbase = lds_addr(stage);
            auto fetch = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
#pragma unroll
                for (int q = 0; q < NB; ++q) u[q][c] = lds_read_b64<c * 8>(ubase[q]);
                dg[c] = lds_read_b64<((UB + c) * SP + c) * 8>(sbase);   // U(k0+c, k0+c), broadcast
            };
            static_for<BC>(fetch);
            lds_wait();
#pragma unroll
            for (int c = 0; c < BC; ++c) {
                double r = __builtin_amdgcn_rcp(dg[c]);
                r = r * (2.0 - dg[c] * r);
                rd[c] = r * (2.0 - dg[c] * r);
            }
#pragma unroll
            for (int c = BC - 1; c >= 0; --c) {
                const int k = k0 + c;
                if (k <= k1) {
                    const int km = k % RS;
                    const int lk = km & 63, qk = km >> 6;
                    double bsel = bval[0];
#pragma unroll
                    for (int q = 1; q < NB; ++q) bsel = (qk == q) ? bval[q] : bsel;
                    const double bk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bsel), lk),
                                                       __builtin_amdgcn_readlane(__double2loint(bsel), lk));
                    // b(k)/U(k,k): product with the refined reciprocal plus one residual correction
                    const double q0 = bk * rd[c];
                    const double xk = q0 + (bk - q0 * dg[c]) * rd[c];
                    const double t = -xk;
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        const double upd = bval[q] + t * u[q][c];
                        bval[q] = (lane == lk && qk == q) ? xk : upd;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NB; ++q)
                if (rowi[q] >= k0 && rowi[q] <= k1) b[rowi[q] - 1] = bval[q];
        }
        wave_lds_sync();
    }
    // LL(j, lc) = B((lc-1)*n + j) (disort.f:3624-3633)
    {
        double *ll = P.ll + (size_t)ms * L * n;
        for (int i = lane; i < N; i += 64) ll[i] = b[i];
    }

    // ---- FLUXES (mode 0) ----
    if (mazim != 0) return;
    {
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
                    for (int iq = 1; iq <= nn; ++iq) {
                        const double u = u0c[li * n + iq - 1];
                        uavg = uavg + cwt[nn - iq] * u;
                        fldn = fldn + cwt[nn - iq] * cmu[nn - iq] * u;
                    }
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
