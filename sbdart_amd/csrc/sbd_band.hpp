// Band kernel: one wave per (work item, azimuth mode) assembles and solves the two-point
// boundary system for the constants of integration (SETMTX + SOLVE0, disort.f:2702-2994,
// 3322-3637, on LINPACK's SGBFA/SGBSL, disutil.f:771-1092) and, for mode 0, evaluates
// the fluxes at the requested levels (FLUXES, disort.f:1780-2042).
//
// The reference builds a dense LINPACK band array (LDA x N, 296 KB at NSTR=16, 33 layers)
// and factors it in place.  Here the matrix is never materialised: partial-pivot LU only
// ever touches rows k..k+NCD and columns k..k+2*NCD, so the wave keeps exactly that
// (NCD+1) x (2*NCD+1) sliding window in LDS.  Rows enter the window generated on the fly
// from the layer eigenvectors (GC rows, unit-stride HBM reads), pivot-row interchanges
// are pointer swaps in a slot table, the multipliers are applied to the right-hand side
// immediately (so L is never stored), and each finished U row is streamed to the HBM
// workspace in column-band order so that back-substitution reads unit-stride columns.
// Pivot choice (first maximal |a|), multiplier scaling (-1/pivot) and the element-wise
// update order are LINPACK's, so the factors agree with the reference up to FMA
// contraction.
#pragma once
#include "sbd_common.hpp"

namespace sbd {

struct BandLds {   // per-wave carve-up (doubles)
    int rw, cw, cwp, win, b, mult, misc, total;
    __host__ __device__ BandLds(int n, int nn, int L, int nlev)
    {
        const int ncd = 3 * nn - 1;
        rw = ncd + 1;
        cw = 2 * ncd + 1;
        cwp = cw | 1;
        win = 0;
        int winsz = rw * cwp;
        const int fluxsz = 2 * 16 * n + 64;   // E / U0C staging for 16 levels at a time
        if (winsz < fluxsz) winsz = fluxsz;
        b = win + winsz;
        mult = b + n * L;
        misc = mult + rw;            // pslot ints (rw) + a few scalars
        total = misc + rw / 2 + 8 + 2 * n;
        total = (total + 1) & ~1;
        (void)nlev;
    }
};

// (value, index) arg-max across the wave with LINPACK's first-maximum tie rule.
SBD_DEVICE void wave_argmax(double &v, int &idx)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

__global__ void __launch_bounds__(64) band_kernel(Params P)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const long long ms = blockIdx.x;
    const int nmode = P.nmode;
    const int mazim = (int)(ms % nmode);
    const int slot = (int)(ms / nmode);
    if (slot >= P.nslot) return;
    const int L = P.L, n = P.n, nn = P.nn;
    int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const int st0 = svi[SBD_SVI_STATUS];
    const double fbeam = P.fbeam[slot];
    const bool dead = (st0 & (0x20 | 0x10 | 0x08)) != 0;
    if (mazim > 0 && (fbeam == 0.0 || dead)) return;
    const int nlev = P.nlev;
    double *flux = P.flux + (size_t)slot * SBD_NFLUX_ * nlev;
    if (dead) {   // DISORT returned before computing anything: outputs stay zero (ZEROAL)
        for (int i = lane; i < SBD_NFLUX_ * nlev; i += 64) flux[i] = 0.0;
        return;
    }
    const int ncut = svi[SBD_SVI_NCUT];
    const bool lyrcut = svi[SBD_SVI_LYRCUT] != 0;
    const bool plank = P.plank[slot] != 0;
    const SV o(L);
    const double *sv = P.sv + (size_t)slot * P.sv_stride;
    const double *taucpr = sv + o.taucpr();
    const double *expbea = sv + o.expbea();
    const double albedo = P.albedo[slot];
    const double delm0 = (mazim == 0) ? 1.0 : 0.0;
    const double umu0 = P.umu0;
    const double *cmu = P.t.cmu, *cwt = P.t.cwt;

    const BandLds lds(n, nn, L, nlev);
    const int RW = lds.rw, CW = lds.cw, CWP = lds.cwp, ncd = RW - 1;
    double *win = smem + lds.win;
    double *b = smem + lds.b;
    double *mult = smem + lds.mult;
    double *sbot = smem + lds.misc + RW / 2 + 2;      // [n] surface-reflection sums (bottom BC)

    const double *gc = P.gc + (size_t)ms * L * n * n;
    const double *kk = P.kk + (size_t)ms * L * n;
    const double *ek = P.ek + (size_t)ms * L * nn;
    const double *zz = P.zz + (size_t)ms * L * n;
    // thermal particular solutions exist for mode 0 only
    const double *zp0 = P.zp0 + (size_t)(ms - mazim) * L * n;
    const double *zp1 = P.zp1 + (size_t)(ms - mazim) * L * n;
    double *ufac = P.ufac + (size_t)ms * (size_t)(L * n) * CW;
    const int N = ncut * n;
#define GC(i, j, lc) gc[((size_t)((lc) - 1) * n + ((i) - 1)) * n + ((j) - 1)]
#define KK(i, lc) kk[((lc) - 1) * n + ((i) - 1)]
#define EK(i, lc) ek[((lc) - 1) * nn + ((i) - 1)]
#define ZZ(i, lc) zz[((lc) - 1) * n + ((i) - 1)]
#define ZP0(i, lc) zp0[((lc) - 1) * n + ((i) - 1)]
#define ZP1(i, lc) zp1[((lc) - 1) * n + ((i) - 1)]
#define WIN(s, j) win[(s) * CWP + ((j) % CW)]

    const bool refl = !(lyrcut || delm0 == 0.0);   // LAMBER: surface couples only for m = 0 (2925)
    // ---- bottom-boundary reflection sums: S(IQ) = sum_k CWT(k) CMU(k) BDR GC(nn+1-k, IQ, ncut),
    //      Lambertian BDR = albedo for every pair (SURFAC, disort.f:3746-3763) ----
    if (lane < n) {
        double s = 0.0;
        if (refl)
            for (int k = 1; k <= nn; ++k) s = s + cwt[k - 1] * cmu[k - 1] * albedo * GC(nn + 1 - k, lane + 1, ncut);
        sbot[lane] = s;
    }
    // ---- right-hand side B (SOLVE0, disort.f:3434-3599), unknown index = (lc-1)*n + iq ----
    const double bplank = sv[o.bplank()], tplank = sv[o.tplank()];
    const bool beam = fbeam > 0.0;
    for (int it = lane + 1; it <= N; it += 64) {
        double v;
        if (it <= nn) {   // top boundary
            const int iq = it;
            v = 0.0;
            if (mazim == 0) {
                if (beam) v = -ZZ(nn + 1 - iq, 1) - ZP0(nn + 1 - iq, 1) + P.fisot + tplank;
                else v = -ZP0(nn + 1 - iq, 1) + P.fisot + tplank;
            } else {
                v = -ZZ(nn + 1 - iq, 1);
            }
        } else if (it > N - nn) {   // bottom boundary
            const int iq = it - (N - nn);
            if (mazim > 0) {
                v = -ZZ(iq + nn, ncut) * expbea[ncut];   // LYRCUT or Lambertian (disort.f:3441-3452)
            } else if (lyrcut) {
                if (beam) v = -ZZ(iq + nn, ncut) * expbea[ncut] - ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
                else v = -ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
            } else {
                const double bdr = albedo, bem = 1.0 - albedo;
                double sum = 0.0;
                if (beam) {
                    for (int jq = 1; jq <= nn; ++jq)
                        sum = sum + cwt[jq - 1] * cmu[jq - 1] * bdr *
                                        (ZZ(nn + 1 - jq, ncut) * expbea[ncut] + ZP0(nn + 1 - jq, ncut)
                                         + ZP1(nn + 1 - jq, ncut) * taucpr[ncut]);
                    v = 2.0 * sum + (bdr * umu0 * fbeam / P.pi - ZZ(iq + nn, ncut)) * expbea[ncut]
                        + bem * bplank - ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
                } else {
                    for (int jq = 1; jq <= nn; ++jq)
                        sum = sum + cwt[jq - 1] * cmu[jq - 1] * bdr *
                                        (ZP0(nn + 1 - jq, ncut) + ZP1(nn + 1 - jq, ncut) * taucpr[ncut]);
                    v = 2.0 * sum + bem * bplank - ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
                }
            }
        } else {   // interface lc | lc+1
            const int q = it - nn - 1;
            const int lc = q / n + 1, iq = q % n + 1;
            if (mazim > 0) {
                v = (ZZ(iq, lc + 1) - ZZ(iq, lc)) * expbea[lc];
            } else if (beam) {
                v = (ZZ(iq, lc + 1) - ZZ(iq, lc)) * expbea[lc] + ZP0(iq, lc + 1) - ZP0(iq, lc)
                    + (ZP1(iq, lc + 1) - ZP1(iq, lc)) * taucpr[lc];
            } else {
                v = ZP0(iq, lc + 1) - ZP0(iq, lc) + (ZP1(iq, lc + 1) - ZP1(iq, lc)) * taucpr[lc];
            }
        }
        b[it - 1] = v;
    }
    wave_lds_sync();

    // ---- matrix entry generator (SETMTX, disort.f:2844-2990): element (r, col) of the
    //      coefficient matrix as a product g*f of one GC element and one STWJ factor ----
    auto entry = [&](int r, int col, double &g, double &f) {
        g = 0.0;
        f = 1.0;
        if (col < 1 || col > N) return;
        if (r <= nn) {                       // top boundary: GC(nn+1-r, j, 1) * exp(KK(j,1)*TAUCPR(1))
            if (col <= n) {
                g = GC(nn + 1 - r, col, 1);
                if (col <= nn) f = exp(KK(col, 1) * taucpr[1]);
            }
        } else if (r > N - nn) {             // bottom boundary, Lambertian reflection folded in
            const int iq = col - (N - n);
            if (iq >= 1) {
                g = GC(nn + (r - (N - nn)), iq, ncut);
                if (refl) g = g - (1.0 + delm0) * sbot[iq - 1];
                if (iq > nn) f = EK(n + 1 - iq, ncut);
            }
        } else {                             // continuity between layers lc and lc+1
            const int q = r - nn - 1;
            const int lc = q / n + 1, jq = q - (lc - 1) * n + 1;
            const int d = col - (lc - 1) * n;
            if (d >= 1 && d <= n) {
                g = GC(jq, d, lc);
                if (d > nn) f = EK(n + 1 - d, lc);
            } else if (d > n && d <= 2 * n) {
                g = -GC(jq, d - n, lc + 1);
                if (d - n <= nn) f = EK(d - n, lc + 1);
            }
        }
    };

    // logical row r lives in slot r % RW, column j at position j % CW; both are tracked
    // with wrap-around counters (no integer division in the loop)
    {
        const int nfirst = (N < RW) ? N : RW;
        for (int r = 1; r <= nfirst; ++r) {
            const int s = r % RW;
            for (int c = lane; c < CW; c += 64) {      // window columns 1..CW at start
                double g, f;
                entry(r, c + 1, g, f);
                win[s * CWP + ((c + 1) % CW)] = g * f;
            }
        }
        wave_lds_sync();
    }

    // ---- banded LU with partial pivoting + forward elimination of B ----
    int status = 0;
    int ju = 0;
    int kr = 1 % RW, kc = 1 % CW;            // slot of row k, position of column k
    const bool two = CW > 64;                // second pass of lanes over the window width
    for (int k = 1; k <= N - 1; ++k) {
        const int lm = (ncd < N - k) ? ncd : N - k;
        // (A) prefetch the row entering after this step (r = k+RW): lane c <-> column k+1+c
        const int rin = k + RW;
        double pg0 = 0.0, pf0 = 1.0, pg1 = 0.0, pf1 = 1.0;
        if (rin <= N) {
            entry(rin, k + 1 + lane, pg0, pf0);
            if (two) entry(rin, k + 1 + lane + 64, pg1, pf1);
        }
        // (B) pivot search over rows k..k+lm of column k (ISAMAX's first-maximum rule);
        //     lane t keeps the signed element a(k+t, k) for the multiplier
        double ak = 0.0;
        double v = -1.0;
        int idx = 1 << 30;
        if (lane <= lm) {
            int s = kr + lane;
            if (s >= RW) s -= RW;
            ak = win[s * CWP + kc];
            v = fabs(ak);
            idx = lane;
        }
        wave_argmax(v, idx);
        if (v == 0.0) idx = 0;               // all-zero column: keep the diagonal, flag it
        const int l = k + idx;
        int sl = kr + idx;
        if (sl >= RW) sl -= RW;
        const int sk = kr;
        const double piv = __shfl(ak, idx, 64);
        const double akk = __shfl(ak, 0, 64);
        {
            const int junew = ncd + l;
            ju = (ju > junew) ? ju : junew;
            if (ju > N) ju = N;
        }
        wave_lds_sync();
        // (C) row interchange (physical, whole window width; column k handled apart: the
        //     pivot goes to the diagonal, every sub-diagonal slot of column k is cleared
        //     because column k+CW reuses it -- LINPACK's fill-in zeroing) + RHS interchange,
        // (D) multipliers (-a/pivot) straight from the registers of the pivot search,
        //     applied to B at once (SGBSL's forward sweep, disutil.f:1019-1036)
        if (idx != 0) {
            for (int c = lane; c < CW; c += 64) {
                if (c != kc) {
                    const double a = win[sk * CWP + c], bb = win[sl * CWP + c];
                    win[sk * CWP + c] = bb;
                    win[sl * CWP + c] = a;
                }
            }
        }
        const double bk_old = b[k - 1], bl_old = b[l - 1];
        const double bk = (idx != 0) ? bl_old : bk_old;      // B(k) after the interchange
        if (piv == 0.0) status |= 0x01;
        const double tinv = (piv != 0.0) ? -1.0 / piv : 0.0;
        if (lane == 0) {
            win[sk * CWP + kc] = piv;
            b[k - 1] = bk;
        }
        if (lane >= 1 && lane <= lm) {
            int s = kr + lane;
            if (s >= RW) s -= RW;
            const double aik = (lane == idx) ? akk : ak;     // element below the pivot after the swap
            const double m = aik * tinv;
            win[s * CWP + kc] = 0.0;
            mult[lane] = m;
            const double bi = (lane == idx) ? bk_old : b[k + lane - 1];
            b[k + lane - 1] = bi + bk * m;
        }
        wave_lds_sync();
        // (E) rank-1 update: lane <-> column, rows in register chunks of 8 (loads, FMAs,
        //     stores) so that the LDS latency is paid per chunk, not per row
        if (piv != 0.0) {
            const int ncols = ju - k;
            for (int c = lane; c < ncols; c += 64) {
                int pc = kc + 1 + c;
                if (pc >= CW) pc -= CW;
                const double tj = win[sk * CWP + pc];
                if (tj != 0.0) {                   // SAXPY's early return (disutil.f:1711)
                    double *col = win + pc;
                    int s0 = kr;
                    for (int i0 = 1; i0 <= lm; i0 += 8) {
                        double a[8];
                        int so[8];
                        int s = s0;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            s = (s + 1 == RW) ? 0 : s + 1;
                            so[u] = s * CWP;
                            a[u] = (i0 + u <= lm) ? col[so[u]] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (i0 + u <= lm) a[u] = a[u] + tj * mult[i0 + u];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (i0 + u <= lm) col[so[u]] = a[u];
                        s0 = s;
                    }
                }
            }
        }
        // (F) retire row k: stream U(k, k..k+2ncd) to HBM in column-band order (zeros beyond
        //     ju belong to U's band), then put the prefetched row into the freed slot
        {
            const int wmax = (2 * ncd < N - k) ? 2 * ncd : N - k;
            for (int c = lane; c <= wmax; c += 64) {
                int pc = kc + c;
                if (pc >= CW) pc -= CW;
                const int j = k + c;
                ufac[(size_t)(j - 1) * CW + (2 * ncd - c)] = win[sk * CWP + pc];
            }
        }
        wave_lds_sync();
        if (rin <= N) {
            {
                int pc = kc + 1 + lane;          // column k+1+lane
                if (pc >= CW) pc -= CW;
                if (lane < CW) win[sk * CWP + pc] = pg0 * pf0;
            }
            if (two && lane + 64 < CW) {
                int pc = kc + 1 + lane + 64;
                if (pc >= CW) pc -= CW;
                win[sk * CWP + pc] = pg1 * pf1;
            }
        }
        wave_lds_sync();
        kr = (kr + 1 == RW) ? 0 : kr + 1;
        kc = (kc + 1 == CW) ? 0 : kc + 1;
    }
    {   // last row
        const double d = win[kr * CWP + kc];
        if (d == 0.0) status |= 0x01;
        if (lane == 0) ufac[(size_t)(N - 1) * CW + 2 * ncd] = d;
    }
    __threadfence_block();
    wave_lds_sync();

    // ---- back-substitution, column oriented (SGBSL second loop, disutil.f:1038-1050) ----
    {
        const int M = CW;   // ml + mu + 1
        double ucol = 0.0, ucol2 = 0.0, diag = 0.0;
        // column k: entries U(k-lm..k-1, k) live at ufac[k][2ncd-lm .. 2ncd-1]; lane c <-> row k-1-c
        auto load_col = [&](int k, double &u0, double &u1, double &dg) {
            const int lmk = ((k < M) ? k : M) - 1;
            const double *col = ufac + (size_t)(k - 1) * CW;
            u0 = (lane < lmk) ? col[2 * ncd - 1 - lane] : 0.0;
            u1 = (lane + 64 < lmk) ? col[2 * ncd - 1 - (lane + 64)] : 0.0;
            dg = col[2 * ncd];
        };
        load_col(N, ucol, ucol2, diag);
        for (int k = N; k >= 1; --k) {
            double nu0 = 0.0, nu1 = 0.0, nd = 0.0;
            if (k > 1) load_col(k - 1, nu0, nu1, nd);   // prefetch the next column
            const int lmk = ((k < M) ? k : M) - 1;
            const double xk = b[k - 1] / diag;
            wave_lds_sync();
            if (lane == 0) b[k - 1] = xk;
            const double t = -xk;
            if (lane < lmk) b[k - 2 - lane] = b[k - 2 - lane] + t * ucol;
            if (lane + 64 < lmk) b[k - 2 - (lane + 64)] = b[k - 2 - (lane + 64)] + t * ucol2;
            wave_lds_sync();
            ucol = nu0; ucol2 = nu1; diag = nd;
        }
    }
    // LL(j, lc) = B((lc-1)*n + j) (disort.f:3624-3633)
    {
        double *ll = P.ll + (size_t)ms * L * n;
        for (int i = lane; i < N; i += 64) ll[i] = b[i];
    }
    if (status && lane == 0) atomicOr(&svi[SBD_SVI_STATUS], status);

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
#undef EK
#undef ZZ
#undef ZP0
#undef ZP1
#undef WIN
}

}  // namespace sbd
