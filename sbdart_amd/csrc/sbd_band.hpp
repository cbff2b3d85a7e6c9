// Band kernel: one wave per (work item, azimuth mode) assembles and solves the two-point
// boundary system for the constants of integration (SETMTX + SOLVE0, disort.f:2702-2994,
// 3322-3637, on LINPACK's SGBFA/SGBSL, disutil.f:771-1092) and, for mode 0, evaluates
// the fluxes at the requested levels (FLUXES, disort.f:1780-2042).
//
// The reference builds a dense LINPACK band array (LDA x N, 296 KB at NSTR=16, 33 layers)
// and factors it in place.  Here the matrix is never materialised: partial-pivot LU only
// ever touches rows k..k+NCD and columns k..k+2*NCD, so the wave keeps exactly that sliding
// window in LDS.  NSTR is a template parameter, so the window geometry is compile time:
//   * rows are stored without wrap-around in RW+MARGIN physical rows (re-based every MARGIN
//     steps), so the rank-1 update addresses them with immediate offsets;
//   * columns use a ring of CW positions (one add/compare per lane per step);
//   * rows enter the window generated on the fly from the layer eigenvectors (prefetched
//     one step ahead, unit-stride HBM reads), the pivot-row interchange is physical;
//   * the multipliers come straight from the registers of the pivot search (found with a
//     DPP max-scan, no LDS round trip), reach the other lanes as broadcast LDS reads, and are
//     applied to the right-hand side at once (L is never stored);
//   * rows enter from the matrix-ready interface blocks ga/gb the layer kernel wrote;
//   * each finished U row is streamed to HBM row-major (coalesced); back-substitution
//     re-reads U in blocks of 16 columns through an LDS transpose stage and runs LINPACK's
//     column-oriented sweep from there.
// Pivot choice (first maximal |a|), multiplier scaling (-1/pivot) and the element-wise
// update order are LINPACK's, so the factors agree with the reference up to FMA
// contraction.
#pragma once
#include "sbd_common.hpp"

namespace sbd {

constexpr int kBandMargin = 8;    // extra physical rows before the window is re-based
constexpr int kBackBlock = 16;    // columns per back-substitution block

struct BandLds {   // per-wave carve-up (doubles)
    int rw, cw, cwp, win, bw, x, mult, misc, total;
    __host__ __device__ BandLds(int n, int nn, int L, int nlev)
    {
        const int ncd = 3 * nn - 1;
        rw = ncd + 1;
        cw = 2 * ncd + 1;
        cwp = cw | 1;
        win = 0;
        const int winsz = (rw + kBandMargin) * cwp;                 // LU phase: window ...
        bw = win + winsz;                                           // ... + RHS window
        const int lusz = winsz + ((rw + kBandMargin + 2) & ~1);
        const int fluxsz = 2 * 16 * n + 64;                         // E / U0C staging, 16 levels at a time
        const int stagesz = (2 * ncd + kBackBlock) * (kBackBlock + 1);   // back-substitution stage
        x = (stagesz > fluxsz ? stagesz : fluxsz);                  // solve phase: stage|flux + X(N)
        x = (x + 1) & ~1;
        const int solvesz = x + n * L;
        mult = ((lusz > solvesz ? lusz : solvesz) + 1) & ~1;
        misc = mult + ((rw + 2) & ~1);
        total = misc + 8 + n;
        total = (total + 1) & ~1;
        (void)nlev;
    }
};

// (value, index) arg-max over the low lanes of the wave with LINPACK's first-maximum tie
// rule, on the DPP network (no LDS round trips): an inclusive max-scan towards the higher
// lanes inside each row of 16 (row_shr 1,2,4,8), then row_bcast15 (and row_bcast31 when more
// than 32 candidates).  At every step the incoming value stems from lower lanes only, so
// "incoming >= mine" keeps the smallest index among equal maxima.  Result in lane 31 / 63.
template <int CTRL, int ROWMASK>
SBD_DEVICE void argmax_step(double &v, int &idx)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
    const int oi = __builtin_amdgcn_update_dpp(idx, idx, CTRL, ROWMASK, 0xF, false);
    const double ov = __hiloint2double(ohi, olo);
    const bool take = ov >= v;
    v = take ? ov : v;
    idx = take ? oi : idx;
}
template <bool WIDE>
SBD_DEVICE void wave_argmax(double &v, int &idx)
{
    argmax_step<0x111, 0xF>(v, idx);   // row_shr:1
    argmax_step<0x112, 0xF>(v, idx);   // row_shr:2
    argmax_step<0x114, 0xF>(v, idx);   // row_shr:4
    argmax_step<0x118, 0xF>(v, idx);   // row_shr:8
    argmax_step<0x142, 0xA>(v, idx);   // row_bcast:15 -> rows 1,3
    if (WIDE) argmax_step<0x143, 0xC>(v, idx);   // row_bcast:31 -> rows 2,3
    constexpr int SRC = WIDE ? 63 : 31;
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
    v = __hiloint2double(hi, lo);
    idx = __builtin_amdgcn_readlane(idx, SRC);
}

// broadcast lane `src` (compile-time) of a double to the whole wave through SGPRs
template <int SRC>
SBD_DEVICE double bcast_lane(double x)
{
    const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(x), SRC);
    const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(x), SRC);
    return __hiloint2double((int)hi, (int)lo);
}

template <int I, int CNT>
struct RowUpdate {   // a(i) += tj * m(i) for CNT rows, m(i) = multiplier held by lane i
    SBD_DEVICE static void load(const double *colp, int stride, double *a)
    {
        if constexpr (I <= CNT) { a[I - 1] = colp[I * stride]; RowUpdate<I + 1, CNT>::load(colp, stride, a); }
    }
    SBD_DEVICE static void fma(double tj, const double *m, double *a)
    {
        if constexpr (I <= CNT) { a[I - 1] = a[I - 1] + tj * m[I]; RowUpdate<I + 1, CNT>::fma(tj, m, a); }
    }
    SBD_DEVICE static void store(double *colp, int stride, const double *a)
    {
        if constexpr (I <= CNT) { colp[I * stride] = a[I - 1]; RowUpdate<I + 1, CNT>::store(colp, stride, a); }
    }
};

template <int NN>
__global__ void __launch_bounds__(64) band_kernel(Params P)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const long long ms = blockIdx.x;
    const int nmode = P.nmode;
    const int mazim = (int)(ms % nmode);
    const int slot = (int)(ms / nmode);
    if (slot >= P.nslot) return;
    constexpr int n = 2 * NN, nn = NN;
    const int L = P.L;
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
    constexpr int ncd = 3 * NN - 1, RW = ncd + 1, CW = 2 * ncd + 1, CWP = CW | 1, MG = kBandMargin;
    double *win = smem + lds.win;
    double *bw = smem + lds.bw;                       // RHS entries of the window rows (LU phase)
    double *b = smem + lds.x;                         // solution vector (solve phase)
    double *yv = P.yv + (size_t)ms * L * n;           // RHS / forward-eliminated RHS in HBM
    double *mult = smem + lds.mult;                   // [RW] multipliers of the current step
    double *sbot = smem + lds.misc + 4;               // [n] surface-reflection sums (bottom BC)

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
        yv[it - 1] = v;
        if (it <= RW) bw[it - 1] = v;
    }
    __threadfence_block();   // RHS in HBM is re-read by this wave (row prefetch)
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

    // logical row k+i lives in physical row kq+i (kq = k - kbase < MARGIN, re-based every
    // MARGIN steps); column j sits at ring position j % CW, tracked by a wrap-around counter
    {
        const int nfirst = (N < RW) ? N : RW;
        for (int r = 1; r <= nfirst; ++r) {
            for (int c = lane; c < CW; c += 64) {      // window columns 1..CW at start
                double g, f;
                entry(r, c + 1, g, f);
                win[(r - 1) * CWP + ((c + 1) % CW)] = g * f;
            }
        }
        wave_lds_sync();
    }

    // ---- banded LU with partial pivoting + forward elimination of B ----
    int status = 0;
    int ju = 0;
    int kq = 0, kc = 1 % CW;                 // physical row of row k, ring position of column k
    constexpr bool two = CW > 64;            // second pass of lanes over the window width
    // Rows enter the window U steps after their HBM loads were issued (software pipeline of
    // depth U over the unrolled step loop): the step never waits for memory latency.
    struct Pre { double g0, f0, g1, f1, bv; };
    const double *ga_ms = P.ga + (size_t)ms * L * n * n;
    const double *gb_ms = P.gb + (size_t)ms * L * n * n + (size_t)n * n;    // block of layer lc+1
    auto load_row = [&](int r, Pre &q) {     // lane c <-> column r-RW+1+c (the window after step r-RW)
        q.g0 = 0.0; q.f0 = 1.0; q.g1 = 0.0; q.f1 = 1.0;
        q.bv = (r <= N) ? yv[r - 1] : 0.0;
        if (r <= N - nn) {                   // interface row: matrix-ready blocks, unit stride
            const int qq = r - nn - 1;                       // row jq = qq % n of interface lc = qq / n + 1
            const double *ga_r = ga_ms + (size_t)qq * n, *gb_r = gb_ms + (size_t)qq * n;
            const int d0 = r - RW + 1 + lane - (qq / n) * n;  // 1..2n inside the row's support
            if (d0 >= 1 && d0 <= n) q.g0 = ga_r[d0 - 1];
            else if (d0 > n && d0 <= 2 * n) q.g0 = gb_r[d0 - n - 1];
            if (two) {
                const int d1 = d0 + 64;
                if (d1 >= 1 && d1 <= n) q.g1 = ga_r[d1 - 1];
                else if (d1 > n && d1 <= 2 * n) q.g1 = gb_r[d1 - n - 1];
            }
        } else if (r <= N) {                 // bottom-boundary rows
            entry(r, r - RW + 1 + lane, q.g0, q.f0);
            if (two) entry(r, r - RW + 1 + lane + 64, q.g1, q.f1);
        }
    };
    constexpr int U = 4;
    Pre pre[U];
#pragma unroll
    for (int u = 0; u < U; ++u) load_row(RW + 1 + u, pre[u]);
    auto step = [&](const int k, Pre &pq) {
        const int lm = (ncd < N - k) ? ncd : N - k;
        const int rin = k + RW;
        // (B) pivot search over rows k..k+lm of column k (ISAMAX's first-maximum rule);
        //     lane t keeps the signed element a(k+t, k) for the multiplier
        double ak = 0.0;
        double v = -1.0;
        int idx = 1 << 30;
        if (lane <= lm) {
            ak = win[(kq + lane) * CWP + kc];
            v = fabs(ak);
            idx = lane;
        }
        // -1/a for every candidate, computed while the max-scan runs (off the critical path)
        const double rk = -1.0 / ak;
        wave_argmax<(RW > 32)>(v, idx);
        if (!(v > 0.0)) idx = 0;             // all-zero (or NaN) column: keep the diagonal, flag it
        const int l = k + idx;
        // idx is wave-uniform: v_readlane with a scalar lane select instead of a bpermute
        const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ak), idx),
                                            __builtin_amdgcn_readlane(__double2loint(ak), idx));
        const double tsel = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rk), idx),
                                             __builtin_amdgcn_readlane(__double2loint(rk), idx));
        const double akk = bcast_lane<0>(ak);
        {
            const int junew = ncd + l;
            ju = (ju > junew) ? ju : junew;
            if (ju > N) ju = N;
        }
        double *rowk = win + kq * CWP;
        wave_lds_sync();
        // (C) row interchange (physical, whole window width; column k handled apart: the
        //     pivot goes to the diagonal, every sub-diagonal slot of column k is cleared
        //     because column k+CW reuses it -- LINPACK's fill-in zeroing) + RHS interchange,
        // (D) multipliers (-a/pivot) straight from the registers of the pivot search,
        //     applied to B at once (SGBSL's forward sweep, disutil.f:1019-1036)
        if (idx != 0) {
            double *rowl = rowk + idx * CWP;
            for (int c = lane; c < CW; c += 64) {
                if (c != kc) {
                    const double a = rowk[c], bb = rowl[c];
                    rowk[c] = bb;
                    rowl[c] = a;
                }
            }
        }
        const double bk_old = bw[kq], bl_old = bw[kq + idx];
        const double bk = (idx != 0) ? bl_old : bk_old;      // B(k) after the interchange
        if (piv == 0.0) status |= 0x01;
        const double tinv = (piv != 0.0) ? tsel : 0.0;
        double mreg = 0.0;                                   // lane t: multiplier of row k+t
        if (lane == 0) {
            rowk[kc] = piv;
            yv[k - 1] = bk;                  // forward-eliminated RHS, final for row k
        }
        if (lane >= 1 && lane <= lm) {
            const double aik = (lane == idx) ? akk : ak;     // element below the pivot after the swap
            mreg = aik * tinv;
            mult[lane] = mreg;
            rowk[lane * CWP + kc] = 0.0;
            const double bi = (lane == idx) ? bk_old : bw[kq + lane];
            bw[kq + lane] = bi + bk * mreg;
        }
        wave_lds_sync();
        // (E) rank-1 update: lane <-> column; the rows of a column are loaded into registers
        //     with immediate offsets, updated with the multipliers read from lanes 1..lm by
        //     v_readlane, and stored back (one LDS latency per column, not per row)
        if (piv != 0.0) {
            const int ncols = ju - k;
            for (int c0 = 0; c0 < ncols; c0 += 64) {
                const int c = c0 + lane;
                int pc = kc + 1 + c;
                if (pc >= CW) pc -= CW;
                const bool actv = c < ncols;
                double tj = actv ? rowk[pc] : 0.0;
                double *colp = rowk + pc;
                if (lm == ncd) {                   // full window: compile-time row count
                    double a[ncd], m[ncd + 1];
#pragma unroll
                    for (int i = 1; i <= ncd; ++i) m[i] = mult[i];       // uniform LDS reads (broadcast)
                    RowUpdate<1, ncd>::load(colp, CWP, a);   // inactive lanes read harmless LDS
                    RowUpdate<1, ncd>::fma(tj, m, a);
                    if (actv && tj != 0.0) RowUpdate<1, ncd>::store(colp, CWP, a);
                } else {                           // the last NCD steps: shrinking window
                    for (int i = 1; i <= lm; ++i) {
                        const double mi = __shfl(mreg, i, 64);
                        if (actv && tj != 0.0) colp[i * CWP] = colp[i * CWP] + tj * mi;
                    }
                }
            }
        }
        // (F) retire row k: stream U(k, k..k+2ncd) to HBM row-major (zeros beyond ju belong
        //     to U's band), then put the prefetched row at the bottom of the window
        {
            const int wmax = (2 * ncd < N - k) ? 2 * ncd : N - k;
            double *urow = ufac + (size_t)(k - 1) * CW;
            for (int c = lane; c <= wmax; c += 64) {
                int pc = kc + c;
                if (pc >= CW) pc -= CW;
                urow[c] = rowk[pc];
            }
        }
        if (rin <= N) {
            double *rowin = rowk + RW * CWP;
            {
                int pc = kc + 1 + lane;          // column k+1+lane
                if (pc >= CW) pc -= CW;
                if (lane < CW) rowin[pc] = pq.g0 * pq.f0;
            }
            if (two && lane + 64 < CW) {
                int pc = kc + 1 + lane + 64;
                if (pc >= CW) pc -= CW;
                rowin[pc] = pq.g1 * pq.f1;
            }
            if (lane == 0) bw[kq + RW] = pq.bv;
        }
        wave_lds_sync();
        kq = kq + 1;
        kc = (kc + 1 == CW) ? 0 : kc + 1;
        load_row(rin + U, pq);               // issue the loads for the row this slot serves next
        if (kq == MG) {                       // re-base: physical rows MG.. -> 0.. (lane <-> column)
            for (int c = lane; c < CW; c += 64) {
#pragma unroll
                for (int i = 0; i < RW; ++i) win[i * CWP + c] = win[(MG + i) * CWP + c];
            }
            const double bmove = (lane < RW) ? bw[MG + lane] : 0.0;
            wave_lds_sync();
            if (lane < RW) bw[lane] = bmove;
            kq = 0;
            wave_lds_sync();
        }
    };
    for (int k = 1; k <= N - 1; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k + u <= N - 1) step(k + u, pre[u]);
    }
    {   // last row
        const double d = win[kq * CWP + kc];
        if (d == 0.0) status |= 0x01;
        if (lane == 0) { ufac[(size_t)(N - 1) * CW] = d; yv[N - 1] = bw[kq]; }
    }
    __threadfence_block();
    wave_lds_sync();
    // forward-eliminated RHS into LDS; agent-scope (sc1, L2-served) loads: these addresses were
    // read earlier by the prefetch and rewritten since, so the CU's L1 may hold stale lines
    for (int i = lane; i < N; i += 64) {
        const unsigned long long bits = __hip_atomic_load((const unsigned long long *)&yv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b[i] = __longlong_as_double((long long)bits);
    }
    wave_lds_sync();

    // ---- back-substitution, column oriented (SGBSL second loop, disutil.f:1038-1050).
    //      U is row-major in HBM (ufac[i][j-i]); blocks of 16 columns are transposed through
    //      an LDS stage: stage[r][c] = U(i0+r, k0+c), rows i0 = k0-2ncd .. k1 ----
    {
        constexpr int BC = kBackBlock, SP = BC + 1, NR = 2 * ncd + BC;
        double *stage = win;
        for (int k1 = N; k1 >= 1; k1 -= BC) {
            const int k0 = (k1 - BC + 1 > 1) ? k1 - BC + 1 : 1;
            const int i0 = k0 - 2 * ncd;                   // may be <= 0: rows < 1 hold zeros
            wave_lds_sync();
            {
                const int rr = lane >> 4, c = lane & 15;
                const int j = k0 + c;
#pragma unroll 4
                for (int r0 = 0; r0 < NR; r0 += 4) {
                    const int r = r0 + rr;
                    const int i = i0 + r;
                    double val = 0.0;
                    if (r < NR && i >= 1 && i <= k1 && j <= k1 && j >= i && j - i <= 2 * ncd)
                        val = ufac[(size_t)(i - 1) * CW + (j - i)];
                    if (r < NR) stage[r * SP + c] = val;
                }
            }
            wave_lds_sync();
            for (int k = k1; k >= k0; --k) {
                const int c = k - k0;
                const int lmk = ((k < CW) ? k : CW) - 1;
                const double diag = stage[(k - i0) * SP + c];
                const double xk = b[k - 1] / diag;
                wave_lds_sync();
                if (lane == 0) b[k - 1] = xk;
                const double t = -xk;
                // row i = k-1-lane  ->  stage row (i - i0)
                if (lane < lmk) b[k - 2 - lane] = b[k - 2 - lane] + t * stage[(k - 1 - lane - i0) * SP + c];
                if (two && lane + 64 < lmk)
                    b[k - 2 - (lane + 64)] = b[k - 2 - (lane + 64)] + t * stage[(k - 1 - (lane + 64) - i0) * SP + c];
                wave_lds_sync();
            }
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
