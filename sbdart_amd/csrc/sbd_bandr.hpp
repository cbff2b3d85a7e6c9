// Band LU for NSTR 34..40, a row per lane: one wave per (work item, azimuth mode) factors the two-point boundary
// system of the constants of integration (SETMTX + SOLVE0, disort.f:2702-2994, 3322-3637) with LINPACK's partial-pivot
// band LU (SGBFA, disutil.f:771-912) and carries the right-hand side through SGBSL's forward sweep
// (disutil.f:1019-1036), like band_kernel (sbd_band.hpp) and with the same outputs -- U row-major by layer block
// (Params::ublock) and the eliminated right-hand side, finished by sbd_solve.hpp.
//
// Why a row per lane.  While the columns of layer lc are eliminated, the rows with anything left in them are the
// 3*NSTR/2 rows up to the end of interface lc|lc+1 (<= 60: a wave), and everything they hold -- fill-in included --
// lies in the 2*NSTR columns of x_lc and x_lc+1 (<= 80: more than a wave, which is what keeps band1_kernel's column per
// lane from NSTR > 32).  So lane <-> row, registers <-> columns: cur[NSTR] (x_lc, eliminated one per sub-step) and
// nxt[NSTR] (x_lc+1).  A sub-step is
//   pivot search over the live lanes' cur[J] (DPP max scan + ballot; ties to the first row in LINPACK's order, which a
//   per-lane logical row index keeps through the interchanges -- the interchange itself moves no data),
//   multipliers -a/pivot per lane (the pivot's reciprocal by v_rcp_f64 + two Newton steps),
//   for every column right of J: the pivot row's element through an SGPR pair (two v_readfirstlane with EXEC narrowed
//   to the pivot lane, 16 columns a group), one FMA per lane.
// The columns right of J are a suffix of one fixed sequence "columns 1 .. 2*NSTR-1, right-hand side": the sequence is
// written once, in place on fixed registers, and sub-step J jumps into it (generated inline asm, sbd_bandr_step.inc;
// in C++ the compiler's PHI copies and out-of-place FMAs cost a third of the time and twice the registers).  A retired pivot row stays in its lane until the layer is
// done; then the NSTR retired lanes store their U rows and right-hand sides, the survivors move nxt to cur, and the
// freed lanes load the rows of the next interface straight from the matrix-ready blocks ga/gb (boundary rows come
// through a small LDS stage from the generic entry generator).  No window in LDS, no barrier inside a layer.
// Pivot choice, multiplier scaling and update order are LINPACK's; the factors agree with the reference up to FMA
// contraction and the last bit of the reciprocal, as in the other band kernels.
#pragma once
#include "sbd_common.hpp"
#include "sbd_surface.hpp"
#include "sbd_band.hpp"

namespace sbd {

struct BandRowsLds {   // per-wave carve-up (doubles)
    int stage, sbot, total;
    __host__ __device__ BandRowsLds(int n, int nn)
    {
        stage = 0;                                   // [nn][n + 1] boundary rows on their way to the lanes
        sbot = (nn * (n + 1) + 1) & ~1;              // [n] surface-reflection sums
        total = (sbot + n + 1) & ~1;
    }
};

SBD_DEVICE double pick_lane(double x, int src)       // lane `src` (wave-uniform) of a double, through SGPRs
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src),
                            __builtin_amdgcn_readlane(__double2loint(x), src));
}

// The boundary-value system of one (item, mode): right-hand side and matrix rows (SOLVE0 / SETMTX)
template <int NN>
struct BandSystem {
    static constexpr int n = 2 * NN, nn = NN;
    int N, ncut, mazim;
    bool lyrcut, beam, brdf, refl;
    double fbeam, albedo, delm0, umu0, pi, fisot, tplank, bplank;
    const double *gc, *kk, *ek, *zz, *zp0, *zp1, *ga, *gb, *taucpr, *expbea, *cmu, *cwt, *bdrt, *bemt;
    double *sbot;

    SBD_DEVICE double GC(int i, int j, int lc) const { return gc[((size_t)(lc - 1) * n + (i - 1)) * n + (j - 1)]; }
    SBD_DEVICE double KK(int i, int lc) const { return kk[(lc - 1) * n + (i - 1)]; }
    SBD_DEVICE double EK(int i, int lc) const { return ek[(lc - 1) * nn + (i - 1)]; }
    SBD_DEVICE double ZZ(int i, int lc) const { return zz[(lc - 1) * n + (i - 1)]; }
    SBD_DEVICE double ZP0(int i, int lc) const { return zp0[(lc - 1) * n + (i - 1)]; }
    SBD_DEVICE double ZP1(int i, int lc) const { return zp1[(lc - 1) * n + (i - 1)]; }

    SBD_DEVICE void init(const Params &P, int slot, int mazim_, long long ms, const int32_t *svi, double *sbot_)
    {
        const int L = P.L;
        mazim = mazim_;
        ncut = svi[SBD_SVI_NCUT];
        lyrcut = svi[SBD_SVI_LYRCUT] != 0;
        N = ncut * n;
        const SV o(L);
        const double *sv = P.sv + (size_t)slot * P.sv_stride;
        taucpr = sv + o.taucpr();
        expbea = sv + o.expbea();
        bplank = sv[o.bplank()];
        tplank = sv[o.tplank()];
        fbeam = P.fbeam[slot];
        beam = fbeam > 0.0;
        albedo = P.albedo[slot];
        delm0 = (mazim == 0) ? 1.0 : 0.0;
        umu0 = P.umu0; pi = P.pi; fisot = P.fisot;
        cmu = P.t.cmu; cwt = P.t.cwt;
        gc = P.gc + (size_t)ms * L * n * n;
        kk = P.kk + (size_t)ms * L * n;
        ek = P.ek + (size_t)ms * L * nn;
        zz = P.zz + (size_t)ms * L * n;
        zp0 = P.zp0 + (size_t)(ms - mazim) * L * n;   // thermal particular solutions exist for mode 0 only
        zp1 = P.zp1 + (size_t)(ms - mazim) * L * n;
        ga = P.ga + (size_t)ms * L * n * n;
        gb = P.gb + (size_t)ms * L * n * n + (size_t)n * n;    // block of layer lc+1
        // the surface: Lambertian (couples only for m = 0, disort.f:2925) or bidirectional (SURFAC's tables of this mode)
        brdf = P.ibdrf != 0;
        const size_t sidx = surf_index(P, slot, mazim);
        bdrt = brdf ? surf_bdr(P, sidx) : nullptr;
        bemt = brdf ? surf_bem(P, sidx) : nullptr;
        refl = !lyrcut && (brdf || delm0 != 0.0);
        sbot = sbot_;
    }
    // bottom-boundary reflection sums: S(IQ) = sum_k CWT(k) CMU(k) BDR GC(nn+1-k, IQ, ncut), Lambertian BDR = albedo
    // for every pair (SURFAC, disort.f:3746-3763); lane < n
    SBD_DEVICE void fill_sbot(int lane) const
    {
        if (lane < n) {
            double s = 0.0;
            if (refl && !brdf)
#pragma nounroll
                for (int k = 1; k <= nn; ++k) s = s + cwt[k - 1] * cmu[k - 1] * albedo * GC(nn + 1 - k, lane + 1, ncut);
            sbot[lane] = s;
        }
    }
    // right-hand side B (SOLVE0, disort.f:3434-3599), unknown index = (lc-1)*n + iq
    SBD_DEVICE double rhs(int it) const
    {
        double v;
        if (it <= nn) {   // top boundary
            const int iq = it;
            if (mazim == 0) {
                if (beam) v = -ZZ(nn + 1 - iq, 1) - ZP0(nn + 1 - iq, 1) + fisot + tplank;
                else v = -ZP0(nn + 1 - iq, 1) + fisot + tplank;
            } else {
                v = -ZZ(nn + 1 - iq, 1);
            }
        } else if (it > N - nn) {   // bottom boundary
            const int iq = it - (N - nn);
            if (lyrcut) {                                  // nothing comes back from below the cut (disort.f:3441-3452)
                if (mazim > 0) v = -ZZ(iq + nn, ncut) * expbea[ncut];
                else if (beam) v = -ZZ(iq + nn, ncut) * expbea[ncut] - ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
                else v = -ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
            } else {
                v = surf_bottom_rhs(iq, mazim, beam, fbeam, umu0, pi, albedo, bdrt, bemt, nn, cwt, cmu,
                                    zz + (ncut - 1) * n, zp0 + (ncut - 1) * n, zp1 + (ncut - 1) * n,
                                    expbea[ncut], taucpr[ncut], bplank);
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
        return v;
    }
    // element (r, col) of a boundary row (SETMTX, disort.f:2844-2990): a GC element times its STWJ factor
    SBD_DEVICE double boundary_elem(int r, int col) const
    {
        if (col < 1 || col > N) return 0.0;
        double g = 0.0, f = 1.0;
        if (r <= nn) {                       // top boundary: GC(nn+1-r, j, 1) * exp(KK(j,1)*TAUCPR(1))
            if (col <= n) {
                g = GC(nn + 1 - r, col, 1);
                if (col <= nn) f = exp(KK(col, 1) * taucpr[1]);
            }
        } else {                             // bottom boundary, the surface's reflection folded in
            const int iq = col - (N - n);
            if (iq >= 1) {
                g = GC(nn + (r - (N - nn)), iq, ncut);
                if (refl && brdf) {                        // row r - (N - nn) of BDR meets the downward streams (disort.f:2946-2952)
                    double sr = 0.0;
#pragma nounroll
                    for (int k = 1; k <= nn; ++k)
                        sr = sr + cwt[k - 1] * cmu[k - 1] * SBD_BDR(bdrt, r - (N - nn), k) * GC(nn + 1 - k, iq, ncut);
                    g = g - (1.0 + delm0) * sr;
                } else if (refl) g = g - (1.0 + delm0) * sbot[iq - 1];
                if (iq > nn) f = EK(n + 1 - iq, ncut);
            }
        }
        return g * f;
    }
};

template <int NN>
struct BandRows {      // the wave's registers
    static constexpr int n = 2 * NN;
    double cur[n], nxt[n];   // the lane's row: columns of x_lc, of x_lc+1
    double b;                // its right-hand side
    double m;                // its multiplier of the sub-step
    int P;                   // pivot lane of the sub-step (wave-uniform)
    int lrow;                // the row's index in LINPACK's order (moves with the interchanges)
    int k;                   // column being eliminated (wave-uniform)
    int myk, myJ;            // retired lanes: the row of U they hold, its first column inside the layer
    double mypiv;            // ... and its pivot
    bool live, retired;
    double pv_min, pv_max;   // smallest / largest |pivot| among the rows this lane retired
    bool pv_nan;
};

#include "sbd_bandr_step.inc"   // BandRowsStep<NN>: generated inline asm (tools/gen_bandr_step.py)

// sub-step J of a layer, the part before the updates: pivot search over ak = cur[J], multipliers, bookkeeping
template <int NN>
SBD_DEVICE void band_rows_pivot(BandRows<NN> &s, const double akJ, const int J)
{
    const int lane = threadIdx.x;
    const double ak = akJ;               // (lanes without a live row: whatever their registers hold -- never candidates)
    // -1/a for every candidate while the max-scan runs (v_rcp_f64 + two Newton steps, as sbd_band.hpp)
    double rk = __builtin_amdgcn_rcp(ak);
    rk = rk * (2.0 - ak * rk);
    rk = rk * (2.0 - ak * rk);
    rk = -rk;
    // ISAMAX over the live rows: largest |a|, first in LINPACK's row order among equals
    const unsigned hi = s.live ? ((unsigned)__double2hiint(ak) & 0x7fffffffu) : 0u;
    const unsigned mhi = wave_umax<true>(hi);
    unsigned long long hit = __ballot(s.live && hi == mhi);
    if (hit & (hit - 1ull)) {
        const bool c2 = s.live && hi == mhi;
        const unsigned lo = c2 ? (unsigned)__double2loint(ak) : 0u;
        const unsigned mlo = wave_umax<true>(lo);
        hit = __ballot(c2 && lo == mlo);
        if (hit & (hit - 1ull)) {            // equal magnitudes: the lowest logical row
            unsigned long long h = hit;
            int best = 0x7fffffff;
            while (h) {
                const int q = __ffsll((long long)h) - 1;
                h &= h - 1ull;
                const int lr = __builtin_amdgcn_readlane(s.lrow, q);
                if (lr < best) { best = lr; hit = 1ull << q; }
            }
        }
    }
    const int Pl = hit ? __ffsll((long long)hit) - 1 : 0;
    s.P = Pl;
    const double piv = pick_lane(ak, Pl), tsel = pick_lane(rk, Pl);
    const double tinv = (piv != 0.0) ? tsel : 0.0;
    // the interchange: the row that sat at position k takes the pivot row's place in the order
    const int lP = __builtin_amdgcn_readlane(s.lrow, Pl);
    if (s.live && s.lrow == s.k) s.lrow = lP;
    const bool other = s.live && lane != Pl;
    s.m = other ? ak * tinv : 0.0;
    if (lane == Pl && s.live) { s.live = false; s.retired = true; s.myJ = J; s.myk = s.k; s.mypiv = ak; }
}

template <int NN>
__global__ void __launch_bounds__(64, 2) band_rows_kernel(Params P)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int nmode = P.nmode;
    // (blocks in mode-major order, as band_kernel)
    const int mazim = (int)(blockIdx.x / (unsigned)P.nslot);
    const int slot = (int)(blockIdx.x % (unsigned)P.nslot);
    if (mazim >= nmode) return;
    const long long ms = (long long)slot * nmode + mazim;
    constexpr int n = 2 * NN, nn = NN, UW = u_width(n), SP = n + 1;
    const int L = P.L;
    int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const int st0 = svi[SBD_SVI_STATUS];
    const bool dead = (st0 & (0x20 | 0x10 | 0x08)) != 0;
    if (mazim > 0 && (mazim > svi[SBD_SVI_NAZ] || dead)) return;
    if (dead) {   // DISORT returned before computing anything: outputs stay zero (ZEROAL)
        double *flux = P.flux + (size_t)slot * SBD_NFLUX_ * P.nlev;
        for (int i = lane; i < SBD_NFLUX_ * P.nlev; i += 64) flux[i] = 0.0;
        return;
    }
    const BandRowsLds lds(n, nn);
    double *stage = smem + lds.stage;
    BandSystem<NN> S;
    S.init(P, slot, mazim, ms, svi, smem + lds.sbot);
    S.fill_sbot(lane);
    wave_lds_sync();
    const int ncut = __builtin_amdgcn_readfirstlane(S.ncut), N = ncut * n;   // (wave-uniform, and known to be)
    double *yv = P.yv + (size_t)ms * L * n;
    double *ufac = P.ufac + (size_t)ms * (size_t)(L * n) * UW;

    BandRows<NN> s;
    static_for<n>([&](auto cc) { s.cur[decltype(cc)::value] = 0.0; s.nxt[decltype(cc)::value] = 0.0; });
    s.b = 0.0; s.m = 0.0; s.P = 0; s.lrow = 0; s.k = 1; s.myk = 0; s.myJ = 0; s.mypiv = 0.0; s.live = false; s.retired = false;
    s.pv_min = 1.0e300; s.pv_max = 0.0; s.pv_nan = false;

    // nb boundary rows r0, r0+1, .. (columns c0+1 .. c0+n) to the free lanes of rank rank0.. through the LDS stage
    auto enter_boundary = [&](int r0, int c0, int nb, bool isfree, int rank, int rank0) {
        wave_lds_sync();
        for (int idx = lane; idx < nb * n; idx += 64) {
            const int i = idx / n, c = idx - i * n;
            stage[i * SP + c] = S.boundary_elem(r0 + i, c0 + 1 + c);
        }
        wave_lds_sync();
        if (isfree && rank >= rank0 && rank < rank0 + nb) {
            const int i = rank - rank0;
            const double *row = stage + i * SP;
            static_for<n>([&](auto cc) { constexpr int c = decltype(cc)::value; s.cur[c] = row[c]; s.nxt[c] = 0.0; });
            s.lrow = r0 + i;
            s.b = S.rhs(r0 + i);
            s.live = true;
        }
    };

    for (int lc = 1; lc <= ncut; ++lc) {
        // ---- the rows that enter for this layer take the lanes free since the last one ----
        const bool isfree = !s.live;
        const unsigned long long fmask = __ballot(isfree);
        const int rank = __popcll(fmask & ((1ull << lane) - 1ull));
        int taken = 0;
        if (lc == 1) { enter_boundary(1, 0, nn, isfree, rank, 0); taken = nn; }
        if (lc < ncut) {                      // interface lc | lc+1: rows nn+(lc-1)n+1 .., matrix-ready blocks, unit stride
            if (isfree && rank >= taken && rank < taken + n) {
                const int qq = (lc - 1) * n + (rank - taken);
                const double2 *pa = (const double2 *)(S.ga + (size_t)qq * n), *pb = (const double2 *)(S.gb + (size_t)qq * n);
                static_for<NN>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    const double2 va = pa[c], vb = pb[c];
                    s.cur[2 * c] = va.x; s.cur[2 * c + 1] = va.y;
                    s.nxt[2 * c] = vb.x; s.nxt[2 * c + 1] = vb.y;
                });
                s.lrow = nn + qq + 1;
                s.b = S.rhs(nn + qq + 1);
                s.live = true;
            }
        } else {
            enter_boundary(N - nn + 1, N - n, nn, isfree, rank, taken);
        }
        // ---- the layer's NSTR columns ----
#pragma nounroll
        for (int J = 0; J < n; ++J) {
            double akJ;
            BandRowsStep<NN>::pick(s.cur, __builtin_amdgcn_readfirstlane(J), akJ);
            band_rows_pivot<NN>(s, akJ, J);
            BandRowsStep<NN>::run(s.cur, s.nxt, s.b, s.m, 1ull << __builtin_amdgcn_readfirstlane(s.P), __builtin_amdgcn_readfirstlane(J), __builtin_amdgcn_readfirstlane(lc == ncut ? 1 : 0));
            s.k = s.k + 1;
        }
        // ---- the retired rows leave: U(k, .) by layer block (register c <-> column c of [x_lc, x_lc+1]), B(k) ----
        if (s.retired) {
            double *urow = ufac + (size_t)(s.myk - 1) * UW;
            static_for<NN>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                *(double2 *)(urow + 2 * c) = make_double2(s.cur[2 * c], s.cur[2 * c + 1]);   // (left of the diagonal: never read)
            });
            if (lc < ncut) {
                static_for<NN>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    *(double2 *)(urow + n + 2 * c) = make_double2(s.nxt[2 * c], s.nxt[2 * c + 1]);
                });
            }
            yv[s.myk - 1] = s.b;
            { const double ap = fabs(s.mypiv); s.pv_nan = s.pv_nan || (ap != ap); s.pv_min = fmin(s.pv_min, ap); s.pv_max = fmax(s.pv_max, ap); }
            s.retired = false;
        }
        // ---- the survivors' x_lc+1 becomes x_lc ----
        static_for<n>([&](auto cc) { constexpr int c = decltype(cc)::value; s.cur[c] = s.nxt[c]; s.nxt[c] = 0.0; });
    }
    // 1 + min|pivot| / max|pivot| == 1 (a zero pivot included), silent on NaN like the reference's 1 + RCOND == 1
    // (the pivot ratio stands in for RCOND as in band1 / band4, sbd_band1.hpp)
    double pmin = s.pv_min, pmax = s.pv_max;
    for (int d = 32; d >= 1; d >>= 1) {
        pmin = fmin(pmin, __shfl_xor(pmin, d));
        pmax = fmax(pmax, __shfl_xor(pmax, d));
    }
    const bool any_nan = __ballot(s.pv_nan) != 0ull;
    if (!any_nan && pmin <= 1.1102230246251565e-16 * pmax && lane == 0) atomicOr(&svi[SBD_SVI_STATUS], 0x01);
}

}  // namespace sbd
