// Band LU kernel for NSTR <= 16: FOUR boundary-value systems per wave, block form.
//
// Same job as sbd_band.hpp (SETMTX + SOLVE0's right-hand side + SGBFA + the forward half of
// SGBSL, disort.f:2702-2994, 3322-3637, disutil.f:771-912, 1019-1036) and the same outputs (U
// factor row-major 2*NSTR wide, forward-eliminated right-hand side), but the elimination walks
// the matrix LAYER BY LAYER instead of column by column through a LINPACK-wide window:
//
//   unknowns x_lc (NSTR per layer); rows: NN top-boundary rows, NSTR continuity rows per
//   interface [A_lc | B_lc+1] (the matrix-ready blocks ga/gb of the layer kernel), NN bottom rows.
//   Layer step lc holds NN carry rows (what is left of the rows above after x_1..x_lc-1 are
//   gone; the top-boundary rows for lc = 1) and the NSTR rows of interface lc: RW = 3 NN rows
//   over the columns of x_lc and x_lc+1.  NSTR elimination sub-steps with partial pivoting retire
//   NSTR rows to U and leave NN rows that only touch x_lc+1: the next carry.  These are exactly
//   the rows and columns LINPACK's band LU touches (the rows of interface lc+1 it also scans
//   are structurally zero in the pivot column), so pivots and factors are the same up to
//   rounding and the order in which exactly tied candidates are taken.
//
// Mapping (gfx950, wave64): a system owns one ROW OF 16 LANES; lane q of the row holds column q
// of x_lc ("slot 0"), column q of x_lc+1 ("slot 1") and a copy of the right-hand side ("slot 2")
// for all RW window rows in registers (3 RW doubles).  One sub-step J for four systems at once:
//   * pivot search inside lane J (the column's RW-J live rows are that lane's own registers);
//   * the pivot row leaves its registers and the last live row takes its place (an interchange
//     that keeps the live rows in registers 0..RW-2-J, static for the unrolled code); the row
//     index differs per system, so this runs once per distinct index (<= 4 passes, each a
//     computed jump into a table of six-move cases: generated inline asm, sbd_band4_take.inc);
//   * elimination a_s[p] += a_0[p](lane J) * (t_s * (-1/pivot)): the multiplier column is read
//     straight from lane J's registers by the DP-ALU DPP form of the FMA (row_newbcast:J) -- no
//     transposition through LDS, no multiplier registers, no barriers, no LDS at all;
//   * the retired row goes to U in HBM (16 lanes x 8 B per system and slot), B(k) beside it.
// Instruction count per system and elimination step is about a quarter of the one-system-per-
// wave kernels of sbd_band.hpp.
#pragma once
#include "sbd_common.hpp"
#include "sbd_band.hpp"
#include "sbd_surface.hpp"

namespace sbd {

template <int J>
SBD_DEVICE double fmac_lane_bcast(double acc, double m, double t)   // acc + m(lane J of the row) * t
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(m), "v"(t), "n"(J));
    return acc;
}
template <int J>
SBD_DEVICE int int_lane_bcast(int x)                                // x of lane J of the row
{
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + J, 0xF, 0xF, false);
}
template <int J>
SBD_DEVICE double dbl_lane_bcast(double x)
{
    return __hiloint2double(int_lane_bcast<J>(__double2hiint(x)), int_lane_bcast<J>(__double2loint(x)));
}

// sum over the 16 lanes of a row, result in every lane
SBD_DEVICE double row_sum16(double v)
{
    auto dpp = [](double x, auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), C, 0xF, 0xF, false),
                                __builtin_amdgcn_update_dpp(0, __double2loint(x), C, 0xF, 0xF, false));
    };
    v = v + dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v = v + dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v = v + dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v = v + dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    return v;
}

#include "sbd_band4_take.inc"   // TakeRows<RW, LAST>: generated inline asm (tools/gen_band4_take.py)

// FUSED (flux-only runs that ask for the fluxes at the top of the first layer and at the surface only: IOUT 1 / 10
// with the default ZOUT, drt.f:376-381): no factor leaves the kernel.  FLUXES' three angular sums at a level are
// linear functionals c^T x of that layer's integration constants (c = weights^T GC diag(E), disort.f:1945-1994);
// a functional is carried through the elimination as one more row of the augmented system [A b; c^T 0] that is
// never a pivot candidate -- when every unknown is gone its right-hand side holds -c^T A^-1 b.  The three rows of
// the top level live in registers of their own (F0/F1/F2), the three of the surface level ride in the zero rows
// that pad the bottom-boundary block, scaled by 2^-300 (exact, and they never win a pivot search) and marked by a
// tag in their unused x_lc+1 slot.  The U factor, the eliminated right-hand side, LL and the whole back-substitution
// kernel (two thirds of the pipeline's HBM bytes) do not exist in this mode.
// PIVDBG (tests only): the register index of every pivot row goes to Params::pivdbg -- the host replays the window's
// bookkeeping from it and compares the ROWS chosen with LINPACK's (tests/test_gpu_parity.py::test_pivot_sequence...).
// EXACT (sbd_run_cfg::pivot_exact): ISAMAX's first maximum of |a| instead of the keyed search -- a compare and three
// selects per live row instead of 1.5 instructions.
template <int NN, bool FUSED = false, bool PIVDBG = false, bool EXACT = false>
__global__ void __launch_bounds__(64, 2) band4_kernel(Params P)
{
    constexpr int n = 2 * NN, nn = NN, RW = nn + n, UW = u_width(n);
    static_assert(n <= 16, "band4_kernel: a layer's columns must fit a row of 16 lanes");
    static_assert(!FUSED || NN >= 3, "band4_kernel: the fused variant needs three padding rows in the bottom block");
    constexpr double kTiny = 4.909093465297727e-91, kHuge = 2.037035976334486e+90;   // 2^-300, 2^300
    const int lane = threadIdx.x, q = lane & 15;
    const int nmode = P.nmode, L = P.L;
    // (blocks in mode-major order: the items of mode 0 first.  Item-major, the modes an item does not need -- no beam, no
    //  moment left, SBD_SVI_NAZ -- left their live blocks on a few of the eight XCDs: block b goes to XCD b mod 8)
    const long long bid = (long long)blockIdx.x * 4 + (lane >> 4);
    if (bid >= (long long)P.nslot * nmode) return;
    const int mazim = (int)((unsigned)bid / (unsigned)P.nslot);
    const int slot = (int)((unsigned)bid % (unsigned)P.nslot);
    const long long ms = (long long)slot * nmode + mazim;
    int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const int st0 = svi[SBD_SVI_STATUS];
    const double fbeam = P.fbeam[slot];
    const bool dead = (st0 & (0x20 | 0x10 | 0x08)) != 0;
    if (mazim > 0 && (mazim > svi[SBD_SVI_NAZ] || dead)) return;
    const int nlev = P.nlev;
    if (dead) {   // DISORT returned before computing anything: outputs stay zero (ZEROAL)
        double *flux = P.flux + (size_t)slot * SBD_NFLUX_ * nlev;
        for (int i = q; i < SBD_NFLUX_ * nlev; i += 16) flux[i] = 0.0;
        if constexpr (FUSED) { if (q == 0) P.status[slot] = st0; }
        return;
    }
    const int ncut = svi[SBD_SVI_NCUT];
    const bool lyrcut = svi[SBD_SVI_LYRCUT] != 0;
    const SV o(L);
    const double *sv = P.sv + (size_t)slot * P.sv_stride;
    const double *taucpr = sv + o.taucpr();
    const double *expbea = sv + o.expbea();
    const double albedo = P.albedo[slot];
    const double delm0 = (mazim == 0) ? 1.0 : 0.0;
    const double umu0 = P.umu0;
    const double *cmu = P.t.cmu, *cwt = P.t.cwt;
#ifdef SBD_AB_SHARED_INPUTS   // developer A/B (wrong results): every system reads the layer outputs of one of the
    const long long msr = ms % 16;   // first 16 systems (the four of a wave: four different ones) -- 0.85 MB that stay in L2, the
                                     // band kernel's HBM reads all but vanish.  Does its time move?  (tools/ab_traffic.sh)
#else
    const long long msr = ms;
#endif
    const double *gc = P.gc + (size_t)msr * L * n * n;
    const double *kk = P.kk + (size_t)msr * L * n;
    const double *ek = P.ek + (size_t)msr * L * nn;
    const double *zz = P.zz + (size_t)msr * L * n;
    const double *zp0 = P.zp0 + (size_t)(msr - mazim) * L * n;     // thermal solutions: mode 0 only
    const double *zp1 = P.zp1 + (size_t)(msr - mazim) * L * n;
    double *yv = P.yv + (size_t)ms * L * n;
    double *ufac = FUSED ? nullptr : P.ufac + (size_t)ms * (size_t)(L * n) * UW;
    double *bcb = P.bcb + (size_t)ms * n * n;                  // bottom-boundary rows (below)
    const int N = ncut * n;
#define GC(i, j, lc) gc[((size_t)((lc) - 1) * n + ((i) - 1)) * n + ((j) - 1)]
#define KK(i, lc) kk[((lc) - 1) * n + ((i) - 1)]
#define EK(i, lc) ek[((lc) - 1) * nn + ((i) - 1)]
#define ZZ(i, lc) zz[((lc) - 1) * n + ((i) - 1)]
#define ZP0(i, lc) zp0[((lc) - 1) * n + ((i) - 1)]
#define ZP1(i, lc) zp1[((lc) - 1) * n + ((i) - 1)]
    // the surface: Lambertian (couples only for m = 0, disort.f:2925) or bidirectional (SURFAC's tables of this mode)
    const bool brdf = P.ibdrf != 0;
    const size_t sidx = surf_index(P, slot, mazim);
    const double *bdrt = brdf ? surf_bdr(P, sidx) : nullptr, *bemt = brdf ? surf_bem(P, sidx) : nullptr;
    const bool refl = !lyrcut && (brdf || delm0 != 0.0);
    const bool col = q < n;                        // this lane carries a column
    const int iq1 = q + 1;                         // its 1-based index inside a layer

    // ---- right-hand side B (SOLVE0, disort.f:3434-3599), unknown index = (lc-1)*n + iq.  The 2 NN
    //      boundary entries are computed here (lanes q < nn: top row q+1, the next nn lanes: the bottom
    //      rows) and parked in yv; the interface entries are formed on the fly as their rows enter ----
    const bool beam = fbeam > 0.0;
    if (col) {
        const double bplank = sv[o.bplank()], tplank = sv[o.tplank()];
        const int it = (q < nn) ? q + 1 : N - n + q + 1;
        double v;
        if (q < nn) {   // top boundary
            const int iq = it;
            if (mazim == 0) {
                if (beam) v = -ZZ(nn + 1 - iq, 1) - ZP0(nn + 1 - iq, 1) + P.fisot + tplank;
                else v = -ZP0(nn + 1 - iq, 1) + P.fisot + tplank;
            } else {
                v = -ZZ(nn + 1 - iq, 1);
            }
        } else {        // bottom boundary
            const int iq = it - (N - nn);
            if (lyrcut) {                                  // nothing comes back from below the cut (disort.f:3441-3452)
                if (mazim > 0) v = -ZZ(iq + nn, ncut) * expbea[ncut];
                else if (beam) v = -ZZ(iq + nn, ncut) * expbea[ncut] - ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
                else v = -ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
            } else {
                v = surf_bottom_rhs(iq, mazim, beam, fbeam, umu0, P.pi, albedo, bdrt, bemt, nn, cwt, cmu,
                                    zz + (ncut - 1) * n, zp0 + (ncut - 1) * n, zp1 + (ncut - 1) * n,
                                    expbea[ncut], taucpr[ncut], bplank);
            }
        }
        // (ncut = 1: the bottom rows are rows nn+1..n of the same layer; top entries first, then these)
        yv[it - 1] = v;
    }
    // interface lci | lci+1, row iq = q+1: particular solutions of the two layers at the interface
    struct Z3 { double zz, p0, p1; };
    auto load_z = [&](int l) -> Z3 {                       // layer l clamped into 1..L: always a valid address
        const int lz = (l < L) ? l : L;
        const int ix = (lz - 1) * n + (col ? q : 0);
        return Z3{zz[ix], zp0[ix], zp1[ix]};
    };
    // (SOLVE0's three source combinations in one branch-free form: without a beam ZZ is exactly zero and
    //  the first product vanishes; modes m > 0 carry no thermal terms)
    auto interface_rhs = [&](const Z3 &up, const Z3 &dn, double eb, double tc) -> double {
        const double vb = (dn.zz - up.zz) * eb;
        const double vt = vb + dn.p0 - up.p0 + (dn.p1 - up.p1) * tc;
        return (mazim > 0) ? vb : vt;
    };
    // ---- bottom-boundary rows (disort.f:2919-2990), Lambertian reflection folded in:
    //      GC(nn+r, j, ncut) - (1 + delta_m0) * sum_k CWT(k) CMU(k) ALBEDO GC(nn+1-k, j, ncut), times EK(n+1-j)
    //      for j > nn.  They go to a block of their own, padded to NSTR rows with zeros (zero rows never
    //      win a pivot search), so that the last elimination step reads its rows like the others ----
    if (col) {
        double sb = 0.0;
        if (refl && !brdf)
            for (int k = 1; k <= nn; ++k) sb = sb + cwt[k - 1] * cmu[k - 1] * albedo * GC(nn + 1 - k, iq1, ncut);
        const double f = (iq1 > nn) ? EK(n + 1 - iq1, ncut) : 1.0;
        // (stored in the quarter layout the rows of every step are read in, see step_rows below)
        double *bchi = bcb + (q / nn) * (2 * nn * nn) + (q % nn), *bclo = bchi + nn * nn;
        double cb[3] = {0.0, 0.0, 0.0};
        if constexpr (FUSED) {   // the surface level's functionals (only a level inside layer ncut has any)
            const int levb = P.t.level_out[1];
            if (svi[SBD_SVI_LAYRU + levb] == ncut) {
                const double upb = sv[o.utaupr() + levb];
                const double refb = (q < nn) ? taucpr[ncut] : taucpr[ncut - 1];
                const double eb = exp(-KK(iq1, ncut) * (upb - refb)) * kTiny;
                double sa = 0.0, sd = 0.0, su = 0.0;
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    const int iw = (i < nn) ? nn - 1 - i : i - nn;
                    const double g = GC(i + 1, iq1, ncut), w = cwt[iw];
                    sa = sa + w * g;
                    if (i < nn) sd = sd + (w * cmu[iw]) * g;
                    else su = su + (w * cmu[iw]) * g;
                }
                cb[0] = sa * eb; cb[1] = sd * eb; cb[2] = su * eb;
            }
        }
#pragma unroll
        for (int r = 0; r < n; ++r) {
            double g = 0.0;
            if (r < nn) {
                g = GC(nn + 1 + r, iq1, ncut);
                if (refl && brdf) {                        // row r+1 of BDR meets the downward streams (disort.f:2946-2952)
                    double s = 0.0;
                    for (int k = 1; k <= nn; ++k) s = s + cwt[k - 1] * cmu[k - 1] * SBD_BDR(bdrt, r + 1, k) * GC(nn + 1 - k, iq1, ncut);
                    g = g - (1.0 + delm0) * s;
                } else if (refl) g = g - (1.0 + delm0) * sb;
                g = g * f;
            }
            if constexpr (FUSED) { if (r >= nn && r < nn + 3) g = cb[r - nn]; }
            if (r >= nn) bchi[(r - nn) * nn] = g;
            else bclo[(nn - 1 - r) * nn] = g;
        }
    }
    __threadfence_block();   // B and the boundary block are re-read by this wave as rows enter the window

    // raw GC words of the layer below the current interface: lane (system, q) keeps its 2 NN words of GC(lc+1)
    // from the step in which they were slot 1 to the step in which they are slot 0 (own words only: no
    // exchange between lanes, LDS operations of a wave run in order -- no fence)
    __shared__ double gc_next[n * 64];
    double *keep = gc_next + lane;
    // ---- window: RW rows x (x_lc | x_lc+1 | B) ----
    double a0[RW], a1[RW], a2[RW];
    // carry of the first step = the top-boundary rows (SETMTX, disort.f:2887-2915):
    // GC(nn+1-r, j, 1) * exp(KK(j,1)*TAUCPR(1)) for j <= nn (STWJ scaling)
    {
        const double f = (col && iq1 <= nn) ? exp(KK(iq1, 1) * taucpr[1]) : 1.0;
#pragma unroll
        for (int r = 1; r <= nn; ++r) {
            a0[r - 1] = col ? GC(nn + 1 - r, iq1, 1) * f : 0.0;
            a1[r - 1] = 0.0;
            a2[r - 1] = yv[r - 1];
        }
    }
    // FUSED: the top level's three functional rows (mean intensity, downward and upward flux sums) over
    // (x_lc | x_lc+1 | B); they enter with the first step: the level lies in layer 1
    double F0[3] = {0.0, 0.0, 0.0}, F1[3] = {0.0, 0.0, 0.0}, F2[3] = {0.0, 0.0, 0.0};
    if constexpr (FUSED) {
        if (col) {
            const int levt = P.t.level_out[0];
            const double upt = sv[o.utaupr() + levt];
            const double reft = (q < nn) ? taucpr[1] : taucpr[0];
            const double et = exp(-KK(iq1, 1) * (upt - reft));
            double sa = 0.0, sd = 0.0, su = 0.0;
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const int iw = (i < nn) ? nn - 1 - i : i - nn;
                const double g = GC(i + 1, iq1, 1), w = cwt[iw];
                sa = sa + w * g;
                if (i < nn) sd = sd + (w * cmu[iw]) * g;
                else su = su + (w * cmu[iw]) * g;
            }
            F0[0] = sa * et; F0[1] = sd * et; F0[2] = su * et;
        }
    }
    int status = 0;
    // smallest and largest pivot: the stand-in for SGBCO's condition estimate (errmsg 2, disort.f:3607-3610),
    // see near_singular() in sbd_layer.hpp; lane J sees the pivot of sub-step J
    // (kept as leading words: lane J remembers the pivot of sub-step J, one select per sub-step; min / max once per step)
    unsigned pkey = 0u, kpmin = 0x7fffffffu, kpmax = 0u;
    // The NSTR rows that enter with step lc are interface lc's [ga(lc) | gb(lc+1)] with right-hand
    // sides B(nn + (lc-1) n + r), or for lc = ncut the boundary block above beside zeros with B(N-nn+r),
    // r < nn.  They are fetched while step lc-1 is being eliminated: straight into the registers
    // of rows that have been retired (sub-step J frees register RW-1-J), the last E of them --
    // whose registers come free too late to cover the HBM latency -- through E buffer rows, the
    // right-hand sides as one value per lane (lane r <-> row r) spread by DPP at the hand-over.
    constexpr int E = (n < 4) ? n : 4;
    double bufb[E], rhsn = 0.0;
    // Step lci < ncut: interface lci's rows are [GC(lci) * fa' | GC(lci+1) * fb'] (SETMTX, disort.f:2851-2876)
    // with the STWJ factors fa'(j) = EK(n+1-j, lci) for j > nn (else 1), fb'(j) = -EK(j, lci+1) for j <= nn
    // (else -1).  GC is read from its two independent quarters (Params::gcc: V0(iq,jq) = GC(iq+nn, jq+nn) =
    // -GC(nn+1-iq, nn+1-jq), V1(iq,jq) = GC(nn+1-iq, jq+nn) = -GC(iq+nn, nn+1-jq), disort.f:3290-3312): a lane
    // of the upper columns (q >= nn, jq = q-nn+1) reads V0 for the rows r >= nn (iq = r-nn+1) and V1 for the
    // rows r < nn (iq = nn-r); a lane of the lower columns (jq = nn-q) the other way round with the sign
    // flipped, which goes into the factor.  So a row r is one load from `hi + (r-nn) nn` or `lo + (nn-1-r) nn`
    // with per-lane bases -- and the boundary block of step ncut is stored to be read the same way
    // (beside zeros, unscaled).  Beyond ncut: valid memory, never used.
    struct RowSrc { const double *hi, *lo; };
    auto step_rows = [&](int lci, RowSrc &pa, RowSrc &pb, const bool first = false) {
        const bool inner = lci < ncut, last = lci == ncut;
        const int qq = col ? q : 0;
        const bool upper = qq >= nn;
        const int jo = upper ? qq - nn : nn - 1 - qq;                      // jq - 1
        const double *ca = P.gcc + ((size_t)msr * L + ((lci < L ? lci : L) - 1)) * 2 * nn * nn + jo;
        const double *cb = P.gcc + ((size_t)msr * L + ((lci + 1 < L ? lci + 1 : L) - 1)) * 2 * nn * nn + jo;
        const double *bc = bcb + (qq / nn) * (2 * nn * nn) + (qq % nn);
        const double *zr = P.t.zeros + (qq % nn);
        const double *tg = FUSED ? P.t.tags + (qq % nn) : zr;              // (rows nn..nn+2 of the last step: 1, 2, 3)
        // (slot 0 of an inner step is GC(lci): the same words that were slot 1 of the step before -- kept in LDS
        //  across the step, not read from HBM a second time; only the first step fetches them, `first`.  The
        //  loads of the other inner steps stay in the code, branch-free, and hit the zeros table)
        const bool fetch_a = inner && first;
        pa.hi = fetch_a ? (upper ? ca : ca + nn * nn) : (last ? bc : zr);
        pa.lo = fetch_a ? (upper ? ca + nn * nn : ca) : (last ? bc + nn * nn : zr);
        pb.hi = inner ? (upper ? cb : cb + nn * nn) : tg;
        pb.lo = inner ? (upper ? cb + nn * nn : cb) : zr;
    };
    auto row_of = [&](const RowSrc &p, auto rr) -> double {                 // row r (compile time) of a step's block
        constexpr int r = decltype(rr)::value;
        if constexpr (r >= nn) return p.hi[(r - nn) * nn];
        else return p.lo[(nn - 1 - r) * nn];
    };
    auto factor_ptrs = [&](int lci, const double *&pea, const double *&peb) {   // always valid addresses
        const int la = (lci < L) ? lci : L, lb = (lci + 1 < L) ? lci + 1 : L;
        pea = ek + (la - 1) * nn + ((q >= nn && q < n) ? n - q - 1 : 0);
        peb = ek + (lb - 1) * nn + ((q < nn) ? q : 0);
    };
    auto factors = [&](int lci, double eka, double ekb, double &fa, double &fb) {   // (quarter signs folded in)
        const bool inner = lci < ncut;
        fa = inner ? ((q >= nn) ? eka : -1.0) : 1.0;
        fb = inner ? ((q < nn) ? ekb : -1.0) : 1.0;
    };
    const double *pyb = yv + (N - nn) + ((q < nn) ? q : 0);    // bottom-boundary B, lane r <-> row r < nn
    // B of the rows of step lci for lane r <-> row r: an interface (lci < ncut), the bottom boundary, nothing
    auto step_rhs = [&](int lci, const Z3 &up, const Z3 &dn, double eb, double tc, double yb) -> double {
        const double vi = interface_rhs(up, dn, eb, tc), vl = (lci == ncut && q < nn) ? yb : 0.0;
        return (lci < ncut) ? vi : vl;
    };
    {   // rows of step 1 (exposed once per system)
        RowSrc pa, pb;
        step_rows(1, pa, pb, true);
        const Z3 z1 = load_z(1), z2 = load_z(2);
        const double yq = step_rhs(1, z1, z2, expbea[1], taucpr[1], *pyb);
        const double *pea, *peb;
        factor_ptrs(1, pea, peb);
        double fa, fb;
        factors(1, *pea, *peb, fa, fb);
        static_for<n>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            const double va = row_of(pa, rr), vb = row_of(pb, rr);
            keep[r * 64] = vb;
            a0[nn + r] = col ? va * fa : 0.0;
            a1[nn + r] = col ? vb * fb : 0.0;
            a2[nn + r] = dbl_lane_bcast<r>(yq);
        });
    }
    // every load so far has landed before the loop: the waits inside then only count the loop's own
    // memory operations (vmcnt(0), expcnt/lgkmcnt unconstrained)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long lane0 = __ballot(q == 0);         // lane 0 of every system
    for (int lc = 1; lc <= ncut; ++lc) {
        RowSrc pna, pnb;                                        // next step's rows
        step_rows(lc + 1, pna, pnb);
        const int lcb = (lc + 1 < L) ? lc + 1 : L;              // (a valid level index whatever ncut is)
        Z3 zu, zn;                                              // the two layers at interface lc+1
        double ebn, tcn, ybn, fan = 1.0, fbn = 1.0;
        const double *pea, *peb;
        factor_ptrs(lc + 1, pea, peb);
        double *urow0 = FUSED ? nullptr : ufac + (size_t)(lc - 1) * n * UW;      // U rows of this layer
        int qo = q;                                             // (opaque per step: keeps the compiler from hoisting
        asm volatile("" : "+v"(qo));                            //  16 per-sub-step store addresses out of the loop)
        double *yrow0 = yv + (lc - 1) * n;
        double tpair = 0.0;
        // ---- NSTR elimination sub-steps ----
        static_for<n>([&](auto jj) {
            constexpr int J = decltype(jj)::value;
            constexpr int LAST = RW - 1 - J;                    // live rows: registers 0..LAST
            // (1) pivot search in column J = lane J's own registers.  Magnitudes of doubles order like
            //     their bit patterns: key = leading word without the sign, its last 5 bits replaced by
            //     31 - p, so that one unsigned maximum (v_and_or_b32 + v_max3_u32) returns the largest
            //     |a| and, among candidates that agree in the leading 27 bits, the first row.  Partial
            //     pivoting with threshold 1 - 2^-15 (LINPACK's ISAMAX takes the exact maximum; a
            //     pivot within 3e-5 of it bounds the multipliers by 1.00003 instead of 1)
            int idx;
            if constexpr (EXACT) {                              // LINPACK's rule to the letter (disutil.f:2060-2072)
                double best = fabs(a0[0]);
                idx = 0;
#pragma unroll
                for (int p = 1; p <= LAST; ++p) {
                    const double m = fabs(a0[p]);
                    const bool g = m > best;
                    best = g ? m : best;
                    idx = g ? p : idx;
                }
            } else {
                unsigned kmax = 0u;
#pragma unroll
                for (int p = 0; p <= LAST; ++p) {
                    const unsigned key = ((unsigned)__double2hiint(a0[p]) & 0x7fffffe0u) | (unsigned)(31 - p);
                    kmax = (key > kmax) ? key : kmax;
                }
                idx = 31 - (int)(kmax & 31u);
            }
            const int idxb = int_lane_bcast<J>(idx);            // ... to the 16 lanes of the system
            if constexpr (PIVDBG) {
                if (q == 0) P.pivdbg[(size_t)ms * L * n + (size_t)(lc - 1) * n + J] = idxb;
            }
            // (2) pivot row out of its registers, the last live row into them: one pass per
            //     distinct row index among the systems of the wave
            //     The right-hand side is the same in the 16 lanes of a system, a third register set that only sub-step 0
            //     works on: from sub-step 1 on it RIDES IN LANE 0 OF SLOT 0 -- column 0 is finished by then and the
            //     elimination treats the lane like any other column (tp0 of lane 0 = -B(pivot row) / pivot: the very
            //     product the third set's update formed, bit for bit) -- and returns to its set at the hand-over.
            //     That takes the third FMA per row and two of the six moves per take-out pass off 15 of 16 sub-steps.
            constexpr bool RL = J >= 1;
            double t0, t1, t2 = 0.0;
            if constexpr (RL) {
                TakeRows2<RW, LAST>::run(a0, a1, idxb, t0, t1);
                if constexpr (!FUSED) t2 = dbl_lane_bcast<0>(t0);
            } else TakeRows<RW, LAST>::run(a0, a1, a2, idxb, t0, t1, t2);
            // register LAST is free from here on: next interface's row LAST - nn moves in
            // (slot 1 only: slot 0 of an inner step waits in LDS, the bottom-boundary block of the last step is
            //  read at the hand-over -- once per system, not worth registers in every step)
            if constexpr (LAST - nn >= E) {
                const double vb = row_of(pnb, std::integral_constant<int, LAST - nn>{});
                a1[LAST] = (n == 16 || col) ? vb : 0.0;
            }
            if constexpr (J < E) {
                const double vb = row_of(pnb, std::integral_constant<int, J>{});
                bufb[J] = (n == 16 || col) ? vb : 0.0;
            }
            if constexpr (J == 0) {                             // (loads whatever the step is: no branches,
                zu = load_z(lc + 1);                            //  exact wait counts)
                zn = load_z(lc + 2);
                ebn = expbea[lcb];
                tcn = taucpr[lcb];
                ybn = *pyb;
                fan = *pea;
                fbn = *peb;
            }
            if constexpr (J == ((n > 3) ? 3 : n - 1)) {
                rhsn = step_rhs(lc + 1, zu, zn, ebn, tcn, ybn);
                factors(lc + 1, fan, fbn, fan, fbn);            // (raw EK values -> the rows' scale factors)
            }
            // (3) -1/pivot (v_rcp + two Newton steps) in lane J, a zero pivot is flagged and skipped
            double rn = __builtin_amdgcn_rcp(t0);
            rn = rn * (2.0 - t0 * rn);
            rn = rn * (2.0 - t0 * rn);
            rn = (t0 != 0.0) ? -rn : 0.0;
            pkey = (q == J) ? (unsigned)__double2hiint(t0) : pkey;
            // (4) the retired row.  U goes out by layer block: row J of the block holds x_lc's columns in
            //     words 0..n-1 (the finished ones, q < J, carry multipliers nobody reads: masking them out
            //     makes partial-line writes, measured slower) and x_lc+1's in n..2n-1 -- two aligned
            //     128-byte lines per system, no branches; B(k) is the same in the 16 lanes
            if constexpr (FUSED) {
                // nothing is stored: the functional rows below take the place of the back-substitution
            } else if constexpr (n == 16) {
                // 16-column layers: the block is 28 lines of 16 words -- U1's rows (0..15), U0's rows 0..7, and
                // the live halves (columns 8..15) of U0's rows 8..15 two to a line: rows 8+2k | 9+2k.  The
                // even row of a pair waits one sub-step in `tpair`; its words move to lanes 0..7 (row_ror:8)
                // and leave with the odd row as one full line.
                urow0[J * 16 + qo] = t1;
                if constexpr (J < 8) urow0[(16 + J) * 16 + qo] = t0;
                else if constexpr ((J & 1) == 0) tpair = t0;
                else {
                    const double lo8 = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(tpair), 0x128, 0xF, 0xF, false),
                                                        __builtin_amdgcn_update_dpp(0, __double2loint(tpair), 0x128, 0xF, 0xF, false));
                    urow0[(24 + (J - 8) / 2) * 16 + qo] = (q < 8) ? lo8 : t0;
                }
                yrow0[J] = t2;
            } else {
                double *urow = urow0 + J * UW;
                if (col) {
                    urow[qo] = t0;
                    urow[n + qo] = t1;
                }
                if (q == J) yrow0[J] = t2;
            }
            // (5) elimination: a_s[p] += a_0[p](lane J) * (t_s * -1/pivot); columns <= J of
            //     slot 0 are finished (their registers keep the unscaled multipliers)
            const double rnb = dbl_lane_bcast<J>(rn);
            const double tp1 = rnb * t1, tp2 = RL ? 0.0 : rnb * t2;
            // (no mask on slot 0: the finished columns q <= J then collect garbage instead of keeping the multipliers --
            //  nobody reads them again; the lanes left of the diagonal are skipped where U is used)
            const double tp0 = rnb * t0;
#pragma unroll
            for (int p = 0; p < LAST; ++p) {
                a1[p] = fmac_lane_bcast<J>(a1[p], a0[p], tp1);
                if constexpr (!RL) a2[p] = fmac_lane_bcast<J>(a2[p], a0[p], tp2);
                a0[p] = fmac_lane_bcast<J>(a0[p], a0[p], tp0);
            }
            if constexpr (FUSED) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    F1[k] = fmac_lane_bcast<J>(F1[k], F0[k], tp1);
                    if constexpr (!RL) F2[k] = fmac_lane_bcast<J>(F2[k], F0[k], tp2);
                    F0[k] = fmac_lane_bcast<J>(F0[k], F0[k], tp0);
                }
            }
            if constexpr (J == 0) {                             // the right-hand side moves into the finished column's lane
                InjectRhs<RW>::run(a0, a2, lane0);
                if constexpr (FUSED) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) F0[k] = (q == 0) ? F2[k] : F0[k];
                }
            }
        });
        if (col) {
            const unsigned h = pkey & 0x7fffffffu;
            kpmin = (h < kpmin) ? h : kpmin;
            kpmax = (h > kpmax) ? h : kpmax;
        }
        // ---- the nn rows left over only touch x_lc+1: next step's carry ----
#pragma unroll
        for (int p = 0; p < nn; ++p) { a2[p] = dbl_lane_bcast<0>(a0[p]); a0[p] = a1[p]; a1[p] = 0.0; }   // (B back to its set)
        if constexpr (FUSED) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { F2[k] = dbl_lane_bcast<0>(F0[k]); F0[k] = F1[k]; F1[k] = 0.0; }
        }
        // ... and the prefetched rows of the next step complete the window, scaled as they arrive
        // slot 0 of the next step: GC(lc+1) from LDS (inner step), the bottom-boundary block (last step, below)
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const double ga_ = keep[r * 64];
            keep[r * 64] = bufb[r];
            a0[nn + r] = ((n == 16 || col) ? ga_ : 0.0) * fan;
            a1[nn + r] = bufb[r] * fbn;
        }
#pragma unroll
        for (int r = E; r < n; ++r) {
            const double ga_ = keep[r * 64];
            keep[r * 64] = a1[nn + r];
            a0[nn + r] = ((n == 16 || col) ? ga_ : 0.0) * fan;
            a1[nn + r] = a1[nn + r] * fbn;
        }
        if (lc + 1 == ncut) {
            const int qq = col ? q : 0;
            const RowSrc pbot = {bcb + (qq / nn) * (2 * nn * nn) + (qq % nn), bcb + (qq / nn) * (2 * nn * nn) + (qq % nn) + nn * nn};
            static_for<n>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                const double va = row_of(pbot, rr);
                a0[nn + r] = (n == 16 || col) ? va : 0.0;
            });
        }
        static_for<n>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            a2[nn + r] = dbl_lane_bcast<r>(rhsn);
        });
    }
    {   // errmsg 2: min|pivot| <= 8 N eps max|pivot| over the N pivots of the system (a zero pivot included)
        unsigned ka = kpmax, kp = kpmin;
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {
            const unsigned oa = (unsigned)__shfl_xor((int)ka, d, 16), op = (unsigned)__shfl_xor((int)kp, d, 16);
            ka = (oa > ka) ? oa : ka;
            kp = (op < kp) ? op : kp;
        }
        const double am = __hiloint2double((int)ka, 0), pm = __hiloint2double((int)kp, 0);   // (leading words: 2^-20 relative)
        // (1 + ratio == 1, the reference's test on RCOND with the pivot ratio in its place: see sbd_band1.hpp)
        // (pm <= ..., not !(pm > ...): a system full of NaN -- conservative scattering at NSTR 4 makes them in the reference
        //  as well -- has RCOND = NaN there, and 1 + NaN == 1 is false: no warning.  The fuzz's last ten differing
        //  SBDART_WARNING sets of round 4 were all of this kind, tools/warn_probe.py)
        // (round 6: a FILTER, <= 1e-10 -- the system is listed for band_rcond_kernel, which forms the reference's own band
        //  matrix and raises errmsg 2 on LINPACK's own estimate, sbd_refband.hpp)
        if (q == 0 && (pm <= 1.0e-10 * am || P.rcflag[ms] == 2)) rcond_candidate(P, ms);
    }
    if (status) atomicOr(&svi[SBD_SVI_STATUS], status);
    if constexpr (FUSED) {
        if (q == 0) P.status[slot] = st0 | status;          // (the last kernel of a fused pass: no finish_kernel)
        // ---- FLUXES (disort.f:1780-2042) at the two levels from the functionals ----
        // top level: the right-hand sides of the rows F; surface level: the nn rows left over by the last step are the
        // padding rows, three of them tagged 1..3 in what was their x_lc+1 slot (moved to slot 0 by the hand-over)
        double fs[2][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            fs[0][k] = -F2[k];
            double v = 0.0;
#pragma unroll
            for (int p = 0; p < nn; ++p) v = (a0[p] == (double)(k + 1)) ? a2[p] : v;
            fs[1][k] = -v * kHuge;
        }
        const int cq = col ? q : 0;
        const int iqw = (cq < nn) ? nn - 1 - cq : cq - nn;
        const double wq = col ? cwt[iqw] : 0.0, wmq = col ? cwt[iqw] * cmu[iqw] : 0.0;
        const double pi = P.pi;
        const int32_t *layru = svi + SBD_SVI_LAYRU;
        const double *utau = sv + o.utau(), *utaupr = sv + o.utaupr(), *ssalbv = sv + o.ssalb();
        const double *xr0 = sv + o.xr0(), *xr1 = sv + o.xr1();
        double *flux = P.flux + (size_t)slot * SBD_NFLUX_ * nlev;
#pragma unroll
        for (int ol = 0; ol < 2; ++ol) {
            const int lev = P.t.level_out[ol];
            const int lc = layru[lev];
            double rfldir = 0.0, rfldn = 0.0, flup = 0.0, dfdt = 0.0, uavg = 0.0;
            if (lc <= ncut) {       // (levels below a cut-off layer stay zero, disort.f:1907-1916)
                const double up = utaupr[lev];
                // particular solutions' share of U0C(iq): ZZ e^{-tau'/mu0} + ZPLK0 + ZPLK1 tau' (disort.f:1945-1960)
                double part = zp0[(lc - 1) * n + cq] + zp1[(lc - 1) * n + cq] * up;
                if (beam) part = zz[(lc - 1) * n + cq] * exp(-up / umu0) + part;
                const double uavg_s = fs[ol][0] + row_sum16(wq * part);
                const double fldn_s = fs[ol][1] + row_sum16((q < nn) ? wmq * part : 0.0);
                const double flup_s = fs[ol][2] + row_sum16((q >= nn) ? wmq * part : 0.0);
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
#undef GC
#undef KK
#undef EK
#undef ZZ
#undef ZP0
#undef ZP1
}

}  // namespace sbd
