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
#include "sbd_bandsys.hpp"

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
    bool func;               // FUSED: the lane holds a row of FLUXES' functionals -- eliminated like a live row, never a pivot
    int fk;                  // ... which: 0..2 the top level's (mean intensity, down, up), 3..5 the surface level's
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
    // the pivot and its row's place in the order, from the pivot lane: v_readfirstlane with EXEC narrowed to that lane
    // (one issue slot each; v_readlane with an SGPR lane select costs 1.7, profiles/r05_valu_rates.txt)
    int plo, phi, lP;
    asm volatile("s_mov_b64 exec, %3\n\tv_readfirstlane_b32 %0, %4\n\tv_readfirstlane_b32 %1, %5\n\tv_readfirstlane_b32 %2, %6\n\t"
                 "s_mov_b64 exec, -1\n\ts_nop 1"
                 : "=s"(plo), "=s"(phi), "=s"(lP)
                 : "s"(1ull << Pl), "v"(__double2loint(ak)), "v"(__double2hiint(ak)), "v"(s.lrow));
    const double piv = __hiloint2double(phi, plo);
    // -1/pivot (v_rcp_f64 + two Newton steps, as sbd_band.hpp forms it lane by lane: the same bits)
    double rk = __builtin_amdgcn_rcp(piv);
    rk = rk * (2.0 - piv * rk);
    rk = rk * (2.0 - piv * rk);
    const double tinv = (piv != 0.0) ? -rk : 0.0;
    // the interchange: the row that sat at position k takes the pivot row's place in the order
    if (s.live && s.lrow == s.k) s.lrow = lP;
    const bool other = (s.live || s.func) && lane != Pl;
    s.m = other ? ak * tinv : 0.0;
    if (lane == Pl && s.live) { s.live = false; s.retired = true; s.myJ = J; s.myk = s.k; s.mypiv = ak; }
}

SBD_DEVICE double wave_sum(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = v + __shfl_xor(v, d, 64);
    return v;
}

// FUSED (flux-only run, output levels = top of layer 1 and the surface; sbd_band1.hpp / sbd_band4.hpp have the long
// version): FLUXES' three angular sums at a level are linear functionals c^T x of the solution; each rides through the
// elimination as one more row of [A b; c^T 0] that never takes part in the pivot search, so that after the last step
// its right-hand side is -c^T x.  A row is a lane here, and 3 NSTR/2 + 3 <= 63: the top level's three rows take lanes
// that would idle, the surface level's three join for the last layer.  No U, no B, no back-substitution kernel.
template <int NN, bool FUSED>
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
        if constexpr (FUSED) { if (lane == 0) P.status[slot] = st0; }     // (the last kernel of a fused pass: no finish_kernel)
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
    double *ufac = FUSED ? nullptr : P.ufac + (size_t)ms * (size_t)(L * n) * UW;
    const SV o(L);
    const double *sv = P.sv + (size_t)slot * P.sv_stride;

    BandRows<NN> s;
    static_for<n>([&](auto cc) { s.cur[decltype(cc)::value] = 0.0; s.nxt[decltype(cc)::value] = 0.0; });
    s.b = 0.0; s.m = 0.0; s.P = 0; s.lrow = 0; s.k = 1; s.myk = 0; s.myJ = 0; s.mypiv = 0.0; s.live = false; s.retired = false; s.func = false; s.fk = 0;
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

    // FUSED: the three functional rows of output level `lev`, which lies in layer lcf (the layer whose columns are x_lc
    // now), to the free lanes of rank rank0..: c_k(j) = sum_i w_i GC(i, j, lcf) * exp(-KK(j, lcf) (utau' - reference depth
    // of column j's scaling)), k = mean intensity / downward / upward weights (sbd_band1.hpp)
    auto enter_functionals = [&](int lcf, int lev, bool valid, int fk0, bool isfree, int rank, int rank0) {
        wave_lds_sync();
        if (lane < n) {
            double c0 = 0.0, c1 = 0.0, c2 = 0.0;
            if (valid) {
                const int jq = lane + 1;
                const double up = sv[o.utaupr() + lev];
                const double ref = (lane < nn) ? S.taucpr[lcf] : S.taucpr[lcf - 1];
                const double e = exp(-S.KK(jq, lcf) * (up - ref));
                double sa = 0.0, sd = 0.0, su = 0.0;
#pragma nounroll
                for (int i = 0; i < n; ++i) {
                    const int iw = (i < nn) ? nn - 1 - i : i - nn;
                    const double g = S.GC(i + 1, jq, lcf), w = S.cwt[iw];
                    sa = sa + w * g;
                    if (i < nn) sd = sd + (w * S.cmu[iw]) * g;
                    else su = su + (w * S.cmu[iw]) * g;
                }
                c0 = sa * e; c1 = sd * e; c2 = su * e;
            }
            stage[0 * SP + lane] = c0; stage[1 * SP + lane] = c1; stage[2 * SP + lane] = c2;
        }
        wave_lds_sync();
        if (isfree && rank >= rank0 && rank < rank0 + 3) {
            const int i = rank - rank0;
            const double *row = stage + i * SP;
            static_for<n>([&](auto cc) { constexpr int c = decltype(cc)::value; s.cur[c] = row[c]; s.nxt[c] = 0.0; });
            s.b = 0.0;
            s.func = true;
            s.fk = fk0 + i;
        }
    };

    for (int lc = 1; lc <= ncut; ++lc) {
        // ---- the rows that enter for this layer take the lanes free since the last one ----
        const bool isfree = !s.live && !s.func;
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
            taken += n;
        } else {
            enter_boundary(N - nn + 1, N - n, nn, isfree, rank, taken);
            taken += nn;
        }
        if constexpr (FUSED) {
            if (lc == 1) { enter_functionals(1, P.t.level_out[0], true, 0, isfree, rank, taken); taken += 3; }
            if (lc == ncut) {      // (only a level inside layer ncut has any: below a cut-off layer the fluxes stay zero)
                const int levb = P.t.level_out[1];
                enter_functionals(ncut, levb, svi[SBD_SVI_LAYRU + levb] == ncut, 3, isfree, rank, taken);
                taken += 3;
            }
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
          if constexpr (!FUSED) {
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
          }
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
    // (round 6: a FILTER, <= 1e-10 -- the system is listed for band_rcond_kernel, which forms the reference's own band matrix
    //  and raises errmsg 2 on LINPACK's own estimate, sbd_refband.hpp)
    if (lane == 0 && ((!any_nan && pmin <= 1.0e-10 * pmax) || P.rcflag[ms] == 2)) rcond_candidate(P, ms);
    const int status = 0;
    if (status && lane == 0) atomicOr(&svi[SBD_SVI_STATUS], status);
    if constexpr (FUSED) {
        if (lane == 0) P.status[slot] = st0 | status;       // (the last kernel of a fused pass: no finish_kernel)
        // ---- FLUXES (disort.f:1780-2042) at the two levels from the functionals' right-hand sides ----
        double fs[2][3];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const unsigned long long w = __ballot(s.func && s.fk == k);
            fs[k / 3][k % 3] = w ? -pick_lane(s.b, __ffsll((long long)w) - 1) : 0.0;
        }
        const bool mycol = lane < n;
        const int qc = mycol ? lane : 0;
        const int iqw = (qc < nn) ? nn - 1 - qc : qc - nn;
        const double wq = mycol ? S.cwt[iqw] : 0.0, wmq = mycol ? S.cwt[iqw] * S.cmu[iqw] : 0.0;
        const double pi = P.pi, umu0 = P.umu0, fbeam = S.fbeam;
        const bool beam = S.beam;
        const int32_t *layru = svi + SBD_SVI_LAYRU;
        const double *utau = sv + o.utau(), *utaupr = sv + o.utaupr(), *ssalbv = sv + o.ssalb();
        const double *xr0 = sv + o.xr0(), *xr1 = sv + o.xr1();
        const int nlev = P.nlev;
        double *flux = P.flux + (size_t)slot * SBD_NFLUX_ * nlev;
#pragma unroll
        for (int ol = 0; ol < 2; ++ol) {
            const int lev = P.t.level_out[ol];
            const int lc = layru[lev];
            double rfldir = 0.0, rfldn = 0.0, flup = 0.0, dfdt = 0.0, uavg = 0.0;
            if (lc <= ncut) {       // (levels below a cut-off layer stay zero, disort.f:1907-1916)
                const double up = utaupr[lev];
                // particular solutions' share of U0C(iq): ZZ e^{-tau'/mu0} + ZPLK0 + ZPLK1 tau' (disort.f:1945-1960)
                double part = S.zp0[(lc - 1) * n + qc] + S.zp1[(lc - 1) * n + qc] * up;
                if (beam) part = S.zz[(lc - 1) * n + qc] * exp(-up / umu0) + part;
                const double uavg_s = fs[ol][0] + wave_sum(wq * part);
                const double fldn_s = fs[ol][1] + wave_sum((qc < nn) ? wmq * part : 0.0);
                const double flup_s = fs[ol][2] + wave_sum((qc >= nn) ? wmq * part : 0.0);
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
            if (lane == 0) {
                flux[0 * nlev + ol] = rfldir;
                flux[1 * nlev + ol] = rfldn;
                flux[2 * nlev + ol] = flup;
                flux[3 * nlev + ol] = dfdt;
                flux[4 * nlev + ol] = uavg;
            }
        }
    }
}

}  // namespace sbd
