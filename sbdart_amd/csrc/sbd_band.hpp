// Band LU kernel: one wave per (work item, azimuth mode) assembles the two-point boundary
// system for the constants of integration (SETMTX + SOLVE0, disort.f:2702-2994, 3322-3637),
// factors it with LINPACK's partial-pivot band LU (SGBFA, disutil.f:771-912) and runs the
// forward half of SGBSL (disutil.f:1019-1036) on the right-hand side.  The U factor (row-major)
// and the eliminated right-hand side go to HBM; sbd_solve.hpp finishes the solve.
//
// The reference builds a dense LINPACK band array (LDA x N, 296 KB at NSTR=16, 33 layers)
// and factors it in place.  Here the matrix is never materialised: partial-pivot LU only
// ever touches rows k..k+NCD and columns k..k+2*NCD, so the wave keeps exactly that sliding
// window on chip.  NSTR is a template parameter, so the window geometry is compile time.
// The window lives in LDS -- rows without wrap-around in RW+MARGIN physical rows (re-based every MARGIN steps) so that
// the rank-1 update addresses them with immediate offsets, columns on a ring of CW positions, the pivot row kept in
// registers, the right-hand side of the window rows beside it.  (Round 1's second home, a register window for
// NSTR <= 20, went when band4_kernel / band1_kernel took NSTR <= 32: it was only reachable through a developer switch.)
// This kernel serves NSTR 34..40, and every NSTR under SBD_BAND_V1=1 as the cross-check of the block-form kernels.
// Rows enter from the matrix-ready interface blocks ga/gb the layer kernel
// wrote (unit stride, prefetched U steps ahead); the pivot search is a DPP max-scan plus a
// ballot; multipliers come from the registers of the pivot search and are applied to the
// right-hand side at once (L is never stored); only the rows with a non-zero multiplier
// take part in a step; each finished U row is streamed to HBM.
// Pivot choice (first maximal |a|), multiplier scaling (-1/pivot, by reciprocal + Newton)
// and the element-wise update order are LINPACK's, so the factors agree with the reference
// up to FMA contraction and the last bit of the reciprocal.
#pragma once
#include "sbd_common.hpp"
#include "sbd_surface.hpp"
#include "sbd_bandsys.hpp"

namespace sbd {

constexpr int kBandMargin = 4;    // extra physical rows before the window is re-based
constexpr int kBackBlock = 8;     // columns per back-substitution block
// U's upper bandwidth: LINPACK allows 2*NCD, but the layer-block structure caps it at 2*NSTR-1 --
// every candidate pivot row of a column of layer lc ends with layer lc+1's columns, and fill-in
// only copies pivot-row supports -- so U rows are stored NSTR*2 wide (a third fewer bytes)
SBD_DEVICE constexpr int u_width(int n) { return 2 * n; }

struct BandLds {   // per-wave carve-up of the LU kernel (doubles)
    int rw, cw, cwp, win, bw, misc, total;
    __host__ __device__ BandLds(int n, int nn, bool reg)
    {
        const int ncd = 3 * nn - 1;
        rw = ncd + 1;
        cw = reg ? 2 * ncd + 1 : 2 * n;
        cwp = cw | 1;
        win = 0;                                                    // sliding window ...
        bw = win + (reg ? 32 : (rw + kBandMargin) * cwp);           // (register variant: the pivot column only, 2 x 16)
        misc = bw + (reg ? 0 : ((rw + kBandMargin + 2) & ~1));      // ... + its RHS entries
        total = (misc + 8 + n + 1) & ~1;                            // [n] surface-reflection sums
    }
};

struct SolveLds {   // per-wave carve-up of the back-substitution + flux kernel (doubles)
    int stage, x, total;
    __host__ __device__ SolveLds(int n, int nn, int L)
    {
        const int fluxsz = 2 * 16 * n + 64;                         // E / U0C staging, 16 levels at a time
        const int stagesz = (2 * n - 1 + kBackBlock + 1) * (kBackBlock + 1);   // U block (2n-1 super-diagonals) + a zero row
        stage = 0;
        x = ((stagesz > fluxsz ? stagesz : fluxsz) + 1) & ~1;       // X(N) behind the stage
        total = (x + n * L + 1) & ~1;
    }
};

// Pivot search over the low lanes of the wave with LINPACK's first-maximum tie rule, on the DPP
// network (no LDS round trips).  Magnitudes of doubles order like their bit patterns, so the
// maximum is found on 32-bit words: an inclusive v_max_u32 scan of the leading words towards
// the higher lanes inside each row of 16 (row_shr 1,2,4,8; invalid sources read 0), then
// row_bcast15 (and row_bcast31 when more than 32 candidates), lane 31 / 63 holds the maximum;
// the lanes that hold it are found by a ballot, and only if several do, the same scan settles
// it on the trailing words.  The first such lane (find-first-set) is ISAMAX's answer; an
// all-zero column returns lane 0 (the diagonal; the caller flags the zero pivot).
template <int CTRL, int ROWMASK, bool BOUND>
SBD_DEVICE unsigned dpp_umax_step(unsigned v)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(BOUND ? 0 : (int)v, (int)v, CTRL, ROWMASK, 0xF, BOUND);
    return v > o ? v : o;
}
template <bool WIDE>
SBD_DEVICE unsigned wave_umax(unsigned v)
{
    v = dpp_umax_step<0x111, 0xF, true>(v);    // row_shr:1
    v = dpp_umax_step<0x112, 0xF, true>(v);    // row_shr:2
    v = dpp_umax_step<0x114, 0xF, true>(v);    // row_shr:4
    v = dpp_umax_step<0x118, 0xF, true>(v);    // row_shr:8
    v = dpp_umax_step<0x142, 0xA, false>(v);   // row_bcast:15 -> rows 1,3
    if (WIDE) v = dpp_umax_step<0x143, 0xC, false>(v);   // row_bcast:31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, WIDE ? 63 : 31);
}
template <bool WIDE>
SBD_DEVICE int wave_first_max(double a, int lm)
{
    const bool cand = (int)threadIdx.x <= lm;
    const unsigned hi = cand ? ((unsigned)__double2hiint(a) & 0x7fffffffu) : 0u;
    const unsigned mhi = wave_umax<WIDE>(hi);
    unsigned long long hit = __ballot(cand && hi == mhi);
    if (hit & (hit - 1ull)) {                       // several lanes share the leading word
        const bool c2 = cand && hi == mhi;
        const unsigned lo = c2 ? (unsigned)__double2loint(a) : 0u;
        const unsigned mlo = wave_umax<WIDE>(lo);
        hit = __ballot(c2 && lo == mlo);
    }
    return hit ? __ffsll((long long)hit) - 1 : 0;
}

// broadcast lane `src` (compile-time) of a double to the whole wave through SGPRs
template <int SRC>
SBD_DEVICE double bcast_lane(double x)
{
    const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(x), SRC);
    const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(x), SRC);
    return __hiloint2double((int)hi, (int)lo);
}

// ds_read_b64 with an immediate offset, issued without the compiler's pairing into
// ds_read2_b64 (which moves the same bytes at half the LDS rate on gfx950: 128 vs 256 B/clk,
// MI355X_MICROARCH.md section LDS).  The caller waits with lds_wait() before using the values.
template <int OFF>
SBD_DEVICE double lds_read_b64(unsigned addr)
{
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
SBD_DEVICE void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
SBD_DEVICE unsigned lds_addr(const double *p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) double *)p;
}

template <int I, int LAST, int STRIDE>
struct RowUpdate {   // rows I..LAST of one column: a(i) += tj * m(i); lane i holds m(i) in mreg
    template <int U>
    SBD_DEVICE static void loads(unsigned addr, double *a)
    {
        if constexpr (I + U <= LAST) {
            a[U] = lds_read_b64<(I + U) * STRIDE * 8>(addr);
            loads<U + 1>(addr, a);
        }
    }
    template <int U>
    SBD_DEVICE static void fmas(double tj, double mreg, double *a)
    {
        if constexpr (I + U <= LAST) {
            a[U] = a[U] + tj * bcast_lane<I + U>(mreg);      // multiplier through SGPRs (v_readlane)
            fmas<U + 1>(tj, mreg, a);
        }
    }
    SBD_DEVICE static void run(double *colp, double mreg, double tj)
    {
        if constexpr (I <= LAST) {
            constexpr int CNT = LAST - I + 1;
            double a[CNT];
            loads<0>(lds_addr(colp), a);
            lds_wait();
            fmas<0>(tj, mreg, a);
            if (tj != 0.0) {                                            // SAXPY's early return
#pragma unroll
                for (int u = 0; u < CNT; ++u) colp[(I + u) * STRIDE] = a[u];
            }
        }
    }
};

// rows 1..R of one window column, R = the smallest of four compile-time counts >= lme
template <int R, int STRIDE>
SBD_DEVICE void update_rows_fixed(double *colp, double mreg, double tj)
{
    constexpr int H = (R + 1) / 2;                   // two batches: loads in flight vs registers
    RowUpdate<1, H, STRIDE>::run(colp, mreg, tj);
    RowUpdate<H + 1, R, STRIDE>::run(colp, mreg, tj);
}
template <int NN>
SBD_DEVICE void update_rows(double *colp, double mreg, double tj, int lme)
{
    constexpr int ncd = 3 * NN - 1, CWP = (4 * NN) | 1, D = (2 * NN + 3) / 4;   // (LDS variant: 2*NSTR columns)
    if (lme > ncd - D) update_rows_fixed<ncd, CWP>(colp, mreg, tj);
    else if (lme > ncd - 2 * D) update_rows_fixed<ncd - D, CWP>(colp, mreg, tj);
    else if (lme > ncd - 3 * D) update_rows_fixed<ncd - 2 * D, CWP>(colp, mreg, tj);
    else update_rows_fixed<ncd - 3 * D, CWP>(colp, mreg, tj);
}

// register-window helpers (band_kernel<NN, true>)
// t = a[idx], a[idx] = a[0] for a wave-uniform idx: a tree of scalar branches, one leaf runs
template <int LO, int HI, int RW>
SBD_DEVICE void take_row(double (&a)[RW], int idx, double &t)
{
    if constexpr (LO == HI) {
        t = a[LO];
        if constexpr (LO != 0) a[LO] = a[0];
        asm volatile("" : "+v"(t));                  // keep the leaves as branches, not selects
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (idx <= MID) take_row<LO, MID>(a, idx, t);
        else take_row<MID + 1, HI>(a, idx, t);
    }
}
// a[0..] to LDS doubles addr[0..], two registers per ds_write2_b64 (offsets in units of 8 bytes)
template <int J, int RW>
SBD_DEVICE void write_pairs(unsigned addr, const double (&a)[RW])
{
    if constexpr (2 * J + 1 < RW) {
        asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4"
                     :: "v"(addr), "v"(a[2 * J]), "v"(a[2 * J + 1]), "n"(2 * J), "n"(2 * J + 1) : "memory");
        write_pairs<J + 1>(addr, a);
    }
}
// a[i-1] = a[i] + t * m(i) for i <= R, a[i-1] = a[i] beyond.  The multipliers are replicated in
// every row of 16 lanes (lane 16r+q holds m(q) in mlo and m(16+q) in mhi), so that the DP-ALU DPP
// form of the FMA can take them straight from a lane of its own row (row_newbcast): one
// instruction per row of the window instead of two v_readlane and an FMA.
template <int I>
SBD_DEVICE void fmac_rowbcast(double &acc, double m, double t)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(m), "v"(t), "n"(I & 15));
}
template <int R, int RW>
SBD_DEVICE void update_shift(double (&a)[RW], double t, double mlo, double mhi)
{
    static_assert(RW <= 32, "two multiplier registers cover 32 window rows");
    static_for<RW - 1>([&](auto ii) {
        constexpr int i = decltype(ii)::value + 1;
        a[i - 1] = a[i];
        if constexpr (i <= R) fmac_rowbcast<i>(a[i - 1], (i < 16) ? mlo : mhi, t);
    });
}

template <int NN, bool REG = false>   // (REG: always false since round 3)
__global__ void __launch_bounds__(64) band_kernel(Params P)
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
    int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const int st0 = svi[SBD_SVI_STATUS];
    const bool dead = (st0 & (0x20 | 0x10 | 0x08)) != 0;
    if (mazim > 0 && (mazim > svi[SBD_SVI_NAZ] || dead)) return;
    const int nlev = P.nlev;
    double *flux = P.flux + (size_t)slot * SBD_NFLUX_ * nlev;
    if (dead) {   // DISORT returned before computing anything: outputs stay zero (ZEROAL)
        for (int i = lane; i < SBD_NFLUX_ * nlev; i += 64) flux[i] = 0.0;
        return;
    }
    const BandLds lds(n, nn, REG);
    // window width: LINPACK's 2*NCD+1 columns for the register variant (a column per lane), the
    // structural 2*NSTR (u_width) for the LDS variant, which is fed by columns as well as by rows
    constexpr int ncd = 3 * NN - 1, RW = ncd + 1, CW = REG ? 2 * ncd + 1 : 2 * NN * 2, CWP = CW | 1, MG = kBandMargin;
    double *win = smem + lds.win;
    double *bw = smem + lds.bw;                       // RHS entries of the window rows (LU phase)
    double *yv = P.yv + (size_t)ms * L * n;           // RHS / forward-eliminated RHS in HBM
    // the system: right-hand side B (SOLVE0, disort.f:3434-3599; unknown index = (lc-1)*n + iq), the boundary rows
    // (SETMTX, disort.f:2844-2990) and the interface rows from the matrix-ready blocks ga / gb (sbd_bandsys.hpp)
    BandSystem<NN> S;
    S.init(P, slot, mazim, ms, svi, smem + lds.misc + 4);     // [n] surface-reflection sums (bottom BC)
    const int ncut = S.ncut;
    constexpr int UW = u_width(n);
    double *ufac = P.ufac + (size_t)ms * (size_t)(L * n) * UW;
    const int N = ncut * n;
#define WIN(s, j) win[(s) * CWP + ((j) % CW)]
    S.fill_sbot(lane);
    for (int it = lane + 1; it <= N; it += 64) {
        const double v = S.rhs(it);
        yv[it - 1] = v;
        if (!REG && it <= RW) bw[it - 1] = v;
    }
    __threadfence_block();   // RHS in HBM is re-read by this wave (row prefetch)
    wave_lds_sync();
    auto row_elem = [&](int r, int col) -> double { return S.row_elem(r, col); };

    // logical row k+i lives in physical row kq+i (kq = k - kbase < MARGIN, re-based every
    // MARGIN steps); column j sits at ring position j % CW, tracked by a wrap-around counter
    {
        const int nfirst = (N < RW) ? N : RW;
        for (int r = 1; r <= nfirst; ++r) {
            // (row_elem, not entry(): the interface rows come from the ga/gb blocks -- GC itself is only
            //  written for the layers the boundary rows and the output levels need)
            for (int c = lane; c < CW; c += 64)        // window columns 1..CW at start
                win[(r - 1) * CWP + ((c + 1) % CW)] = row_elem(r, c + 1);
        }
        wave_lds_sync();
    }

    // ---- banded LU with partial pivoting + forward elimination of B ----
    int status = 0;
    double pv_min = 1.0e300, pv_max = 0.0;   // errmsg 2: the pivot ratio stands in for RCOND as in band1 / band4 (sbd_band1.hpp)
    bool pv_nan = false;
    int ju = 0;
    int kq = 0, kc = 1 % CW;                 // physical row of row k, ring position of column k
    constexpr bool two = CW > 64;            // second pass of lanes over the window width
    // Rows enter the window U steps after their HBM loads were issued (software pipeline of
    // depth U over the unrolled step loop): the step never waits for memory latency.
    // Lane mapping of a step: lane c <-> window column k+c (c = 0 is the pivot column); its
    // ring position also serves column k+CW of the entering row.
    // The window is 2*NSTR columns wide -- U's structural bandwidth -- although a row's support
    // can reach further right when it enters: those elements are no target of any elimination
    // before their column slides into the window (every pivot row ends inside it), so they are
    // fed late, a column per step (lane t <-> row k+t; only rows k+nn+1.. can reach column k+CW).
    // Interchanges do not disturb this: a row with a non-zero in the pivot column ends inside
    // the window, so neither the pivot row nor the row it displaces has anything left to feed.
    struct Pre { double g0, g1, bv, cv; };
    auto load_row = [&](int r, Pre &q) {
        q.g0 = 0.0; q.g1 = 0.0; q.bv = 0.0; q.cv = 0.0;
        if (r <= N) {
            const int k0 = r - RW;           // the step after which the row (and column k0+CW) enters
            q.bv = yv[r - 1];
            q.g0 = row_elem(r, (lane == 0) ? k0 + CW : k0 + lane);
            if (two) q.g1 = row_elem(r, k0 + lane + 64);
            if (lane > nn && lane <= ncd && k0 + lane <= N) q.cv = row_elem(k0 + lane, k0 + CW);
        }
    };
    constexpr int U = 4;
    Pre pre[U];
#pragma unroll
    for (int u = 0; u < U; ++u) load_row(RW + 1 + u, pre[u]);
    auto step = [&](const int k, Pre &pq) {
        const int lm = (ncd < N - k) ? ncd : N - k;
        const int rin = k + RW;
        int pcl = kc + lane;                 // ring position of column k+lane
        if (pcl >= CW) pcl -= CW;
        int pcl2 = pcl + 64;                 // ... and of column k+lane+64 (wide windows)
        if (pcl2 >= CW) pcl2 -= CW;
        double *rowk = win + kq * CWP;
        // LDS round trip 1 (nothing here depends on the pivot): the pivot column (lane t <->
        // row k+t), the RHS entries of the window rows, and row k itself (lane c <-> column)
        double ak = 0.0, bwl = 0.0;
        if (lane <= lm) {
            ak = rowk[lane * CWP + kc];
            bwl = bw[kq + lane];
        }
        const double tk = rowk[pcl];
        double tk2 = 0.0;
        if (two) tk2 = rowk[pcl2];
        // -1/a for every candidate, computed while the max-scan runs (off the critical path)
        // (v_rcp_f64 + two Newton steps: within an ulp or two of LINPACK's exact -1/pivot)
        double rk = __builtin_amdgcn_rcp(ak);
        rk = rk * (2.0 - ak * rk);
        rk = rk * (2.0 - ak * rk);
        rk = -rk;
        // (B) pivot search over rows k..k+lm of column k (ISAMAX's first-maximum rule)
        const int idx = wave_first_max<(RW > 32)>(ak, lm);   // (an all-zero column keeps the diagonal and is flagged)
        const int l = k + idx;
        // idx is wave-uniform: v_readlane with a scalar lane select instead of a bpermute
        auto pick = [&](double x, int src) {
            return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src),
                                    __builtin_amdgcn_readlane(__double2loint(x), src));
        };
        const double piv = pick(ak, idx), tsel = pick(rk, idx);
        const double akk = bcast_lane<0>(ak);
        const double bk_old = bcast_lane<0>(bwl), bl_old = pick(bwl, idx);
        {
            const int junew = ncd + l;
            ju = (ju > junew) ? ju : junew;
            if (ju > N) ju = N;
        }
        const double bk = (idx != 0) ? bl_old : bk_old;      // B(k) after the interchange
        { const double ap = fabs(piv); pv_nan = pv_nan || (ap != ap); pv_min = fmin(pv_min, ap); pv_max = fmax(pv_max, ap); }
        const double tinv = (piv != 0.0) ? tsel : 0.0;
        // (C) row interchange: row k is retired by this step and never read from LDS again, so
        //     only row l has to receive the old row k (LDS round trip 2: one read, one write);
        //     the pivot row itself lives on in registers (tj).  Column k's slot of the sub-diagonal
        //     rows is handed to column k+CW in (D).
        double tj = tk, tj2 = tk2;
        if (idx != 0) {
            double *rowl = rowk + idx * CWP;
            if (lane >= 1 && lane < CW) { tj = rowl[pcl]; rowl[pcl] = tk; }
            if (two && lane + 64 < CW) { tj2 = rowl[pcl2]; rowl[pcl2] = tk2; }
        }
        // (D) multipliers (-a/pivot) straight from the registers of the pivot search,
        //     applied to B at once (SGBSL's forward sweep, disutil.f:1019-1036); the entering
        //     row takes the physical row below the window
        double mreg = 0.0;                                   // lane t: multiplier of row k+t
        if (lane == 0) {
            yv[k - 1] = bk;                  // forward-eliminated RHS, final for row k
            bw[kq + RW] = pq.bv;
        } else if (lane <= lm) {
            const double aik = (lane == idx) ? akk : ak;     // element below the pivot after the swap
            mreg = aik * tinv;
            rowk[lane * CWP + kc] = pq.cv;                   // column k's slot now serves column k+CW
            const double bi = (lane == idx) ? bk_old : bwl;
            bw[kq + lane] = bi + bk * mreg;
        }
        {
            double *rowin = rowk + RW * CWP;
            if (lane < CW && rin <= N) rowin[pcl] = pq.g0;
            if (two && lane + 64 < CW && rin <= N) rowin[pcl2] = pq.g1;
        }
        // (F) retire row k from registers: stream U(k, k..k+2n-1) to HBM row-major
        {
            const int wmax = (UW - 1 < N - k) ? UW - 1 : N - k;
            double *urow = ufac + (size_t)(k - 1) * UW;
            if (lane <= wmax) urow[lane] = (lane == 0) ? piv : tj;
            if (two && lane + 64 <= wmax) urow[lane + 64] = tj2;
        }
        wave_lds_sync();
        // (E) rank-1 update: lane c <-> column k+c (c >= 1).  LDS round trip 3: all rows of the
        //     lane's column in one batch with immediate offsets; FMAs with the multipliers read
        //     lane by lane through SGPRs; stores.  Only rows up to the last non-zero multiplier
        //     take part: column k of a layer's block is structurally empty below the next
        //     interface (3nn - j rows for the j-th column of a layer instead of NCD), and a
        //     zero multiplier leaves its row unchanged in LINPACK as well.
        const unsigned long long nzm = __ballot(mreg != 0.0);
        if (nzm != 0ull) {
            const int lme = 63 - __clzll((long long)nzm);    // rows k+1..k+lme
            const int ncols = (ju - k < CW - 1) ? ju - k : CW - 1;   // columns k+1..ju (all inside the window)
            const double t1 = (lane >= 1 && lane <= ncols) ? tj : 0.0;    // inactive lanes: no stores
            update_rows<NN>(rowk + pcl, mreg, t1, lme);
            if (two && ncols >= 64) {
                const double t2 = (lane + 64 <= ncols) ? tj2 : 0.0;
                update_rows<NN>(rowk + pcl2, mreg, t2, lme);
            }
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
        { const double ap = fabs(d); pv_nan = pv_nan || (ap != ap); pv_min = fmin(pv_min, ap); pv_max = fmax(pv_max, ap); }
        // 1 + min|pivot| / max|pivot| == 1 (a zero pivot included), silent on NaN like the reference's 1 + RCOND == 1
        // (round 6: a FILTER -- the system is listed for band_rcond_kernel, sbd_refband.hpp)
        if (lane == 0 && ((!pv_nan && pv_min <= 1.0e-10 * pv_max) || P.rcflag[ms] == 2)) rcond_candidate(P, ms);
        if (lane == 0) { ufac[(size_t)(N - 1) * UW] = d; yv[N - 1] = bw[kq]; }
    }
    if (status && lane == 0) atomicOr(&svi[SBD_SVI_STATUS], status);
#undef WIN
}

}  // namespace sbd
