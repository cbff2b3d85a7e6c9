// Layer kernel, fast path.  Same role as sbd_layer.hpp (SOLEIG/UPBEAM/UPISOT/TERPEV/TERPSO
// per (work item, azimuth mode, layer), disort.f:638-693) but the reduced eigenproblem is
// solved the MI355X way instead of with a sequential QR iteration:
//
//   ARRAY = APB*AMB (disort.f:3236-3249) with AMB = M^-1 (S- W - I), APB = M^-1 (S+ W - I),
//   S+ / S- = the even / odd (l-m) parts of sum_l GL_l Y_l(mu_i) Y_l(mu_j)  (both symmetric).
//   With R = (W/M)^1/2 and Q+- = R (W^-1 - S+-) R (symmetric positive definite for physical
//   phase functions), ARRAY is similar to Q- Q+.  Cholesky Q+ = L L^T, Q- = C C^T gives
//   Q- Q+ ~ B^T B with B = C^T L, so the eigenvalues k^2 are the squared singular values of B
//   and the eigenvectors of ARRAY are x = (M R)^-1 L v for the right singular vectors v.
//   B's SVD is computed by ONE-SIDED JACOBI: lane j keeps column j of B in registers,
//   column pairs meet by wave shuffles in a round-robin tournament, every rotation is local
//   to the two lanes -- no LDS traffic, no barriers, NSTR/2 lanes busy per matrix.  The converged
//   column b' = B v gives the eigenvector through C alone (x = (M R)^-1 C^-T b'), see below.
//
// When a Cholesky pivot is not positive (non-physical moments) or some k sits next to 1/mu0, the group
// raises a flag and the QR kernel of sbd_layer.hpp redoes that layer (same outputs, reference algorithm).
//
// UPBEAM/UPISOT (disort.f:4130-4353) solve (D - CC) Z = rhs with the same +-mu symmetry taken
// out: for s = Z+ + Z-, d = Z+ - Z- the NSTR x NSTR system splits into
//     (I - S+ W) s + (M/mu0) d = r+ + r-,     (I - S- W) d + (M/mu0) s = r+ - r-,
// i.e. one NSTR/2 x NSTR/2 system T d = q, T = M/mu0 - mu0 (I - S+ W) M^-1 (I - S- W), for the beam
// source; the thermal one (no M/mu0 coupling there) needs (I - S+ W)^-1 and (I - S- W)^-1.  All three
// inverses are diagonal in the basis of the singular vectors just computed (formulas at the code), so
// no second factorisation is made: each solve is a projection on the columns and a recombination.
#pragma once
#include "sbd_common.hpp"
#include "sbd_layer.hpp"

namespace sbd {

struct Layer2Lds {   // doubles; per-group part + per-block shared part
    int ld, ldq, gl, lu, vec, group_total, shared_y, shared_yu, shared_total;
    __host__ __device__ Layer2Lds(int n, int nn, bool rad, int numu = 0)
    {
        ld = n | 1;
        ldq = nn | 1;                       // Q+-/L/C, later a scratch block: walked by rows and by columns
        gl = 0;
        lu = gl + ((n + 2) & ~1);
        // groups of 8 or 16 lanes (nn > 4) keep L in the lower triangle of ONE block, C transposed in its upper
        // triangle and C's diagonal behind it; the groups of 4 keep Q+ | Q- side by side
        // (nn > 16 -- groups of 32 lanes, NSTR 34-40 -- keep the two blocks like the groups of 4.  Round 3 sized them for the
        //  packed form: Q- ran into the next group's memory, every layer came out "not positive definite" and went to the
        //  reference-algorithm kernel -- right answers, 47 x slower; found by the round-4 profile of NSTR 40)
        vec = lu + ((nn > 4 && nn <= 16) ? nn * ldq + nn : 2 * nn * ldq);
        // radiance mode adds zjs, z0s, z1s, psi[2n]
        group_total = (vec + (rad ? 5 * n : 0) + 1) & ~1;
        // Lane g of a group reads word g of a row: G consecutive doubles.  ds_read_b64 serves 32 lanes per LDS cycle over
        // 64 dword banks, so the 32 / G groups of a half wave are conflict-free when their bases are 2G dwords apart
        // (mod 64): group_total = G (mod 32) doubles.  (At 98 doubles -- NSTR 16 -- the four groups of a half wave
        // overlapped in 12 of their 16 banks: SQ_LDS_BANK_CONFLICT was 56 % of the LDS-active cycles.)
        int G = 4;
        while (G < nn) G <<= 1;
        if (G <= 16) group_total += ((G - group_total) % 32 + 32) % 32;
        shared_y = 0;                       // Y(l, iq) l-major: [n][nn]
        shared_total = (n * nn + 2 * n + 4 * nn + 1) & ~1;   // + R, 1/(M R), 1/W, 1/M tables
        shared_yu = shared_total;           // radiance mode: Ylm of the block's mode at the user angles, [numu][n] (TERPEV)
        if (rad) shared_total += numu * (n + 2);     // (rows n + 2 apart: TERPSO walks the table a row per lane)
    }
};

// 1/sqrt(x) and 1/x for x > 0 from the hardware seeds and two Newton steps each
SBD_DEVICE double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    return y * (1.5 - 0.5 * x * y * y);
}
// ... and with one Newton step: where only the speed of convergence depends on the value (a Jacobi angle)
SBD_DEVICE double rsqrt_n1(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    return y * (1.5 - 0.5 * x * y * y);
}
SBD_DEVICE double rcp_n1(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return r * (2.0 - x * r);
}
SBD_DEVICE double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    return r * (2.0 - x * r);
}

// x of lane (l ^ S) for 1 <= S <= 15 on the DPP network: quad_perm for S < 4, row_half_mirror
// (l ^ 7) and row_mirror (l ^ 15) composed with a smaller S otherwise
template <int CTRL>
SBD_DEVICE double dpp_move(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int S>
SBD_DEVICE double lane_xor(double x)
{
    static_assert(S >= 1 && S <= 15, "lane_xor: 1..15");
    if constexpr (S == 1) return dpp_move<0xB1>(x);          // quad_perm [1,0,3,2]
    else if constexpr (S == 2) return dpp_move<0x4E>(x);     // quad_perm [2,3,0,1]
    else if constexpr (S == 3) return dpp_move<0x1B>(x);     // quad_perm [3,2,1,0]
    else if constexpr (S == 7) return dpp_move<0x141>(x);    // row_half_mirror
    else if constexpr (S < 7) return dpp_move<0x141>(lane_xor<(S ^ 7)>(x));
    else if constexpr (S == 15) return dpp_move<0x140>(x);   // row_mirror
    else return dpp_move<0x140>(lane_xor<(S ^ 15)>(x));
}

#ifdef SBD_PHASE_TICKS   // developer build: shader-clock ticks per phase, summed over waves (tools/layer_phases.py)
static __device__ unsigned long long layer2_ticks[1024 * 16];   // [block % 1024][counter]: spread, or the atomics are the kernel
#define SBD_TICK(i) tick##i = __builtin_readcyclecounter();
#else
#define SBD_TICK(i)
#endif

// acc + m(lane K of the 16-lane row) * t: the DP-ALU DPP form of the FMA, lanes outside BANKS (one bit per
// four lanes of the row) keep acc
template <int K, int BANKS = 0xF>
SBD_DEVICE double row_fmac(double acc, double m, double t)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:%4"
                 : "+v"(acc) : "v"(m), "v"(t), "n"(K), "n"(BANKS));
    return acc;
}
// acc + m(lane K of this lane's GROUP of G lanes) * t, G = 8 or 16: a row of 16 lanes holds 16 / G groups
template <int K, int G>
SBD_DEVICE double group_fmac(double acc, double m, double t)
{
    static_assert(G == 8 || G == 16, "group_fmac: 8 or 16 lanes per group");
    if constexpr (G == 16) return row_fmac<K>(acc, m, t);
    else return row_fmac<K + 8, 0xC>(row_fmac<K, 0x3>(acc, m, t), m, t);
}
// x of lane K of this lane's group
template <int K, int G>
SBD_DEVICE double group_bcast(double x)
{
    static_assert(G == 8 || G == 16, "group_bcast: 8 or 16 lanes per group");
    int lo = __double2loint(x), hi = __double2hiint(x);
    if constexpr (G == 16) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xF, 0xF, false);
    } else {
        const int l0 = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xF, 0x3, false);
        const int h0 = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xF, 0x3, false);
        lo = __builtin_amdgcn_update_dpp(l0, lo, 0x150 + K + 8, 0xF, 0xC, false);
        hi = __builtin_amdgcn_update_dpp(h0, hi, 0x150 + K + 8, 0xF, 0xC, false);
    }
    return __hiloint2double(hi, lo);
}

// x of lane `src` of this lane's group: a sub-wave shuffle for the power-of-two groups, an absolute lane for the groups
// of 20 (NSTR 34-40) -- the same ds_bpermute_b32 either way
template <int G>
SBD_DEVICE double group_shfl(double x, int src)
{
    if constexpr ((G & (G - 1)) == 0) return __shfl(x, src, G);
    else return __shfl(x, ((int)threadIdx.x / G) * G + src);
}

template <int NN, int G, bool RAD>
#ifndef SBD_RAD_WAVES
#define SBD_RAD_WAVES 2      // waves per SIMD of the intensity variant at NN > 12: a 256-register cap, ~170 spills -- and the NSTR 32
                             // radiance layer kernel 33.6 -> 26.5 ms per 768 points (same-box A/B, round 4; 1: 340 registers, no spill)
#endif
#ifndef SBD_BIG_WAVES
#define SBD_BIG_WAVES 2      // ... and of the flux variant at NN > 16 (groups of 32 lanes)
#endif
__global__ void __launch_bounds__(64, (NN > 16 && !RAD) ? SBD_BIG_WAVES : (NN > 12) ? (RAD ? SBD_RAD_WAVES : 2) : 1) layer_kernel2(Params P, int32_t *eigflag)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int n = 2 * NN, nn = NN, GPB = 64 / G;
    const int lane = threadIdx.x;
    const int g = lane % G, gi = lane / G;
    const int L = P.L, nmode = P.nmode, numu = P.numu;
    // blocks are dealt per azimuth mode (the block shares that mode's Ylm table) over the
    // flattened (item, layer) list of the pass, so that every wave is full whatever L % GPB is
    const long long per_mode = (long long)P.nslot * L;
    const unsigned bpmode = (unsigned)((per_mode + GPB - 1) / GPB);
    const int mazim = (int)(blockIdx.x / bpmode);
    const long long fid = (long long)(blockIdx.x % bpmode) * GPB + gi;
    const bool live = gi < GPB && fid < per_mode;            // (the mode's last block may be partial; groups of 20 lanes leave four)
    const int slot = live ? (int)(fid / L) : 0;
    const int lc = live ? (int)(fid % L) + 1 : L + 1;
    const long long ms = (long long)slot * nmode + mazim;

    const Layer2Lds lds(n, nn, RAD, P.numu);
    double *shy = smem;                                  // shared: Y(l, iq), cmu, cwt
    double *scmu = smem + n * nn, *scwt = scmu + n;
    double *srr = scwt + n, *sxi = srr + nn, *swi = sxi + nn, *smi = swi + nn;   // R = (W/M)^1/2, 1/(M R), 1/W, 1/M
    constexpr bool rad = RAD;
    const int me = g + 1;
    const SV o(L);

    // ---- everything the wave reads from global memory is asked for HERE, in one batch: the block's tables,
    //      the layer's scalars, its moments, the beam's Ylm row.  A load in the middle of the kernel costs a
    //      full memory round trip with nothing to hide it behind (two waves per SIMD), and one issued after
    //      the output stores waits for those as well.  (Partial last block: the idle lanes read slot 0, layer 1.) ----
    const int lcl = live ? lc : 1;
    const int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const double *sv = P.sv + (size_t)slot * P.sv_stride;
    const double *ylmc = P.t.ylmc + (size_t)mazim * n * (n + 1);
    constexpr int NYE = (n * nn + 63) / 64;
    double yent[NYE];
#pragma unroll
    for (int t = 0; t < NYE; ++t) {
        const int e = lane + 64 * t, l = e / nn, iq = e % nn;
        yent[t] = (e < n * nn) ? ylmc[iq * (n + 1) + l] : 0.0;
    }
    const double tcmu = (lane < n) ? P.t.cmu[lane] : 1.0, tcwt = (lane < n) ? P.t.cwt[lane] : 1.0;
    constexpr int NYU = RAD ? 12 : 1;                      // radiance: the user-angle Ylm of this mode, 64 per load
    const double *ylmu_g = P.t.ylmu + (size_t)mazim * numu * (n + 1);
    double yub[NYU];
    if constexpr (RAD) {
#pragma unroll
        for (int t = 0; t < NYU; ++t) {
            const int e = lane + 64 * t;
            yub[t] = (e < numu * n) ? ylmu_g[(e / n) * (n + 1) + e % n] : 0.0;
        }
    }
    const int st0 = svi[SBD_SVI_STATUS];
    const int ncut = svi[SBD_SVI_NCUT];
    const int naz_item = svi[SBD_SVI_NAZ];
    const double fbeam = P.fbeam[slot];
    const bool plank = P.plank[slot] != 0;
    const double oprim = sv[o.oprim() + lcl - 1];
    const double f = sv[o.flyr() + lcl - 1];
    const double dtaucp_lc = sv[o.dtaucp() + lcl - 1];
    const double ssalb_lc = P.ssalb[(size_t)slot * L + (lcl - 1)];
    const double xr0 = sv[o.xr0() + lcl - 1], xr1 = sv[o.xr1() + lcl - 1];
    constexpr int NPK = (n + G - 1) / G;
    double pkv[NPK];
    {
        const double *pm = P.pmom + (pmom_item(P, slot) * L + (lcl - 1)) * (P.nmom + 1);
#pragma unroll
        for (int t = 0; t < NPK; ++t) {
            const int k = g + t * G;
            pkv[t] = (k == 0) ? 1.0 : ((k < n && k <= P.nmom) ? pm[k] : 0.0);
        }
    }
    const double *ylm0 = P.t.ylm0 + (size_t)mazim * (n + 1);
    double y0[n];
#pragma unroll
    for (int k = 0; k < n; ++k) y0[k] = ylm0[k];
    // GC itself is only read back for the layers FLUXES / the boundary rows / USRINT need: the layers of the
    // first two output levels are fetched with the batch, further ones (rare) where the answer is needed
    const int32_t *layru = svi + SBD_SVI_LAYRU;
    const int lay0 = (P.nlev > 0) ? layru[P.t.level_out[0]] : 0;
    const int lay1 = (P.nlev > 1) ? layru[P.t.level_out[1]] : 0;

#pragma unroll
    for (int t = 0; t < NYE; ++t) {
        const int e = lane + 64 * t;
        if (e < n * nn) shy[e] = yent[t];
    }
    if constexpr (rad) {   // TERPEV's table: Ylm(mu_user) of this mode, rows of n (the tables' rows hold n + 1)
        double *syu = smem + lds.shared_yu;
#pragma unroll
        for (int t = 0; t < NYU; ++t) {
            const int e = lane + 64 * t;
            if (e < numu * n) syu[(e / n) * (n + 2) + e % n] = yub[t];
        }
        for (int e = lane + 64 * NYU; e < numu * n; e += 64) syu[(e / n) * (n + 2) + e % n] = ylmu_g[(e / n) * (n + 1) + e % n];   // (more than 24 angles at NSTR 32)
    }
    if (lane < n) { scmu[lane] = tcmu; scwt[lane] = tcwt; }
    if (lane < nn) {
        const double r = sqrt(tcwt / tcmu);
        srr[lane] = r;
        sxi[lane] = 1.0 / (tcmu * r);
        swi[lane] = 1.0 / tcwt;
        smi[lane] = 1.0 / tcmu;
    }
    __syncthreads();
    if (lc > L) return;
    if (st0 & (0x20 | 0x10)) return;
    if (lc > ncut) return;
    if (mazim > naz_item) return;

    double *base = smem + lds.shared_total + (size_t)gi * lds.group_total;
    double *gl = base + lds.gl;
    double *lu = base + lds.lu;
    constexpr bool PACKED = (G == 8 || G == 16);          // (see Layer2Lds)
    double *qp = lu, *qm = lu + nn * lds.ldq;             // Q+ -> L , Q- -> C; packed: qm = C's diagonal
    double *vec = base + lds.vec;                         // radiance mode only:
    double *zjs = vec, *z0s = vec + n, *z1s = vec + 2 * n, *psi = vec + 3 * n;   // psi[2n]
    constexpr int ldq = NN | 1;
    const size_t lidx = (size_t)ms * L + (lc - 1);
#define YS(l, iq) shy[(l) * nn + ((iq) - 1)]             // iq in 1..nn ; Y(l,-mu) = (-1)^(l-m) Y(l,mu)
#define QP(i, j) qp[((j) - 1) * ldq + ((i) - 1)]
    auto qm_at = [&](const int i, const int j) -> double & {   // C(i, j), i >= j
        if constexpr (PACKED) return (i == j) ? qm[i - 1] : lu[(i - 1) * ldq + (j - 1)];
        else return qm[(j - 1) * ldq + (i - 1)];
    };
#define QM(i, j) qm_at((i), (j))

#ifdef SBD_PHASE_TICKS
    unsigned long long tick0 = 0, tick1 = 0, tick2 = 0, tick3 = 0, tick4 = 0, tick5 = 0, tick6 = 0, tick5b = 0;
#endif
    SBD_TICK(0)
    // ---- GL(k) (SETDIS, disort.f:2583-2585) ----
#pragma unroll
    for (int t = 0; t < NPK; ++t) {
        const int k = g + t * G;
        if (k < n) gl[k] = (double)(2 * k + 1) * oprim * (pkv[t] - f) / (1.0 - f);
    }
    wave_lds_sync();

    // ---- the beam source's two halves, r+ + r- and r+ - r- (disort.f:4208-4216): nothing but GL and Ylm ----
    double rs = 0.0, rdv = 0.0;
    if (fbeam > 0.0 && me <= nn) {
        double s0 = 0.0, s1 = 0.0;                       // over even k, over odd k (k >= m by the mask)
#pragma unroll
        for (int k = 0; k < n; k += 2) {
            s0 = s0 + ((k >= mazim) ? gl[k] * YS(k, me) : 0.0) * y0[k];
            s1 = s1 + ((k + 1 >= mazim) ? gl[k + 1] * YS(k + 1, me) : 0.0) * y0[k + 1];
        }
        const bool mpar = (mazim & 1) != 0;
        const double se = mpar ? s1 : s0, so = mpar ? s0 : s1;
        const double c = ((mazim == 0) ? 1.0 : 2.0) * fbeam / (4.0 * P.pi);
        rs = 2.0 * c * se;
        rdv = 2.0 * c * so;
    }

    // ---- S+ / S- (even / odd l-m parts), then Q+-, all symmetric; two Cholesky factorisations side by
    //      side, lower factors, lane i <-> row i ----
    bool spd = true;
    double rp[nn], rm[nn];
    double rdiag = 1.0;                                  // 1 / C(me, me) as the factorisation formed it (G = 8, 16)
    if constexpr (G == 8 || G == 16) {
        // Lane j forms column j = row j of S+- (symmetric); the row stays in registers for the factorisation,
        // whose column k travels by DPP inside the group.  L and C go to LDS once, at the end.
        // (The Ylm table is read from LDS, not from the neighbours' registers: a DPP source lane whose group
        //  has left the kernel -- a layer below the LYRCUT level -- delivers nothing.)
        double yj[n];
#pragma unroll
        for (int l = 0; l < n; ++l) yj[l] = (l >= mazim && me <= nn) ? gl[l] * YS(l, me) : 0.0;   // the sums start at l = m
        const double rj = (me <= nn) ? srr[me - 1] : 0.0;
        const bool mpar = (mazim & 1) != 0;              // l - m even <=> l has m's parity
        static_for<nn>([&](auto qq) {
            constexpr int iq = decltype(qq)::value + 1;
            double s0 = 0.0, s1 = 0.0;                   // over even l, over odd l
#pragma unroll
            for (int l = 0; l < n; l += 2) {
                s0 = s0 + YS(l, iq) * yj[l];
                s1 = s1 + YS(l + 1, iq) * yj[l + 1];
            }
            const double se = mpar ? s1 : s0, so = mpar ? s0 : s1;
            const double ri = srr[iq - 1];
            const double dg = (iq == me) ? swi[iq - 1] : 0.0;
            const bool low = iq <= me && me <= nn;
            rp[iq - 1] = low ? ri * rj * (dg - se) : 0.0;
            rm[iq - 1] = low ? ri * rj * (dg - so) : 0.0;
        });
        SBD_TICK(1)
        static_for<nn>([&](auto kk_) {
            constexpr int k = decltype(kk_)::value + 1;
            // pivots of step k: lane k's diagonal, to the whole group
            const double dp = group_bcast<k - 1, G>(rp[k - 1]), dm = group_bcast<k - 1, G>(rm[k - 1]);
            if (!(dp > 0.0) || !(dm > 0.0)) spd = false;
            const double rdp = rsqrt_nr(dp), rdm = rsqrt_nr(dm);      // 1/sqrt(pivot): the factors are an ulp or two off
            if (me == k) { rp[k - 1] = dp * rdp; rm[k - 1] = dm * rdm; rdiag = rdm; }
            else if (me > k) { rp[k - 1] = rp[k - 1] * rdp; rm[k - 1] = rm[k - 1] * rdm; }
            // row me, column j > k: minus L(me,k) L(j,k), L(j,k) = lane j's entry k (the lanes above row j
            // compute on their unused upper part)
            const double np = -rp[k - 1], nm = -rm[k - 1];
            static_for<nn - k>([&](auto jj_) {
                constexpr int j = k + 1 + decltype(jj_)::value;
                rp[j - 1] = group_fmac<j - 1, G>(rp[j - 1], rp[k - 1], np);
                rm[j - 1] = group_fmac<j - 1, G>(rm[j - 1], rm[k - 1], nm);
            });
        });
        if (me <= nn) {
#pragma unroll
            for (int k = 1; k <= nn; ++k)
                if (k <= me) { QP(me, k) = rp[k - 1]; QM(me, k) = rm[k - 1]; }
        }
    } else {
        if (me <= nn) {
            double yj[n];   // GL(l) * Y(l, mu_j): the j-dependent factor of every term
#pragma unroll
            for (int l = 0; l < n; ++l) yj[l] = (l >= mazim) ? gl[l] * YS(l, me) : 0.0;   // the sums start at l = m
            const double rj = srr[me - 1];
            const bool mpar = (mazim & 1) != 0;              // l - m even <=> l has m's parity
            for (int iq = 1; iq <= nn; ++iq) {
                double s0 = 0.0, s1 = 0.0;                   // over even l, over odd l: plain FMAs
#pragma unroll
                for (int l = 0; l < n; l += 2) {
                    s0 = s0 + YS(l, iq) * yj[l];
                    s1 = s1 + YS(l + 1, iq) * yj[l + 1];
                }
                const double se = mpar ? s1 : s0, so = mpar ? s0 : s1;
                const double ri = srr[iq - 1];
                const double dg = (iq == me) ? swi[me - 1] : 0.0;
                QP(iq, me) = ri * rj * (dg - se);
                QM(iq, me) = ri * rj * (dg - so);
            }
        }
        wave_lds_sync();
        SBD_TICK(1)
        // (row me of both matrices lives in registers while it is being eliminated: the lower triangle
        //  is read once, every column k crosses LDS once for the rows below it)
#pragma unroll
        for (int j = 1; j <= nn; ++j) {
            rp[j - 1] = (me <= nn && j <= me) ? QP(me, j) : 0.0;
            rm[j - 1] = (me <= nn && j <= me) ? QM(me, j) : 0.0;
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 1; k <= nn; ++k) {
            const double dp = group_shfl<G>(rp[k - 1], k - 1), dm = group_shfl<G>(rm[k - 1], k - 1);
            if (!(dp > 0.0) || !(dm > 0.0)) spd = false;
            const double rdp = rsqrt_nr(dp), rdm = rsqrt_nr(dm);
            if (me == k) { rp[k - 1] = dp * rdp; rm[k - 1] = dm * rdm; }
            else if (me > k) { rp[k - 1] = rp[k - 1] * rdp; rm[k - 1] = rm[k - 1] * rdm; }
            if (me >= k && me <= nn) { QP(me, k) = rp[k - 1]; QM(me, k) = rm[k - 1]; }
            wave_lds_sync();
#pragma unroll
            for (int j = k + 1; j <= nn; ++j) {
                if (me >= j) {
                    rp[j - 1] = rp[j - 1] - rp[k - 1] * QP(j, k);
                    rm[j - 1] = rm[j - 1] - rm[k - 1] * QM(j, k);
                }
            }
        }
    }
    wave_lds_sync();
    // group-uniform: hand this layer to the QR kernel when the symmetrised problem is not positive
    // definite, and when a thermal source meets conservative scattering (SSALB = 1, dithered to
    // 1 - 2.2e-14): there I - CC is singular to working precision, the particular solution is an O(1)
    // quantity cancelling against the homogeneous one, and only the reference's own sequence of
    // operations (SGECO/SGESL on the full NSTR x NSTR system) reproduces its rounding (5e-5 vs 3e-4 of
    // the column maximum for the Cholesky-reuse solve below).  Such layers are rare.
    // (within 1024 ulps of 1: the dithered value -- 200 ulps -- and the undithered neighbours of 1, for which the full
    //  I - CC is singular to working precision; the reference-algorithm kernel forms the reference's matrix the
    //  reference's way and raises errmsg 4 on LINPACK's own estimate.  Until round 5 the window was 64 ulps and the
    //  dithered layers were served below: their thermal source came out to 5e-5 of the column maximum only)
    const bool hard_thermal = plank && mazim == 0 && ssalb_lc >= 1.0 - 1024.0 * 1.1102230246251565e-16;
    if (!spd || P.force_fallback || hard_thermal) {
#ifdef SBD_L2_DEBUG
        if (g == 0 && blockIdx.x < 8) printf("layer2 list: block %u group %d reason %s\n", blockIdx.x, gi, !spd ? "not SPD" : "thermal/forced");
#endif
        if (g == 0) eigflag[1 + atomicAdd(&eigflag[0], 1)] = (int32_t)lidx;   // count, then the entries
        return;
    }

    SBD_TICK(2)
    // ---- B = C^T L : lane j <-> column j, in registers ----
    double bcol[nn];
    if (me <= nn) {
        // column me of L, its upper part read as zeros, once into registers
        double lcol[nn];
#pragma unroll
        for (int k = 1; k <= nn; ++k) lcol[k - 1] = (k >= me) ? QP(k, me) : 0.0;
#pragma unroll
        for (int i = 1; i <= nn; ++i) {
            double s = 0.0;
            // (C^T L)(i,j) = sum_{k >= max(i,j)} C(k,i) L(k,j): static k >= i, zeros of L below j
#pragma unroll
            for (int k = i; k <= nn; ++k) s = s + QM(k, i) * lcol[k - 1];
            bcol[i - 1] = s;
        }
    } else {
#pragma unroll
        for (int i = 0; i < nn; ++i) bcol[i] = 0.0;
    }

    SBD_TICK(3)
#ifdef SBD_PHASE_TICKS
    int nsweep = 0;
#endif
    // ---- one-sided Jacobi, round-robin pairing over the nn columns (nn even or odd) ----
    {
        constexpr int NP = (nn + 1) & ~1;               // players (a dummy when nn is odd)
        constexpr bool XORS = (NP & (NP - 1)) == 0 && NP <= 16;   // power of two: partner = j ^ s on the DPP network
        const int j = g;                                 // player index 0..NP-1 (lanes >= NP idle)
        const double tol = 2.220446049250313e-16;
        bool rotated = false, coarse = false;
        // Convergence is decided per GROUP (the G lanes of one layer), not per wave: a group that is
        // done stops rotating whatever its wave-mates still do, so an item's result does not depend on
        // which other items share its wave (batch composition, pass size, L % GPB)
        bool done = false;
        const unsigned long long gmask = ((G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull)) << ((gi * G) & 63);
        // one meeting of this lane's column with its partner's (both lanes run it, each keeps its own)
        // (only B's columns rotate: the eigenvectors follow from the converged columns and C afterwards)
        // |b_j|^2 of this lane's column, exact at the start of every sweep, carried through the rotations
        // (alpha' = alpha - t gamma, beta' = beta + t gamma): a meeting costs one inner product, not three
        double nrm = 0.0, wsc = 1.0;
        // Scaled rotations (the fast-Givens idea carried to one-sided Jacobi): lane j keeps its column as
        // bcol = w_j^1/2 x (the true column) and the squared scale w_j beside it.  With a = |p|^2, b = |q|^2,
        // g = p.q of the STORED columns, the pair is rotated as  p' = p - tau q,  q' = q + tau' p  with
        //   e = w_p b - w_q a,  H = (e^2 + 4 g^2 w_p w_q)^1/2,  u = 2 g / (|e| + H),
        //   tau = sign(e) u w_p,  tau' = sign(e) u w_q,   both scales x (1 + tau tau')  [tau tau' = tan^2],
        // which is the exact rotation of the true pair (tan = tau (w_q / w_p)^1/2: the quadratic for tau has no
        // square root of the scales in it) -- no c = (1 + t^2)^-1/2 (a v_rsq_f64 and two Newton steps per
        // meeting), one FMA per element instead of a product and an FMA.  Whatever the rounding of u, the
        // transformation stays orthogonal x diagonal to working precision: u only has to be close (one Newton
        // step on the hardware seeds: the pair comes out orthogonal to ~1e-8 of what it was, the next sweep sees
        // to the rest).  cos^2 = g^2 / (a b) does not see the scales; they leave once, after the last sweep.
        // In both lanes of a pair the own coefficient is  -sign(w_own b_other - w_other a_own) u w_own.
        auto meet = [&](const int partner, const bool valid, const double (&ob)[nn], const double bb, const double wb) {
            if (valid && !done) {
                const double aa = nrm, wa = wsc;
                double gg = 0.0;
#pragma unroll
                for (int i = 0; i < nn; ++i) gg = gg + bcol[i] * ob[i];
                const double ab = aa * bb, g2 = gg * gg;
                if (g2 > tol * tol * ab) {           // |cos(angle)| > tol
                    rotated = true;
                    coarse = coarse || (g2 > 1.0e-13 * ab);   // ... > 3e-7
                    // e of the ordered pair: both lanes must see the SAME magnitude with opposite signs, or both would take
                    // the same branch below and the two columns come out parallel.  Two rounded products and a difference
                    // (no contraction: the partner forms the same two products and subtracts them the other way round,
                    // bit for bit -e); an exact tie, e = +0 in both lanes, goes to the lane order
                    double e;
                    {
#pragma clang fp contract(off)
                        const double p1 = wa * bb, p2 = wb * aa;
                        e = p1 - p2;
                    }
                    const double ww = wa * wb, tg = 2.0 * gg;
                    const double h2 = e * e + (tg * tg) * ww;
                    const double h = h2 * rsqrt_n1(h2);
                    const double u = tg * rcp_n1(fabs(e) + h);
                    const double q1 = 1.0 + (u * u) * ww;
                    const double coef = ((e > 0.0 || (e == 0.0 && j < partner)) ? -u : u) * wa;
#pragma unroll
                    for (int i = 0; i < nn; ++i) bcol[i] = bcol[i] + coef * ob[i];
                    nrm = q1 * (aa + coef * gg);
                    wsc = wa * q1;
                }
            }
        };
        for (int sweep = 0; sweep < 30; ++sweep) {
            rotated = false;
            coarse = false;          // some pair met in this sweep with |cos(angle)| > 3e-7
            if (wsc > 0x1p+64) {             // (a scale doubles at most per meeting: rare; the lane's own business)
                const double rs = rsqrt_nr(wsc);
#pragma unroll
                for (int i = 0; i < nn; ++i) bcol[i] = bcol[i] * rs;
                wsc = 1.0;
            }
            nrm = 0.0;
#pragma unroll
            for (int i = 0; i < nn; ++i) nrm = nrm + bcol[i] * bcol[i];
            if constexpr (XORS) {
                // every pair (j, j ^ s), s = 1..NP-1, meets once per sweep; the exchange is one or two
                // DPP moves per dword (quad_perm / row_half_mirror / row_mirror), no LDS round trip
                static_for<NP - 1>([&](auto ss) {
                    constexpr int sx = decltype(ss)::value + 1;
                    const int partner = j ^ sx;
                    double ob[nn];
#pragma unroll
                    for (int i = 0; i < nn; ++i) ob[i] = lane_xor<sx>(bcol[i]);
                    meet(partner, (j < nn) && (partner < nn), ob, lane_xor<sx>(nrm), lane_xor<sx>(wsc));
                });
            } else {
                for (int s = 0; s < NP - 1; ++s) {
                    // circle method: player 0 fixed, the others rotate
                    int pos = (j == 0) ? 0 : 1 + (j - 1 - s + 2 * (NP - 1)) % (NP - 1);
                    const int ppos = NP - 1 - pos;
                    const int partner = (ppos == 0) ? 0 : 1 + (ppos - 1 + s) % (NP - 1);
                    const int src = (j < NP) ? partner : j;
                    double ob[nn];
#pragma unroll
                    for (int i = 0; i < nn; ++i) ob[i] = group_shfl<G>(bcol[i], src);
                    meet(partner, (j < nn) && (partner < nn) && (j < NP), ob, group_shfl<G>(nrm, src), group_shfl<G>(wsc, src));
                }
            }
            // quadratic convergence: a sweep that started below 3e-7 ends below 1e-13 (eigenvalues to
            // ~1e-26 relative, eigenvectors to ~1e-13: far inside the parity gate)
            if ((__ballot(rotated) & gmask) == 0ull || (__ballot(coarse) & gmask) == 0ull) done = true;
#ifdef SBD_PHASE_TICKS
            ++nsweep;
#endif
            if (!__any(!done)) break;
        }
        if (!done) {   // 30 sweeps without convergence: the reference-algorithm kernel redoes this layer
#ifdef SBD_L2_DEBUG
            if (g == 0 && blockIdx.x < 8) printf("layer2 list: block %u group %d reason no convergence\n", blockIdx.x, gi);
#endif
            if (g == 0) eigflag[1 + atomicAdd(&eigflag[0], 1)] = (int32_t)lidx;
            return;
        }
        {   // the columns' scales leave: b' = B v with |v| = 1 from here on
            const double rs = rsqrt_nr(wsc);
#pragma unroll
            for (int i = 0; i < nn; ++i) bcol[i] = bcol[i] * rs;
        }
    }

    SBD_TICK(4)
    // ---- eigenvalues, eigenvectors, (G+)+(G-) = AMB x / k (disort.f:3264-3286), outputs ----
    // The converged column is b' = B v = C^T L v (|b'| = k).  With y = L v = C^-T b':
    //   x = (M R)^-1 y                                   (G+)-(G-), one back-substitution with C^T;
    //   AMB x = -M^-1 (I - S- W) x = -(M R)^-1 Q- (R^-1 W x), and R^-1 W x = W/(M R^2) y = y, Q- y = C C^T y = C b':
    //   (G+)+(G-) = -(M R)^-1 (C b') / k                  one product with C.
    // C is in LDS since the factorisation; nothing but B's columns went through the rotations.
    double kq = 0.0, lam = 1.0;
    double gp[rad ? nn : 1], xcol[rad ? nn : 1];         // radiance mode (TERPEV) walks them many times
    double rkq_me = 0.0;
    double yv[nn], cb[nn];                               // y = L v and C b' = Q- y: UPISOT / UPBEAM below reuse them
#pragma unroll
    for (int i = 0; i < nn; ++i) { yv[i] = 0.0; cb[i] = 0.0; }
    if (me <= nn) {
        lam = 0.0;
#pragma unroll
        for (int i = 0; i < nn; ++i) lam = lam + bcol[i] * bcol[i];
    }
    // The particular solutions below come from these singular vectors.  With a beam, a k_j next to 1/mu0 makes
    // the system UPBEAM solves singular (disort.f:4227): the reference-algorithm kernel redoes such a layer
    // (and raises the reference's warning from its pivots).  "Next to" is 1e-10 relative: the expansion below
    // and the reference's LU lose the same eps / gap there, and a wider window costs real time -- among the
    // 4 M layers of a 131 k-solve sweep a few eigenvalues fall within 1e-6 of 1/mu0 every time, and ONE layer
    // in the reference-algorithm kernel holds its pass's stream for 0.9 ms.
    // (A tiny k_j -- conservative scattering, SSALB dithered to 1 - 2.2e-14, k ~ 2e-7 -- is served here: its
    //  column b'_j carries the absolute accuracy eps |B| of the rotations, i.e. ~1e-9 relative, in the
    //  eigenvector as in the particular solution; the reference's own k for that mode is good to ~1e-3.)
    const double umu0 = P.umu0;
    {
        const double gap = fabs(1.0 - umu0 * umu0 * lam);
        const bool bad = (me <= nn) && (!(lam > 0.0) || (fbeam > 0.0 && !(gap > 1.0e-10)));
        const unsigned long long gmask = ((G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull)) << ((gi * G) & 63);
        if ((__ballot(bad) & gmask) != 0ull) {
#ifdef SBD_L2_DEBUG
            if (bad && blockIdx.x < 8) printf("layer2 list: block %u group %d lane %d reason lam %g gap %g\n", blockIdx.x, gi, g, lam, gap);
#endif
            if (g == 0) eigflag[1 + atomicAdd(&eigflag[0], 1)] = (int32_t)lidx;
            return;
        }
    }
    if (me <= nn) {
        const double rkq = rsqrt_nr(fabs(lam));           // (lam > 0 here; k to an ulp or two, like the factors)
        kq = fabs(lam) * rkq;
        // one walk over C, column by column from the last: entry C(k,i) serves the back-substitution
        // C^T y = b' (row i) and the product C b' (row k) -- read once, and never more than a column in flight
        static_for<nn>([&](auto ii) {
            constexpr int i = nn - decltype(ii)::value;
            double s = bcol[i - 1];
            const double bi = bcol[i - 1];
            const double cii = QM(i, i);
#pragma unroll
            for (int k = i + 1; k <= nn; ++k) {
                const double c = QM(k, i);
                s = s - c * yv[k - 1];
                cb[k - 1] = cb[k - 1] + c * bi;
            }
            cb[i - 1] = cb[i - 1] + cii * bi;
            // (1 / C(i,i): lane i kept it from the factorisation -- two DPP moves instead of a v_rcp_f64 and two
            //  Newton steps per row)
            if constexpr (G == 8 || G == 16) yv[i - 1] = s * group_bcast<i - 1, G>(rdiag);
            else yv[i - 1] = s * rcp_nr(cii);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (rad) {
#pragma unroll
            for (int i = 0; i < nn; ++i) gp[i] = -(cb[i] * sxi[i]) * rkq;
        }
        if constexpr (rad) {
#pragma unroll
            for (int i = 0; i < nn; ++i) xcol[i] = yv[i] * sxi[i];
        }
        rkq_me = rkq;
        double *kkout = P.kk + lidx * n;
        double *ekout = P.ek + lidx * nn;
        kkout[me + nn - 1] = kq;
        kkout[nn + 1 - me - 1] = -kq;
        const double ekv = exp(-kq * dtaucp_lc);
        ekout[nn + 1 - me - 1] = ekv;
        // GC (disort.f:3290-3312) and the matrix-ready interface blocks (disort.f:2851-2876)
        // straight from registers: this lane owns columns me+nn (k > 0) and nn+1-me (k < 0) of
        // every row, nn lanes write nn consecutive doubles per instruction.  GC itself is only
        // read back for the layers FLUXES / the boundary rows / USRINT need (USRINT: the last layer's, for the
        // surface term; until round 4 the radiance variant wrote all of them -- 8 KB per layer and mode at NSTR 32,
        // a third of this kernel's stores, read by nobody).
        // (gconly: sbd_band4.hpp scales GC by the STWJ factors itself and only needs GC's two independent
        //  quarters, Params::gcc -- a quarter of the bytes of ga + gb)
        bool need_gc = P.all_levels || lc == 1 || lc == ncut || (rad && lc == L) || lay0 == lc || lay1 == lc;
        if constexpr (rad) {
            // (a third output level without all of them does not occur under the Fortran host: then every layer's GC is
            //  written as before -- the loop over the levels below costs this variant 130 more spilled registers)
            need_gc = need_gc || P.nlev > 2;
        } else if (!need_gc) {
            for (int i = 2; i < P.nlev; ++i) need_gc = need_gc || layru[P.t.level_out[i]] == lc;
        }
        double *gcout = P.gc + lidx * n * n;
        const int ja = me + nn - 1, jb = nn - me;          // 0-based columns
        if (P.gconly) {
            double *cc0 = P.gcc + lidx * 2 * nn * nn + (me - 1), *cc1 = cc0 + nn * nn;
#pragma unroll
            for (int iq = 1; iq <= nn; ++iq) {
                const double gpp = -(cb[iq - 1] * sxi[iq - 1]) * rkq_me, gmm = yv[iq - 1] * sxi[iq - 1];
                const double vua = 0.5 * (gpp + gmm), vda = 0.5 * (gpp - gmm);   // rows iq+nn, nn+1-iq of column ja
#ifdef SBD_AB_NO_GC_STORES    // developer A/B (wrong results): GC's quarters -- 33 of the layer kernel's 57 KB per solve -- are not written
                if (P.nslot < 0) { cc0[(iq - 1) * nn] = vua; cc1[(iq - 1) * nn] = vda; }
#else
                cc0[(iq - 1) * nn] = vua;
                cc1[(iq - 1) * nn] = vda;
#endif
                if (need_gc) {
                    const int ru = (iq + nn - 1) * n, rd = (nn - iq) * n;
                    gcout[ru + ja] = vua;  gcout[rd + ja] = vda;
                    gcout[ru + jb] = -vda; gcout[rd + jb] = -vua;
                }
            }
        } else {
            double *gaout = P.ga + lidx * n * n, *gbout = P.gb + lidx * n * n;
#pragma unroll
            for (int iq = 1; iq <= nn; ++iq) {
                const double gpp = -(cb[iq - 1] * sxi[iq - 1]) * rkq_me, gmm = yv[iq - 1] * sxi[iq - 1];
                const double vua = 0.5 * (gpp + gmm), vda = 0.5 * (gpp - gmm);   // rows iq+nn, nn+1-iq of column ja
                const int ru = (iq + nn - 1) * n, rd = (nn - iq) * n;
                gaout[ru + ja] = vua * ekv;  gbout[ru + ja] = -vua;
                gaout[rd + ja] = vda * ekv;  gbout[rd + ja] = -vda;
                gaout[ru + jb] = -vda;       gbout[ru + jb] = vda * ekv;         // column jb: -(gpp - gmm)/2, -(gpp + gmm)/2
                gaout[rd + jb] = -vua;       gbout[rd + jb] = vua * ekv;
                if (need_gc) {
                    gcout[ru + ja] = vua;  gcout[rd + ja] = vda;
                    gcout[ru + jb] = -vda; gcout[rd + jb] = -vua;
                }
            }
        }
    }
    wave_lds_sync();      // (L and C stay in the lu area until UPBEAM builds its matrix there)
    SBD_TICK(5)

    // ---- radiance mode: TERPEV from the register-resident eigenvector columns ----
    if constexpr (rad) {
        // Until round 4 this block was three quarters of the radiance variant's time (tools/layer_phases_rad.py:
        // 236 k of 319 k cycles per wave at NSTR 32): the user-angle Ylm came from global memory one value -- one
        // memory round trip -- at a time, the quadrature-angle Ylm from LDS the same way.  Now: the user-angle table of
        // the block's mode waits in LDS (staged with the first batch of loads), a row of quadrature-angle values is read
        // as one batch, the parity of l - m is a uniform branch instead of a multiplication.  Every sum is formed from
        // the same operands in the same order as before.
        const double *syu = smem + lds.shared_yu;
        double *guout = P.gu + lidx * n * numu;
        if (me <= nn) {
            // EVECC column me (k>0) and me+nn (k<0): rows iq<=nn / iq>nn
            double e11[nn], e21[nn];
#pragma unroll
            for (int i = 0; i < nn; ++i) { e11[i] = 0.5 * (gp[i] + xcol[i]); e21[i] = 0.5 * (gp[i] - xcol[i]); }
            double wkp[n], wkn[n];
            static_for<n>([&](auto ll) {
                constexpr int l = decltype(ll)::value;
                double sp_ = 0.0, sn_ = 0.0;
                if (l >= mazim) {
                    double y[nn];
#pragma unroll
                    for (int jq = 0; jq < nn; ++jq) y[jq] = YS(l, jq + 1) * scwt[jq];
                    if (((l - mazim) & 1) == 0) {                      // Y(l,-mu) = +Y(l,mu)
#pragma unroll
                        for (int jq = 0; jq < nn; ++jq) {
                            sp_ = sp_ + y[jq] * e11[jq] + 1.0 * y[jq] * e21[jq];          // column me
                            sn_ = sn_ + y[jq] * (-e21[jq]) + 1.0 * y[jq] * (-e11[jq]);    // column me+nn
                        }
                    } else {                                           // Y(l,-mu) = -Y(l,mu)
#pragma unroll
                        for (int jq = 0; jq < nn; ++jq) {
                            sp_ = sp_ + y[jq] * e11[jq] + (-y[jq]) * e21[jq];
                            sn_ = sn_ + y[jq] * (-e21[jq]) + (-y[jq]) * (-e11[jq]);
                        }
                    }
                    sp_ = 0.5 * gl[l] * sp_;
                    sn_ = 0.5 * gl[l] * sn_;
                }
                wkp[l] = sp_;
                wkn[l] = sn_;
            });
            // (the terms l < m are +0 x Ylm = +-0 added to a sum that starts at +0: they leave it as it is, so whole
            //  blocks of eight l are taken or skipped)
            for (int iu = 1; iu <= numu; ++iu) {
                const double *yrow = syu + (iu - 1) * (n + 2);
                double s1 = 0.0, s2 = 0.0;
                static_for<(n + 7) / 8>([&](auto bb) {
                    constexpr int l0 = 8 * decltype(bb)::value;
                    if (l0 + 7 >= mazim) {
                        double yu[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) yu[t] = (l0 + t < n) ? yrow[l0 + t] : 0.0;
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            if (l0 + t < n) {
                                s1 = s1 + wkp[l0 + t] * yu[t];
                                s2 = s2 + wkn[l0 + t] * yu[t];
                            }
                        }
                    }
                });
                guout[(me + nn - 1) * numu + (iu - 1)] = s1;        // IQ = me      -> GU(iu, me+nn)
                guout[(nn + 1 - me - 1) * numu + (iu - 1)] = s2;    // IQ = me+nn   -> GU(iu, n+1-(me+nn))
            }
        }
    }

    SBD_TICK(5b)
    // ---- UPBEAM / UPISOT on the +-mu-reduced systems (see the header), from the singular vectors at hand.
    //      With Y = [y_j] (y_j = L v_j), CB' = [C b'_j] = Q- Y and K = diag(k_j):
    //        Y Y^T = Q+,   Y^T (CB') = K^2,   Q+^-1 = (CB') K^-4 (CB')^T,   Q- = (CB') K^-2 (CB')^T,
    //        I - S+ W = R^-1 Q+ R^-1 W,   I - S- W = R^-1 Q- R^-1 W,   M = R^-2 W,
    //        T = R^-1 (I/mu0 - mu0 Q+ Q-) R^-1 W,   Q+ Q- = Y K^2 Y^-1,   Y^-1 = K^-2 (CB')^T.
    //      Every solve is a projection on the columns (lane j <-> column j) and a recombination -- two LDS
    //      round trips -- instead of NSTR/2 dependent elimination steps; S+ and S- themselves are not kept. ----
    auto ylmc_full = [&](int l, int iq) -> double {   // YLMC(l, iq) including the mirrored half
        if (iq <= nn) return YS(l, iq);
        return ((((l - mazim) & 1) == 0) ? 1.0 : -1.0) * YS(l, iq - nn);
    };
    double *scra = lu;                                   // L and C are dead from here on
    // (sum_j y_j[i] t_j, sum_j cb_j[i] t_j) delivered to lane i
    auto combine1 = [&](const double (&a)[nn], const double tj) -> double {
        if (me <= nn) {
#pragma unroll
            for (int i = 0; i < nn; ++i) scra[i * ldq + (me - 1)] = a[i] * tj;
        }
        wave_lds_sync();
        double r = 0.0;
        if (me <= nn) {
#pragma unroll
            for (int m = 0; m < nn; ++m) r = r + scra[(me - 1) * ldq + m];
        }
        wave_lds_sync();
        return r;
    };
    auto combine2 = [&](const double tj, double &ry, double &rc_) {
        ry = combine1(yv, tj);
        rc_ = combine1(cb, tj);
    };
    const bool thermal = plank && mazim == 0;
    int status = 0;
    // errmsg 4 (UPISOT's SGECO, disort.f:4333): the Cholesky pivots of Q+ and Q- (squares of the factors' diagonals,
    // still in LDS) are the FILTER (near_singular() in sbd_layer.hpp): a thermal layer whose pivots span ten orders of
    // magnitude goes to the reference-algorithm kernel, which decides on LINPACK's own estimate
    if (thermal) {
        const double dl = (me <= nn) ? QP(me, me) : 0.0, dc = (me <= nn) ? QM(me, me) : 0.0;
        const bool suspect = near_singular<G>((me <= nn) ? dl * dl : -1.0, 1.0e-10) || near_singular<G>((me <= nn) ? dc * dc : -1.0, 1.0e-10);
        if (suspect) {
            if (g == 0) eigflag[1 + atomicAdd(&eigflag[0], 1)] = (int32_t)lidx;
            return;
        }
        wave_lds_sync();                                 // (the pivots were read from the area that now turns scratch)
    }
    const double rme = (me <= nn) ? srr[me - 1] : 1.0, rw = (me <= nn) ? srr[me - 1] * swi[me - 1] : 0.0;
    const double rk2 = rkq_me * rkq_me;                  // 1 / k^2 (idle lanes: 0, their sums are empty)
    // reciprocals from the block's tables instead of IEEE divisions (each ~17 instructions): 1 / R = mu X, 1 / mu
    const double rrme = (me <= nn) ? scmu[me - 1] * sxi[me - 1] : 1.0, rcmu_me = (me <= nn) ? smi[me - 1] : 1.0;
    if (thermal) {
        // UPISOT (disort.f:4309-4349).  (I - CC) Z1 = (1-w') XR1 (the same in both halves): Z1+- = (1-w') XR1 u,
        // (I - S+ W) u = 1;  (I - CC) Z0 = (1-w') XR0 + CMU Z1: Z0+- = (1-w') XR0 u +- e, (I - S- W) e = mu Z1.
        //   u = W^-1 R Q+^-1 (R 1) = W^-1 R sum_j cb_j t_j,  t_j = (cb_j . R 1) / k_j^4
        //   e = W^-1 R Q-^-1 (R M Z1); R M Z1 = (1-w') XR1 R M W^-1 R Q+^-1 (R 1) = (1-w') XR1 Q+^-1 (R 1)  (R^2 M = W), and
        //   Q-^-1 = Y K^-2 Y^T, Y^T (CB') = K^2:   e = (1-w') XR1 W^-1 R sum_j y_j t_j   -- the same coefficients
        double d1 = 0.0;
#pragma unroll
        for (int i = 0; i < nn; ++i) d1 = d1 + cb[i] * srr[i];
        double vy, vc;
        combine2(d1 * rk2 * rk2, vy, vc);
        const double u = rw * vc;
        const double z1 = (1.0 - oprim) * xr1 * u;
        const double e = (1.0 - oprim) * xr1 * rw * vy;
        const double z0p = (1.0 - oprim) * xr0 * u + e, z0m = (1.0 - oprim) * xr0 * u - e;
        if (me <= nn) {
            double *p0 = P.zp0 + lidx * n, *p1 = P.zp1 + lidx * n;
            p0[me + nn - 1] = z0p; p1[me + nn - 1] = z1;
            p0[nn - me] = z0m;     p1[nn - me] = z1;
            if constexpr (rad) { z0s[me - 1] = z0p; z0s[me + nn - 1] = z0m; z1s[me - 1] = z1; z1s[me + nn - 1] = z1; }
        }
    } else if (mazim == 0 && me <= nn) {
        P.zp0[lidx * n + me - 1] = 0.0; P.zp0[lidx * n + me + nn - 1] = 0.0;
        P.zp1[lidx * n + me - 1] = 0.0; P.zp1[lidx * n + me + nn - 1] = 0.0;
    }

    SBD_TICK(6)
    if (fbeam > 0.0) {
        // UPBEAM (disort.f:4130-4244) for s = Z+ + Z-, d = Z+ - Z-:
        //   q = (r+ + r-) - mu0 (I - S+ W) M^-1 (r+ - r-);  T d = q;  s = mu0 M^-1 ((r+ - r-) - (I - S- W) d).
        // With p = R^-1 W M^-1 (r+ - r-) and a = R (r+ + r-):  (CB')^T R q = (CB')^T a - mu0 K^2 Y^T p, so
        //   c_j = (cb_j . a - mu0 k_j^2 y_j . p) / (k_j^2 (1/mu0 - mu0 k_j^2)),
        //   d = W^-1 R sum_j y_j c_j,   (I - S- W) d = R^-1 (CB') K^-2 (CB')^T Y c = R^-1 sum_j cb_j c_j.
        double *spa = scra, *spp = scra + nn;            // the two spread vectors, read before the products land
        if (me <= nn) {
            spa[me - 1] = rme * rs;
            spp[me - 1] = (scwt[me - 1] * rrme) * (rdv * rcmu_me);
        }
        wave_lds_sync();
        double pa = 0.0, pp = 0.0;
#pragma unroll
        for (int i = 0; i < nn; ++i) {
            pa = pa + cb[i] * spa[i];
            pp = pp + yv[i] * spp[i];
        }
        wave_lds_sync();
        const double cj = (pa - umu0 * lam * pp) * rcp_nr(lam * (P.rumu0 - umu0 * lam));
        double vy, vc;
        combine2(cj, vy, vc);
        const double dv = rw * vy;
        const double sv_ = umu0 * (rdv - vc * rrme) * rcmu_me;
        const double zpl = 0.5 * (sv_ + dv), zmi = 0.5 * (sv_ - dv);     // Z(+mu_me), Z(-mu_me)
        if (me <= nn) {
            double *zzout = P.zz + lidx * n;
            zzout[me + nn - 1] = zpl;                    // ZZ(nn+iq) <- ZJ(iq), ZZ(nn+1-iq) <- ZJ(iq+nn) (disort.f:4238-4241)
            zzout[nn - me] = zmi;
            if constexpr (rad) { zjs[me - 1] = zpl; zjs[me + nn - 1] = zmi; }
        }
        wave_lds_sync();
    } else if (me <= nn) {
        P.zz[lidx * n + me - 1] = 0.0;
        P.zz[lidx * n + me + nn - 1] = 0.0;
    }
    // ---- TERPSO (disort.f:3980-4128) ----
    if constexpr (rad) {
        // (the user-angle Ylm from TERPEV's LDS table, the beam's Ylm from the registers of the first load batch, eight
        //  terms per batch of LDS reads; until round 4 a global load per term.  Same sums, same order.)
        const double *syu = smem + lds.shared_yu;
        double *zbout = P.zb + lidx * numu, *z0uout = P.z0u + lidx * numu, *z1uout = P.z1u + lidx * numu;
        double *psi0 = psi, *psi1 = psi + n;
        if (fbeam > 0.0) {
            const double delm0 = (mazim == 0) ? 1.0 : 0.0;
            for (int l = g; l <= n - 1; l += G) {
                if (l < mazim) continue;
                double psum = 0.0;
                for (int jq = 1; jq <= n; ++jq) psum = psum + scwt[jq - 1] * ylmc_full(l, jq) * zjs[jq - 1];
                psi0[l] = 0.5 * gl[l] * psum;
            }
            wave_lds_sync();
            const double fact = (2.0 - delm0) * fbeam / (4.0 * P.pi);
            for (int iu = me; iu <= numu; iu += G) {
                const double *yrow = syu + (iu - 1) * (n + 2);
                double sum = 0.0;
                static_for<(n + 7) / 8>([&](auto bb) {
                    constexpr int l0 = 8 * decltype(bb)::value;
                    if (l0 + 7 >= mazim) {
                        double yu[8], ps[8], gv[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const int iq = (l0 + t < n) ? l0 + t : n - 1;
                            yu[t] = yrow[iq]; ps[t] = (iq >= mazim) ? psi0[iq] : 0.0; gv[t] = gl[iq];
                        }
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            if (l0 + t < n) {
                                const double nx = sum + yu[t] * (ps[t] + fact * gv[t] * y0[l0 + t]);
                                sum = (l0 + t >= mazim) ? nx : sum;
                            }
                        }
                    }
                });
                zbout[iu - 1] = sum;
            }
            wave_lds_sync();
        } else {
            for (int iu = me; iu <= numu; iu += G) zbout[iu - 1] = 0.0;
        }
        if (thermal) {
            for (int l = g; l <= n - 1; l += G) {
                double psum0 = 0.0, psum1 = 0.0;
                for (int jq = 1; jq <= n; ++jq) {
                    const double y = scwt[jq - 1] * ylmc_full(l, jq);
                    psum0 = psum0 + y * z0s[jq - 1];
                    psum1 = psum1 + y * z1s[jq - 1];
                }
                psi0[l] = 0.5 * gl[l] * psum0;
                psi1[l] = 0.5 * gl[l] * psum1;
            }
            wave_lds_sync();
            for (int iu = me; iu <= numu; iu += G) {
                const double *yrow = syu + (iu - 1) * (n + 2);
                double sum0 = 0.0, sum1 = 0.0;
#pragma unroll
                for (int iq = 0; iq <= n - 1; ++iq) {
                    const double yu = yrow[iq];
                    sum0 = sum0 + yu * psi0[iq];
                    sum1 = sum1 + yu * psi1[iq];
                }
                z0uout[iu - 1] = sum0 + (1.0 - oprim) * xr0;
                z1uout[iu - 1] = sum1 + (1.0 - oprim) * xr1;
            }
        } else if (mazim == 0) {
            for (int iu = me; iu <= numu; iu += G) { z0uout[iu - 1] = 0.0; z1uout[iu - 1] = 0.0; }
        }
    }
    if (status && g == 0) atomicOr(&P.svi[(size_t)slot * P.svi_stride + SBD_SVI_STATUS], status);
#ifdef SBD_PHASE_TICKS
    {
        const unsigned long long tick7 = __builtin_readcyclecounter();
        if (lane == 0) {
            unsigned long long *tk = layer2_ticks + (blockIdx.x & 1023u) * 16;
            atomicAdd(&tk[0], tick1 - tick0); atomicAdd(&tk[1], tick2 - tick1);
            atomicAdd(&tk[2], tick3 - tick2); atomicAdd(&tk[3], tick4 - tick3);
            atomicAdd(&tk[4], tick5 - tick4); atomicAdd(&tk[5], tick6 - tick5);
            atomicAdd(&tk[6], tick7 - tick6); atomicAdd(&tk[7], 1ull);
            atomicAdd(&tk[8], (unsigned long long)nsweep);
            atomicAdd(&tk[9], tick5b - tick5);            // (radiance variant: TERPEV, part of counter 5)
        }
    }
#endif
#undef YS
#undef QP
#undef QM
}

}  // namespace sbd
