// Radiance kernels.
//
// usrint_kernel: azimuthal component UUM(iu, lev) of the intensity at the user angles by
// analytic integration of the source function layer by layer (USRINT, disort.f:4355-4793),
// one wave per (work item, mode), one lane per (output level, user angle) pair so that the
// sums over layers and streams run in the reference's order.  Only the requested output
// levels are evaluated (SBDART prints two of DISORT's NLYR+1 levels, drt.f:996-1006).
// The exp(KK*dtau) factors are the STWJ factors the layer kernel already produced.
//
// azimuth_kernel: Fourier cosine series UU = sum_m UUM_m cos(m (phi - phi0))
// (disort.f:767-825) including the reference's two-in-a-row convergence exit (ACCUR = 0
// in SBDART, drt.f:142), one lane per (level, angle), modes summed in order.
#pragma once
#include "sbd_common.hpp"
#include "sbd_surface.hpp"

namespace sbd {

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) usrint_kernel(Params P)
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
    const int L = P.L, n = P.n, nn = P.nn, numu = P.numu, nlev = P.nlev;
    const int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const int st0 = svi[SBD_SVI_STATUS];
    const double fbeam = P.fbeam[slot];
    double *uum = P.uum + (size_t)ms * nlev * numu;
    const bool dead = (st0 & (0x20 | 0x10 | 0x08)) != 0;
    if (dead || mazim > svi[SBD_SVI_NAZ]) {
        for (int i = lane; i < nlev * numu; i += 64) uum[i] = 0.0;
        return;
    }
    const int ncut = svi[SBD_SVI_NCUT];
    const bool lyrcut = svi[SBD_SVI_LYRCUT] != 0;
    const bool plank = P.plank[slot] != 0;
    const int32_t *layru = svi + SBD_SVI_LAYRU;
    const SV o(L);
    const double *sv = P.sv + (size_t)slot * P.sv_stride;
    const double *taucpr = sv + o.taucpr(), *dtaucp = sv + o.dtaucp(), *expbea = sv + o.expbea();
    const double *utaupr = sv + o.utaupr();
    const double bplank = sv[o.bplank()], tplank = sv[o.tplank()];
    const double albedo = P.albedo[slot];
    const double delm0 = (mazim == 0) ? 1.0 : 0.0;
    const double umu0 = P.umu0, pi = P.pi;
    const double *cmu = P.t.cmu, *cwt = P.t.cwt, *umu = P.t.umu;
    const bool beam = fbeam > 0.0;
    const bool therm = plank && mazim == 0;

    const double *gc = P.gc + (size_t)ms * L * n * n;
    const double *kk = P.kk + (size_t)ms * L * n;
    const double *ek = P.ek + (size_t)ms * L * nn;
    const double *zz = P.zz + (size_t)ms * L * n;
    const double *zp0 = P.zp0 + (size_t)(ms - mazim) * L * n;
    const double *zp1 = P.zp1 + (size_t)(ms - mazim) * L * n;
    const double *ll = P.ll + (size_t)ms * L * n;
    const double *gu = P.gu + (size_t)ms * L * n * numu;
    const double *zb = P.zb + (size_t)ms * L * numu;
    const double *z0u = P.z0u + (size_t)(ms - mazim) * L * numu;
    const double *z1u = P.z1u + (size_t)(ms - mazim) * L * numu;
#define GC(i, j, lc) gc[((size_t)((lc) - 1) * n + ((i) - 1)) * n + ((j) - 1)]
#define KK(i, lc) kk[((lc) - 1) * n + ((i) - 1)]
#define EK(i, lc) ek[((lc) - 1) * nn + ((i) - 1)]
#define ZZ(i, lc) zz[((lc) - 1) * n + ((i) - 1)]
#define ZP0(i, lc) zp0[((lc) - 1) * n + ((i) - 1)]
#define ZP1(i, lc) zp1[((lc) - 1) * n + ((i) - 1)]
#define LL(i, lc) ll[((lc) - 1) * n + ((i) - 1)]
#define GU(iu, iq, lc) gu[((size_t)((lc) - 1) * n + ((iq) - 1)) * numu + ((iu) - 1)]
#define ZB(iu, lc) zb[((lc) - 1) * numu + ((iu) - 1)]
#define Z0U(iu, lc) z0u[((lc) - 1) * numu + ((iu) - 1)]
#define Z1U(iu, lc) z1u[((lc) - 1) * numu + ((iu) - 1)]

    // ---- surface term, identical for every user angle (Lambertian RMU = albedo,
    //      disort.f:4736-4781): bnddfu = sum_{iq=nn..1} (1+delm0) albedo CMU CWT DFUINT(iq) ----
    double bndsrf = 0.0;   // BNDDFU + BNDDIR + DELM0*EMU*BPLANK
    // (a bidirectional surface reflects in every azimuth mode, and differently into every user angle: RMU, EMU of
    //  SURFAC, sbd_surface.hpp -- the sum over the downward streams is then formed per angle, below)
    const bool brdf = P.ibdrf != 0;
    const size_t sidx = surf_index(P, slot, mazim);
    const double *rmut = brdf ? surf_rmu(P, sidx) : nullptr, *emut = brdf ? surf_emu(P, sidx) : nullptr;
    const bool has_surface = !lyrcut && (brdf || mazim == 0);
    double *dfu = smem;   // [nn]
    if (has_surface) {
        if (lane < nn) {
            const int iq = lane + 1;
            double dfuint = 0.0;
            for (int jq = 1; jq <= nn; ++jq) dfuint = dfuint + GC(iq, jq, L) * LL(jq, L);
            for (int jq = nn + 1; jq <= n; ++jq)
                dfuint = dfuint + GC(iq, jq, L) * LL(jq, L) * exp(-KK(jq, L) * dtaucp[L - 1]);
            if (beam) dfuint = dfuint + ZZ(iq, L) * expbea[L];
            dfuint = dfuint + delm0 * (ZP0(iq, L) + ZP1(iq, L) * taucpr[L]);
            dfu[lane] = dfuint;
        }
        wave_lds_sync();
        double bnddfu = 0.0;
        for (int iq = nn; iq >= 1; --iq)
            bnddfu = bnddfu + (1.0 + delm0) * albedo * cmu[nn - iq] * cwt[nn - iq] * dfu[iq - 1];
        double bnddir = 0.0;
        if (beam) bnddir = umu0 * fbeam / pi * albedo * expbea[L];
        bndsrf = bnddfu + bnddir + delm0 * (1.0 - albedo) * bplank;
    }

    const double lh = SBD_F32(0.0001), eps6 = SBD_F32(1.0e-6);
    // 1 / x to working precision: hardware seed + two Newton steps (the IEEE division sequence is a third of this
    // kernel's instructions; the quotients only enter sums that are gated at 5e-6 of the column maximum)
    auto rcp = [](double x) -> double {
        double r = __builtin_amdgcn_rcp(x);
        r = r * (2.0 - x * r);
        return r * (2.0 - x * r);
    };
    // The layer-and-stream sums of one (level, angle) item.  An item is ACTIVE when its path crosses at least one
    // whole layer or a part of its own (looking down from the top level, up from the bottom level ... are not: only
    // the boundary term is left).  With IOUT's usual level pair half of the items are inactive and a lane per item
    // leaves two thirds of the wave idle through the long loop: the active items are listed, and S = 1, 2, 4 or 8 lanes
    // share one item, each taking nn / S streams of either half (S divides nn, S x active items <= 64); slice 0 also
    // carries the beam and thermal terms; the partial sums meet on the DPP network.  (The sums over layers and
    // streams therefore do not run in the reference's order any more: intensities are gated by tolerance.)
    int *alist = (int *)(smem + nn + 2);                 // [nlev * numu] active items, compacted in item order
    const int nitem = nlev * numu;
    auto geometry = [&](const int item, int &lev, int &lyu, int &iu, bool &dark) {
        const int li = item / numu;
        iu = item % numu + 1;
        lev = P.all_levels ? li : P.t.level_out[li];
        lyu = layru[lev];
        dark = lyrcut && lyu > ncut;
    };
    auto is_active = [&](const int lev, const int lyu, const int iu, const bool dark) -> bool {
        if (dark) return false;
        const double um = umu[iu - 1], up = utaupr[lev];
        const bool negumu = um < 0.0;
        const bool whole = negumu ? (lyu - 1 >= 1) : (lyu + 1 <= ncut);
        const bool skip = (fabs(up - taucpr[lyu - 1]) < eps6 && negumu) || (fabs(up - taucpr[lyu]) < eps6 && !negumu);
        return whole || !skip;
    };
    int nactive = 0;
    for (int base = 0; base < nitem; base += 64) {
        const int item = base + lane;
        bool act = false;
        if (item < nitem) {
            int lev, lyu, iu;
            bool dark;
            geometry(item, lev, lyu, iu, dark);
            act = is_active(lev, lyu, iu, dark);
        }
        const unsigned long long m = __ballot(act);
        if (act) alist[nactive + __popcll(m & ((1ull << lane) - 1ull))] = item;
        nactive += __popcll(m);
    }
    wave_lds_sync();
    int S = 1;
    while (S < 8 && 2 * S * nactive <= 64 && nn % (2 * S) == 0) S *= 2;
    const int per_round = 64 / S, sl = lane % S, nq = nn / S;
    // ---- inactive items: the boundary term alone ----
    auto boundary = [&](const int iu, const double um, const double up, const bool negumu) -> double {
        if (negumu && mazim == 0) return (P.fisot + tplank) * exp(up / um);
        if (!negumu && has_surface && brdf) {
            double bnddfu = 0.0;
            for (int iq = nn; iq >= 1; --iq)
                bnddfu = bnddfu + (1.0 + delm0) * SBD_RMU(rmut, iu, nn + 1 - iq) * cmu[nn - iq] * cwt[nn - iq] * dfu[iq - 1];
            double bnddir = 0.0;
            if (beam) bnddir = umu0 * fbeam / pi * SBD_RMU(rmut, iu, 0) * expbea[L];
            return (bnddfu + bnddir + delm0 * emut[iu - 1] * bplank) * exp((up - taucpr[L]) / um);
        }
        if (!negumu && has_surface) return bndsrf * exp((up - taucpr[L]) / um);
        return 0.0;
    };
    for (int item = lane; item < nitem; item += 64) {
        int lev, lyu, iu;
        bool dark;
        geometry(item, lev, lyu, iu, dark);
        if (is_active(lev, lyu, iu, dark)) continue;              // (written once, by the lanes that sum its layers below)
        double result = 0.0;
        if (!dark) {
            const double um = umu[iu - 1];
            result = boundary(iu, um, utaupr[lev], um < 0.0);
        }
        uum[(size_t)(item / numu) * numu + (iu - 1)] = result;
    }
    // ---- active items, S lanes each ----
    for (int base = 0; base < nactive; base += per_round) {
        const int ai = base + lane / S;
        const bool live = ai < nactive;
        const int item = alist[live ? ai : 0];
        int lev, lyu, iu;
        bool dark;
        geometry(item, lev, lyu, iu, dark);
        const double um = umu[iu - 1];
        const double up = utaupr[lev];
        const bool negumu = um < 0.0;
        const double exp0 = beam ? exp(-up / umu0) : 0.0;
        const double rum = rcp(um);
        int lyrstr, lyrend;
        double sgn;
        if (negumu) { lyrstr = 1; lyrend = lyu - 1; sgn = -1.0; }
        else { lyrstr = lyu + 1; lyrend = ncut; sgn = 1.0; }
        if (!live) lyrend = lyrstr - 1;
        const int q0 = sl * nq;                           // this lane's streams: q0+1 .. q0+nq of either half
        double palint = 0.0, plkint = 0.0, exp1 = 0.0, exp2 = 0.0, denom, expn;
        // (the loads of a batch of streams -- KK, EK, LL, GU: 4 per stream -- are issued together, then the batch is
        //  summed in the same order as before: the loop used to pay a memory round trip per stream, 1 056 of them in a
        //  row at NSTR 32 x 33 layers.  exp2 of a layer is exp1 of the next: the same expression on the same operands)
        // (a layer's own scalars -- TAUCPR, DTAUCP, EXPBEA, ZB / Z0U / Z1U -- are asked for one layer ahead)
        const bool any = lyrstr <= lyrend;
        if (any) exp2 = exp((up - taucpr[lyrstr - 1]) * rum);
        const int lf = any ? lyrstr : 1;
        double tc_lo = taucpr[lf - 1], eb_lo = expbea[lf - 1];
        double tc_n = taucpr[lf], eb_n = expbea[lf], dt_n = dtaucp[lf - 1];
        double zb_n = (beam && sl == 0) ? ZB(iu, lf) : 0.0, z0_n = (therm && sl == 0) ? Z0U(iu, lf) : 0.0, z1_n = (therm && sl == 0) ? Z1U(iu, lf) : 0.0;
        for (int lc = lyrstr; lc <= lyrend; ++lc) {
            const double dtau = dt_n, tc_hi = tc_n, eb_hi = eb_n, zbc = zb_n, z0c = z0_n, z1c = z1_n;
            {
                const int ln = (lc < lyrend) ? lc + 1 : lc;      // (the last layer asks for itself again: a valid address)
                tc_n = taucpr[ln]; eb_n = expbea[ln]; dt_n = dtaucp[ln - 1];
                if (beam && sl == 0) zb_n = ZB(iu, ln);
                if (therm && sl == 0) { z0_n = Z0U(iu, ln); z1_n = Z1U(iu, ln); }
            }
            exp1 = exp2;
            exp2 = exp((up - tc_hi) * rum);
            if (sl == 0) {
                if (therm) {
                    const double f0n = sgn * (exp1 - exp2);
                    const double f1n = sgn * ((tc_lo + um) * exp1 - (tc_hi + um) * exp2);
                    plkint = plkint + z0c * f0n + z1c * f1n;
                }
                if (beam) {
                    denom = 1.0 + um / umu0;
                    if (fabs(denom) < lh) expn = (dtau / umu0) * exp0;
                    else expn = (exp1 * eb_lo - exp2 * eb_hi) * sgn / denom;
                    palint = palint + zbc * expn;
                }
            }
            tc_lo = tc_hi; eb_lo = eb_hi;
            const double *kp = kk + (lc - 1) * n + q0, *lp = ll + (lc - 1) * n + q0;
            const double *gp = gu + ((size_t)(lc - 1) * n + q0) * numu + (iu - 1);
            const double *ep = ek + (lc - 1) * nn + q0;                       // EK(iq, lc), iq = q0+1 ..
            const double *er = ek + (lc - 1) * nn + (nn - 1 - q0);            // EK(n+1-iq, lc), iq = nn+q0+1 ..: descending
            const double dlo = dtau * rum * exp2, dhi = -dtau * rum * exp1;
            auto lower = [&](const double k, const double e, const double l, const double g) {   // KK negative
                const double dn = 1.0 + um * k;
                const double ex = (fabs(dn) < lh) ? dlo : sgn * (exp1 * e - exp2) * rcp(dn);
                palint = palint + (g * l) * ex;
            };
            auto upper = [&](const double k, const double e, const double l, const double g) {   // KK positive
                const double dn = 1.0 + um * k;
                const double ex = (fabs(dn) < lh) ? dhi : sgn * (exp1 - exp2 * e) * rcp(dn);
                palint = palint + (g * l) * ex;
            };
            // batches of 8, then 4, 2, 1 streams over the lane's 2 nq streams, the nq of the lower half first (nq = 8 at
            // NSTR 32 with two lanes per item: a batch per half; 4 at NSTR 16: both halves in one batch)
            auto run = [&](auto bb, int &s0) {
                constexpr int B = decltype(bb)::value;
                for (; s0 + B <= 2 * nq; s0 += B) {
                    double k8[B], e8[B], l8[B], g8[B];
#pragma unroll
                    for (int t = 0; t < B; ++t) {
                        const int sq = s0 + t;
                        const bool up_half = sq >= nq;
                        const int c = up_half ? sq - nq : sq, ix = up_half ? nn + c : c;
                        k8[t] = kp[ix]; l8[t] = lp[ix]; g8[t] = gp[(size_t)ix * numu];
                        e8[t] = up_half ? er[-c] : ep[c];
                    }
#pragma unroll
                    for (int t = 0; t < B; ++t) {
                        if (s0 + t >= nq) upper(k8[t], e8[t], l8[t], g8[t]);
                        else lower(k8[t], e8[t], l8[t], g8[t]);
                    }
                }
            };
            int s0 = 0;
            run(std::integral_constant<int, 8>{}, s0);
            run(std::integral_constant<int, 4>{}, s0);
            run(std::integral_constant<int, 2>{}, s0);
            run(std::integral_constant<int, 1>{}, s0);
        }
        // from the output level to the adjacent computational level
        const double dtau1 = up - taucpr[lyu - 1];
        const double dtau2 = up - taucpr[lyu];
        const bool skip = !live || (fabs(dtau1) < eps6 && negumu) || (fabs(dtau2) < eps6 && !negumu);
        if (!skip) {
            if (negumu) exp1 = exp(dtau1 * rum);
            else exp2 = exp(dtau2 * rum);
            if (beam && sl == 0) {
                denom = 1.0 + um / umu0;
                if (fabs(denom) < lh) expn = (dtau1 / umu0) * exp0;
                else if (negumu) expn = (exp0 - expbea[lyu - 1] * exp1) / denom;
                else expn = (exp0 - expbea[lyu] * exp2) / denom;
                palint = palint + ZB(iu, lyu) * expn;
            }
            const double dtau = dtaucp[lyu - 1];
            for (int iq = q0 + 1; iq <= q0 + nq; ++iq) {
                const double kq = KK(iq, lyu);
                denom = 1.0 + um * kq;
                if (fabs(denom) < lh) expn = -dtau2 * rum * exp2;
                else if (negumu) expn = (exp(-kq * dtau2) - exp(kq * dtau) * exp1) * rcp(denom);
                else expn = (exp(-kq * dtau2) - exp2) * rcp(denom);
                palint = palint + (GU(iu, iq, lyu) * LL(iq, lyu)) * expn;
            }
            for (int iq = nn + q0 + 1; iq <= nn + q0 + nq; ++iq) {
                const double kq = KK(iq, lyu);
                denom = 1.0 + um * kq;
                if (fabs(denom) < lh) expn = -dtau1 * rum * exp1;
                else if (negumu) expn = (exp(-kq * dtau1) - exp1) * rcp(denom);
                else expn = (exp(-kq * dtau1) - exp(-kq * dtau) * exp2) * rcp(denom);
                palint = palint + (GU(iu, iq, lyu) * LL(iq, lyu)) * expn;
            }
            if (therm && sl == 0) {
                double fact;
                if (negumu) { expn = exp1; fact = taucpr[lyu - 1] + um; }
                else { expn = exp2; fact = taucpr[lyu] + um; }
                const double f0n = 1.0 - expn;
                const double f1n = up + um - fact * expn;
                plkint = plkint + Z0U(iu, lyu) * f0n + Z1U(iu, lyu) * f1n;
            }
        }
        double part = palint + plkint;
        for (int d = 1; d < S; d <<= 1) part = part + __shfl_xor(part, d, 64);
        if (live && sl == 0) uum[(size_t)(item / numu) * numu + (iu - 1)] = part + boundary(iu, um, up, negumu);
    }
#undef GC
#undef KK
#undef EK
#undef ZZ
#undef ZP0
#undef ZP1
#undef LL
#undef GU
#undef ZB
#undef Z0U
#undef Z1U
}

// CMPINT (disort.f:1658-1778), USRANG = false: the azimuthal intensity components at the QUADRATURE angles, in the
// stored -1 .. +1 order of the streams (SETDIS sets UMU to them, disort.f:2655-2669).  One wave per (item, mode), a
// lane per (output level, stream); needs GC of the output levels' layers and LL.
__global__ void __launch_bounds__(64) cmpint_kernel(Params P)
{
    const int lane = threadIdx.x;
    const int nmode = P.nmode;
    // (blocks in mode-major order: the items of mode 0 first.  Item-major, the modes an item does not need -- no beam, no
    //  moment left, SBD_SVI_NAZ -- left their live blocks on a few of the eight XCDs: block b goes to XCD b mod 8)
    const int mazim = (int)(blockIdx.x / (unsigned)P.nslot);
    const int slot = (int)(blockIdx.x % (unsigned)P.nslot);
    if (mazim >= nmode) return;
    const long long ms = (long long)slot * nmode + mazim;
    const int L = P.L, n = P.n, nn = P.nn, nlev = P.nlev;
    const int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const int st0 = svi[SBD_SVI_STATUS];
    const double fbeam = P.fbeam[slot];
    double *uum = P.uum + (size_t)ms * nlev * n;
    if ((st0 & (0x20 | 0x10 | 0x08)) != 0 || mazim > svi[SBD_SVI_NAZ]) {
        for (int i = lane; i < nlev * n; i += 64) uum[i] = 0.0;
        return;
    }
    const int ncut = svi[SBD_SVI_NCUT];
    const bool lyrcut = svi[SBD_SVI_LYRCUT] != 0;
    const bool therm = P.plank[slot] != 0 && mazim == 0;
    const int32_t *layru = svi + SBD_SVI_LAYRU;
    const SV o(L);
    const double *sv = P.sv + (size_t)slot * P.sv_stride;
    const double *taucpr = sv + o.taucpr(), *utaupr = sv + o.utaupr();
    const double *gc = P.gc + (size_t)ms * L * n * n;
    const double *kk = P.kk + (size_t)ms * L * n;
    const double *zz = P.zz + (size_t)ms * L * n;
    const double *zp0 = P.zp0 + (size_t)(ms - mazim) * L * n;
    const double *zp1 = P.zp1 + (size_t)(ms - mazim) * L * n;
    const double *ll = P.ll + (size_t)ms * L * n;
    for (int item = lane; item < nlev * n; item += 64) {
        const int li = item / n, iq = item % n + 1;
        const int lev = P.all_levels ? li : P.t.level_out[li];
        const int lyu = layru[lev];
        double r = 0.0;
        if (!(lyrcut && lyu > ncut)) {
            const double up = utaupr[lev];
            const size_t lo = (size_t)(lyu - 1) * n;
            double zint = 0.0;
            for (int jq = 1; jq <= nn; ++jq)
                zint = zint + gc[(lo + (iq - 1)) * n + (jq - 1)] * ll[lo + jq - 1] * exp(-kk[lo + jq - 1] * (up - taucpr[lyu]));
            for (int jq = nn + 1; jq <= n; ++jq)
                zint = zint + gc[(lo + (iq - 1)) * n + (jq - 1)] * ll[lo + jq - 1] * exp(-kk[lo + jq - 1] * (up - taucpr[lyu - 1]));
            r = zint;
            if (fbeam > 0.0) r = zint + zz[lo + iq - 1] * exp(-up / P.umu0);
            if (therm) r = r + zp0[lo + iq - 1] + zp1[lo + iq - 1] * up;
        }
        uum[(size_t)li * n + (iq - 1)] = r;
    }
}

// grid: ceil(nslot*nlev*numu / 256) blocks of 256 threads.
__global__ void __launch_bounds__(256) azimuth_kernel(Params P, int naz_run)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nlev = P.nlev, numu = P.numu, nphi = P.nphi, nmode = P.nmode;
    const long long per = (long long)nlev * numu;
    if (tid >= per * P.nslot) return;
    const int slot = (int)(tid / per);
    const int rem = (int)(tid % per);   // li*numu + iu
    const int nazi = P.svi[(size_t)slot * P.svi_stride + SBD_SVI_NAZ];
    const int naz = (nazi < naz_run) ? nazi : naz_run;   // disort.f:577-586 (0 without a beam) and the item's last mode with a moment
    const double *uum = P.uum + (size_t)slot * nmode * per + rem;
    double *uu = P.uu + (size_t)slot * nphi * per + rem;
    // All NAZ modes are added.  DISORT stops the series after two consecutive modes whose largest term is
    // <= ACCUR times the running sum (disort.f:821-825); SBDART always passes ACCUR = 0 (drt.f:142), the
    // C ABI has no ACCUR field, and with ACCUR = 0 the test can only pass when every term of two modes
    // is exactly zero -- adding further exact zeros is then a no-op, so the sums are the reference's.
    for (int j = 0; j < nphi; ++j) {
        double acc = uum[0];
        for (int m = 1; m <= naz; ++m) acc = acc + uum[(size_t)m * per] * P.t.cosmphi[(size_t)m * nphi + j];
        uu[(size_t)j * per] = acc;
    }
}

}  // namespace sbd
