// Setup kernel: per work item, everything DISORT does before its azimuth loop
// (disort.f:482-571): SSALB dither, cumulative optical depth, CHEKIN's per-item
// checks, SETDIS delta-M scaling / beam transmission / LYRCUT, output-level
// bookkeeping, and the band-integrated Planck function at every level.
//
// One lane per layer/level of one work item (64-lane wave = one work item; L<=65
// handled with a 2-pass stride), prefix sums by a serial lane through LDS -- the
// layer count is tiny, what matters is that the HBM reads of dtauc/ssalb/pmom are
// coalesced along the layer axis.
#pragma once
#include "sbd_common.hpp"

namespace sbd {

// PLKAVG (disort.f:5410-5671), stateless.  warn bit0: Simpson no-conv (errmsg 9),
// bit1: zero result (errmsg 10).
SBD_DEVICE double plkf(double x) { return x * x * x / (exp(x) - 1.0); }

SBD_DEVICE double plkavg(double wnumlo, double wnumhi, double t, double pi, int &warn)
{
    const double a1 = (double)(1.0f / 3.0f), a2 = -0.125, a3 = (double)(1.0f / 60.0f),
                 a4 = (double)(-1.0f / 5040.0f), a5 = (double)(1.0f / 272160.0f),
                 a6 = (double)(-1.0f / 13305600.0f);
    const double c2 = SBD_F32(1.438786), sigma = SBD_F32(5.67032e-8), vcut = 1.5;
    const double vcp[7] = {10.25, SBD_F32(5.7), SBD_F32(3.9), SBD_F32(2.9), SBD_F32(2.3), SBD_F32(1.9), 0.0};
    const double vmax = 709.782712893384 /* log(DBL_MAX) */, epsil = 2.220446049250313e-16;
    const double sigdpi = sigma / pi;
    const double conc = 15.0 / (pi * pi * pi * pi);
    if (t < SBD_F32(1.0e-4)) return 0.0;
    double v[2] = {c2 * wnumlo / t, c2 * wnumhi / t};
    const double t4 = (t * t) * (t * t);
    if (v[0] > epsil && v[1] < vmax && (wnumhi - wnumlo) / wnumhi < SBD_F32(1.0e-2)) {
        const double hh = v[1] - v[0];
        double oldval = 0.0, val = 0.0;
        const double val0 = plkf(v[0]) + plkf(v[1]);
        bool conv = false;
        for (int n = 1; n <= 10; ++n) {
            const double del = hh / (double)(2 * n);
            val = val0;
            for (int k = 1; k <= 2 * n - 1; ++k)
                val = val + (double)(2 * (1 + (k & 1))) * plkf(v[0] + (double)k * del);
            val = del / 3.0 * val;
            if (fabs((val - oldval) / val) <= SBD_F32(1.0e-6)) { conv = true; break; }
            oldval = val;
        }
        if (!conv) warn |= 1;
        return sigdpi * t4 * conc * val;
    }
    int smallv = 0;
    double p[2] = {0.0, 0.0}, d[2] = {0.0, 0.0};
    for (int i = 0; i < 2; ++i) {
        if (v[i] < vcut) {
            smallv++;
            const double vsq = v[i] * v[i];
            p[i] = conc * vsq * v[i] * (a1 + v[i] * (a2 + v[i] * (a3 + vsq * (a4 + vsq * (a5 + vsq * a6)))));
        } else {
            int mmax = 0;
            do { mmax++; } while (v[i] < vcp[mmax - 1]);
            const double ex = exp(-v[i]);
            double exm = 1.0, di = 0.0;
            for (int m = 1; m <= mmax; ++m) {
                const double mv = (double)m * v[i];
                exm = ex * exm;
                di = di + exm * (6.0 + mv * (6.0 + mv * (3.0 + mv))) / (double)(m * m * m * m);
            }
            d[i] = conc * di;
        }
    }
    double r;
    if (smallv == 2) r = p[1] - p[0];
    else if (smallv == 1) r = 1.0 - p[0] - d[1];
    else r = d[0] - d[1];
    r = sigdpi * t4 * r;
    if (r == 0.0) warn |= 2;
    return r;
}

// grid: one 64-thread block per work item.
__global__ void __launch_bounds__(64) setup_kernel(Params P)
{
    const int slot = blockIdx.x;
    if (slot >= P.nslot) return;
    const int lane = threadIdx.x;
    if (slot == 0 && lane == 0) { P.eiglist[0] = 0; P.rclist[0] = 0; }      // empty lists for this pass's layer kernels / band_rcond_kernel
    const int L = P.L, n = P.n, nmom = P.nmom;
    const SV o(L);
    double *sv = P.sv + (size_t)slot * P.sv_stride;
    int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const double *dtauc = P.dtauc + (size_t)slot * L;
    const double *ssalb_in = P.ssalb + (size_t)slot * L;
    const double *pmom = P.pmom + pmom_item(P, slot) * L * (nmom + 1);
    const double fbeam = P.fbeam[slot], umu0 = P.umu0;
    const bool plank = P.plank[slot] != 0;
    const double wlo = P.wvnmlo[slot], whi = P.wvnmhi[slot];

    __shared__ double s_dt[kMaxNlyr + 1], s_w[kMaxNlyr + 1], s_f[kMaxNlyr + 1];
    __shared__ double s_tauc[kMaxNlyr + 2], s_taucpr[kMaxNlyr + 2];
    __shared__ double s_pk[kMaxNlyr + 2];
    __shared__ int s_err, s_ncut, s_lyrcut, s_pw, s_kmax;
    if (lane == 0) { s_err = 0; s_pw = 0; s_kmax = 0; }
    __syncthreads();

    // ---- per-layer loads (coalesced along the layer axis) + CHEKIN per-item checks ----
    int err = 0;
    bool nearcons = false;               // a layer within 1e-12 of conservative scattering (SSALB = 1, which DISORT dithers, included)
    for (int lc = lane; lc < L; lc += 64) {
        double w = ssalb_in[lc];
        double dt = dtauc[lc];
        if (w < 0.0 || w > 1.0) err = 1;                     // disort.f:4950-4954
        nearcons = nearcons || (w >= 1.0 - 1.0e-12);
        if (w == 1.0) w = 1.0 - P.dither;                    // disort.f:486
        s_w[lc] = w;
        s_dt[lc] = dt;                                       // unclamped: TAUC uses it (disort.f:487)
        s_f[lc] = (n <= nmom) ? pmom[(size_t)lc * (nmom + 1) + n] : 0.0;   // F = PMOM(NSTR) (disort.f:2577)
        if (plank && (P.t.temper[lc + 1] < 0.0 || (lc == 0 && P.t.temper[0] < 0.0))) err = 1;
    }
    // PMOM range check (disort.f:4972-4981), all lanes stride the whole [L][nmom+1] block
    // (sixteen loads in flight per lane and round: issued one at a time, each with its own wait, this loop WAS the
    //  kernel -- ten dependent HBM round trips per block for a range check)
    for (int i0 = lane; i0 < L * (nmom + 1); i0 += 64 * 16) {
        double pv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + 64 * u;
            pv[u] = (i < L * (nmom + 1)) ? pmom[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (pv[u] < -1.0 || pv[u] > 1.0) err = 1;
    }
    if (fbeam < 0.0 || (fbeam > 0.0 && (umu0 <= 0.0 || umu0 > 1.0))) err = 1;
    if (P.ibdrf == 0) {                  // LAMBER: ALBEDO in [0,1] (disort.f:5075-5078); a bidirectional surface is
        const double alb = P.albedo[slot];   // tested through its flux albedo (sbd_surface.hpp; brdf_bad: once per run)
        if (alb < 0.0 || alb > 1.0) err = 1;
    } else if (P.brdf_bad) err = 1;
    if (plank && (wlo < 0.0 || whi <= wlo)) err = 1;
    if (err) atomicOr(&s_err, 1);
    __syncthreads();
    // ---- the item's last azimuth mode that can differ from zero (radiance runs).  Mode m works with the delta-M scaled
    //      moments GL(l), l = m .. NSTR-1 (disort.f:2583-2585: (2l+1) OPRIM (PMOM(l) - F) / (1 - F)); when every one of them
    //      is zero in every layer -- molecular scattering alone has none beyond l = 2 -- the mode's source, its boundary
    //      terms and therefore its intensities are exactly zero (Lambertian surface: it reflects in mode 0 only): the reference finds two such modes in a row and leaves
    //      its loop (disort.f:821-825 with ACCUR = 0), this engine used to solve all NSTR of them.  No beam: NAZ = 0
    //      (disort.f:577-586). ----
    if (P.nmode > 1) {
        int kmax = 0;
        if (fbeam > 0.0 && P.ibdrf != 0) kmax = P.nmode - 1;   // (a bidirectional surface reflects the beam into every mode)
        else if (fbeam > 0.0) {
            const int nm1 = nmom + 1;
            int lc = lane / nm1, k = lane - lc * nm1;
            for (int i = lane; i < L * nm1; i += 64) {
                if (k < n && k > kmax) {
                    const double w = s_w[lc], f = s_f[lc];
                    if (w != 0.0 && f != 1.0 && pmom[i] != f) kmax = k;
                }
                k += 64;
                while (k >= nm1) { k -= nm1; ++lc; }
            }
        }
        atomicMax(&s_kmax, kmax);
        __syncthreads();
    }

    // ---- serial prefix pass: TAUC, delta-M optical depth, NCUT.  The sums run in the reference's order.  (Round 4 tried
    //      a wave scan on the DPP network instead -- six adds per quantity, 0.60 -> 0.50 ms per step -- and the parity
    //      fuzz refused it: with a layer of optical depth ~1e-9 the reference's own thermal source term XR0 + XR1 * tau
    //      cancels ten digits, and TAUCPR rounded as a tree instead of a chain moved DFDT by 9e-5 of its maximum.  Only
    //      the reference's summation order reproduces the reference there.)  Not on one lane fetching its operands
    //      from LDS, though: but not on one lane fetching its operands from LDS -- three dependent LDS round trips
    //      per layer were most of this kernel's life: every lane walks the chain on wave-uniform operands (v_readlane
    //      from the lanes that loaded them) and keeps the partial sums of its own level ----
    if (L <= 64) {
        const int lc_mine = (lane < L) ? lane : 0;
        const double w_mine = s_w[lc_mine], f_mine = s_f[lc_mine], dtraw_mine = s_dt[lc_mine];
        double tauc = 0.0, taucpr = 0.0, abstau = 0.0, yessct = 0.0;
        double tauc_me = 0.0, taucpr_me = 0.0;            // TAUC(lane), TAUCPR(lane): lane <-> level
        int ncut = L;
        for (int lc = 0; lc < L; ++lc) {
            const double w = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(w_mine), lc), __builtin_amdgcn_readlane(__double2loint(w_mine), lc));
            const double f = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(f_mine), lc), __builtin_amdgcn_readlane(__double2loint(f_mine), lc));
            const double dr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dtraw_mine), lc), __builtin_amdgcn_readlane(__double2loint(dtraw_mine), lc));
            tauc = tauc + dr;
            const double dt = (dr < 0.0) ? 0.0 : dr;               // CHEKIN clamp (disort.f:4944)
            yessct += w;
            if (abstau < 10.0) ncut = lc + 1;                      // ABSCUT (disort.f:2561)
            abstau = abstau + (1.0 - w) * dt;
            taucpr = taucpr + (1.0 - f * w) * dt;                  // disort.f:2579
            if (lane == lc + 1) { tauc_me = tauc; taucpr_me = taucpr; }
        }
        if (lane <= L) { s_tauc[lane] = tauc_me; s_taucpr[lane] = taucpr_me; }
        if (lane < L) s_dt[lane] = (dtraw_mine < 0.0) ? 0.0 : dtraw_mine;
        if (L == 64 && lane == 0) { s_tauc[64] = tauc; s_taucpr[64] = taucpr; }   // (level 64 has no lane)
        if (lane == 0) {
            const int lyrcut = (abstau >= 10.0 && !plank && !P.ibcnd && L > 1) ? 1 : 0;   // disort.f:2602-2603
            s_ncut = lyrcut ? ncut : L;
            s_lyrcut = lyrcut;
            if (yessct > 0.0 && nmom < n) s_err = 1;                     // disort.f:4966
        }
    } else if (lane == 0) {
        double tauc = 0.0, taucpr = 0.0, abstau = 0.0, yessct = 0.0;
        int ncut = L;
        s_tauc[0] = 0.0;
        s_taucpr[0] = 0.0;
        for (int lc = 0; lc < L; ++lc) {
            const double w = s_w[lc];
            tauc = tauc + s_dt[lc];
            s_tauc[lc + 1] = tauc;
            const double dt = (s_dt[lc] < 0.0) ? 0.0 : s_dt[lc];   // CHEKIN clamp (disort.f:4944)
            yessct += w;
            if (abstau < 10.0) ncut = lc + 1;                      // ABSCUT (disort.f:2561)
            abstau = abstau + (1.0 - w) * dt;
            const double f = s_f[lc];
            const double dtp = (1.0 - f * w) * dt;                 // disort.f:2579
            taucpr = taucpr + dtp;
            s_taucpr[lc + 1] = taucpr;
            s_dt[lc] = dt;
        }
        const int lyrcut = (abstau >= 10.0 && !plank && !P.ibcnd && L > 1) ? 1 : 0;   // disort.f:2602-2603
        if (!lyrcut) ncut = L;
        s_ncut = ncut;
        s_lyrcut = lyrcut;
        if (yessct > 0.0 && nmom < n) s_err = 1;                         // disort.f:4966
    }
    __syncthreads();

    // ---- per-layer outputs ----
    for (int lc = lane; lc < L; lc += 64) {
        const double w = s_w[lc], f = s_f[lc], dt = s_dt[lc];
        sv[o.ssalb() + lc] = w;
        sv[o.flyr() + lc] = f;
        sv[o.oprim() + lc] = w * (1.0 - f) / (1.0 - f * w);      // disort.f:2578
        sv[o.dtaucp() + lc] = (1.0 - f * w) * dt;
    }
    // ---- per-level outputs: TAUCPR, EXPBEA, output-level mapping, Planck ----
    int pw = 0;
    for (int lev = lane; lev <= L; lev += 64) {
        const double tp = s_taucpr[lev];
        sv[o.taucpr() + lev] = tp;
        double eb = (lev == 0) ? 1.0 : 0.0;
        if (lev > 0 && fbeam > 0.0) eb = exp(-tp / umu0);        // disort.f:2592
        sv[o.expbea() + lev] = eb;
        // USRTAU = .FALSE.: UTAU(lev+1) = TAUC(lev) (disort.f:2524-2529); LAYRU/UTAUPR (2610-2627)
        const double ut = s_tauc[lev];
        // (the first layer, counted from the top, that holds the level -- LAYRU's search -- found from the level's own
        //  index downwards: TAUC does not decrease, so only layers of zero depth above the level can come first)
        int lc = (lev < 1) ? 1 : lev;
        while (lc > 1 && s_tauc[lc - 1] >= ut) --lc;
        if (!(ut >= s_tauc[lc - 1] && ut <= s_tauc[lc])) {       // (negative optical depths: the literal search)
            for (lc = 1; lc <= L; ++lc)
                if (ut >= s_tauc[lc - 1] && ut <= s_tauc[lc]) break;
            if (lc > L) lc = L;
        }
        sv[o.utau() + lev] = ut;
        sv[o.utaupr() + lev] = s_taucpr[lc - 1] + (1.0 - s_w[lc - 1] * s_f[lc - 1]) * (ut - s_tauc[lc - 1]);
        svi[SBD_SVI_LAYRU + lev] = lc;
    }
    // band-integrated Planck function at every level (disort.f:564-569) and at the two boundary temperatures
    // (disort.f:556-557): L + 3 evaluations of the same routine, one per lane in ONE pass (the boundary two used
    // to follow on lane 0 alone, each as long as the whole pass)
    for (int lev = lane; lev <= L + 2; lev += 64) {
        const double tk = (lev <= L) ? P.t.temper[lev] : ((lev == L + 1) ? P.ttemp : P.btemp);
        const double pk = plank ? plkavg(wlo, whi, tk, P.pi, pw) : 0.0;
        if (lev <= L) {
            s_pk[lev] = pk;
            sv[o.pkag() + lev] = pk;
        } else if (lev == L + 1) {
            // (IBCND = 1: unit isotropic illumination from the top for the even slots -- it enters the boundary rows
            //  exactly where FISOT + TPLANK does, disort.f:3434-3599 vs SOLVE1, disort.f:7232-7317 -- and from the
            //  bottom for the odd ones, where BEM * BPLANK stands over a black surface)
            sv[o.tplank()] = P.ibcnd ? (((P.slot_base + slot) & 1) ? 0.0 : 1.0) : P.temis * pk;
        } else {
            sv[o.bplank()] = P.ibcnd ? (((P.slot_base + slot) & 1) ? 1.0 : 0.0) : pk;
        }
    }
    if (pw) atomicOr(&s_pw, pw);
    __syncthreads();
    // ---- thermal source slopes (disort.f:659-666) ----
    for (int lc = lane; lc < L; lc += 64) {
        double xr1 = 0.0, xr0 = 0.0;
        if (plank) {
            const double dtp = (1.0 - s_f[lc] * s_w[lc]) * s_dt[lc];
            const double p0 = s_pk[lc], p1 = s_pk[lc + 1];
            if (dtp > 0.0) xr1 = (p1 - p0) / dtp;
            xr0 = p0 - xr1 * s_taucpr[lc];
        }
        sv[o.xr0() + lc] = xr0;
        sv[o.xr1() + lc] = xr1;
    }
    {   // errmsg 2 (sbd_refband.hpp): the reference's band system can be singular to working precision only next to a layer
        // whose smallest eigenvalue is tiny against its neighbours' scale -- a layer (all but) conservative.  Hill climbs on
        // the oracle's RCOND with every layer either exactly conservative or at least 1e-6 away never got below 1e-15 from
        // above (tests/golden/make_illcond_warnings.py: search_notes); within 1e-12 of 1 they reach 1e-20.  The item's
        // systems are marked for the band kernels to list (2), everything else starts unlisted (0).
        const int mark = (__ballot(nearcons) != 0ull) ? 2 : 0;
        for (int m = lane; m < P.nmode; m += 64) P.rcflag[(size_t)slot * P.nmode + m] = mark;
    }
    if (lane == 0) {
        int st = 0;
        if (s_err) st |= 0x20;   // SBD_ST_ERR_INPUT
        if (s_pw & 2) st |= 0x40;    // SBD_ST_WARN_PLKAVG  (errmsg 10: returns zero)
        if (s_pw & 1) st |= 0x80;    // SBD_ST_WARN_PLKCONV (errmsg 9: Simpson's rule did not converge)
        // beam angle == quadrature angle (disort.f:2643-2650)
        if (fbeam > 0.0) {
            for (int iq = 0; iq < P.nn; ++iq)
                if (fabs(umu0 - P.t.cmu[iq]) / umu0 < SBD_F32(1.0e-4)) st |= 0x10;
        }
        svi[SBD_SVI_NCUT] = s_ncut;
        svi[SBD_SVI_NAZ] = (fbeam > 0.0) ? ((s_kmax < P.nmode - 1) ? s_kmax : P.nmode - 1) : 0;
        svi[SBD_SVI_LYRCUT] = s_lyrcut;
        svi[SBD_SVI_STATUS] = st;
    }
}

}  // namespace sbd
