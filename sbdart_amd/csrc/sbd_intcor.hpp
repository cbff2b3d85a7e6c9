// Nakajima/Tanaka intensity corrections (CORINT = true): INTCOR disort.f:2044-2297 with SINSCA
// disort.f:2996-3097, SECSCA disort.f:2299-2452, XIFUNC disort.f:4795-4862; applied to the summed
// intensities UU after the azimuth series (disort.f:831-842) at the requested output levels.
//
//   UU(iu, lu, jp) += [single scattering with the exact phase function and the unscaled optical depths]
//                   - [single scattering as the delta-M-truncated discrete-ordinate solution contains it]
//                   - [second-order correction around the forward peak, viewing angles within 10 degrees
//                      of the beam]
// One block per work item.  Phase 1: a thread per (angle pair, layer) sums the Legendre series of the full
// (NMOM moments, 299 in SBDART's CORINT runs) and the truncated phase function -- the recurrence runs per
// thread, the moments are read along the thread's own row; the results go to LDS.  Phase 2: a thread per
// (angle pair, output level) evaluates the three terms.  Angle pairs are taken PAIRS at a time.
#pragma once
#include "sbd_common.hpp"

namespace sbd {

constexpr int kIntcorPairs = 8;      // angle pairs per pass of a block

SBD_DEVICE double xifunc(double umu1, double umu2, double umu3, double tau)
{
    const double x1 = 1.0 / umu1 - 1.0 / umu2;
    const double x2 = 1.0 / umu1 - 1.0 / umu3;
    const double exp1 = exp(-tau / umu1);
    if (umu2 == umu3 && umu1 == umu2) return tau * tau * exp1 / (2.0 * umu1 * umu2);
    if (umu2 == umu3 && umu1 != umu2) return ((tau - 1.0 / x1) * exp(-tau / umu2) + exp1 / x1) / (x1 * umu1 * umu2);
    if (umu2 != umu3 && umu1 == umu2) return ((exp(-tau / umu3) - exp1) / x2 - tau * exp1) / (x2 * umu1 * umu2);
    if (umu2 != umu3 && umu1 == umu3) return ((exp(-tau / umu2) - exp1) / x1 - tau * exp1) / (x1 * umu1 * umu2);
    return ((exp(-tau / umu3) - exp1) / x2 - (exp(-tau / umu2) - exp1) / x1) / (x2 * umu1 * umu2);
}

// singly scattered intensity at optical depth utau in direction umu; phase/omega by layer (0-based), tau by level
SBD_DEVICE double sinsca(double dither, int layru, int nlyr, const double *phase, const double *omega,
                         const double *tau, double umu, double umu0, double utau, double fbeam, double pi)
{
    double s = 0.0;
    double exp0 = exp(-utau / umu0);
    if (fabs(umu + umu0) <= dither) {
        for (int lyr = 1; lyr <= layru - 1; ++lyr) s = s + omega[lyr - 1] * phase[lyr - 1] * (tau[lyr] - tau[lyr - 1]);
        return fbeam / (4.0 * pi * umu0) * exp0 * (s + omega[layru - 1] * phase[layru - 1] * (utau - tau[layru - 1]));
    }
    if (umu > 0.0) {
        for (int lyr = layru; lyr <= nlyr; ++lyr) {
            const double exp1 = exp(-((tau[lyr] - utau) / umu + tau[lyr] / umu0));
            s = s + omega[lyr - 1] * phase[lyr - 1] * (exp0 - exp1);
            exp0 = exp1;
        }
    } else {
        for (int lyr = layru; lyr >= 1; --lyr) {
            const double exp1 = exp(-((tau[lyr - 1] - utau) / umu + tau[lyr - 1] / umu0));
            s = s + omega[lyr - 1] * phase[lyr - 1] * (exp0 - exp1);
            exp0 = exp1;
        }
    }
    return fbeam / (4.0 * pi * (1.0 + umu / umu0)) * s;
}

// grid: one block of 256 threads per work item; dynamic LDS: 2 * kIntcorPairs * L doubles
__global__ void __launch_bounds__(256) intcor_kernel(Params P, int naz_run)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int slot = blockIdx.x;
    if (slot >= P.nslot) return;
    const int L = P.L, nstr = P.n, nmom = P.nmom, numu = P.numu, nphi = P.nphi, nlev = P.nlev;
    const int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    if (svi[SBD_SVI_STATUS] & (0x20 | 0x10 | 0x08)) return;
    const double fbeam = P.fbeam[slot];
    if (fbeam == 0.0) return;                                  // CORINT is off without a beam (disort.f:2695)
    const SV o(L);
    const double *sv = P.sv + (size_t)slot * P.sv_stride;
    const double *ssalb = sv + o.ssalb(), *flyr = sv + o.flyr(), *oprim = sv + o.oprim();
    const double *taucpr = sv + o.taucpr(), *utaupr = sv + o.utaupr(), *tauc = sv + o.utau();
    {
        double yessct = 0.0;                                   // ... and without scattering
        for (int lc = 0; lc < L; ++lc) yessct = yessct + ssalb[lc];
        if (yessct == 0.0) return;
    }
    const int32_t *layru = svi + SBD_SVI_LAYRU;
    const int ncut = svi[SBD_SVI_NCUT];
    const bool lyrcut = svi[SBD_SVI_LYRCUT] != 0;
    const double *pmom = P.pmom + pmom_item(P, slot) * L * (nmom + 1);
    const double *dtauc = P.dtauc + (size_t)slot * L;
    const double umu0 = P.umu0, pi = P.pi, dither = P.dither, rpd = pi / 180.0;
    double *uu = P.uu + (size_t)slot * nphi * nlev * numu;
    double *phast = smem, *phasm = smem + kIntcorPairs * L;
    const int npair = numu * nphi;
    for (int p0 = 0; p0 < npair; p0 += kIntcorPairs) {
        const int np = (npair - p0 < kIntcorPairs) ? npair - p0 : kIntcorPairs;
        // ---- phase functions of every layer at the scattering angle of each pair ----
        for (int t = threadIdx.x; t < np * ncut; t += blockDim.x) {
            const int pc = t / ncut, lc = t % ncut;
            const int pair = p0 + pc, iu = pair / nphi, jp = pair % nphi;
            const double umu = P.t.umu[iu];
            const double cphi = P.t.cosphi[jp];                                  // cos(phi - phi0), disort.f:2188-2190
            const double ctheta = -umu0 * umu + sqrt((1.0 - umu0 * umu0) * (1.0 - umu * umu)) * cphi;
            const double *pm = pmom + (size_t)lc * (nmom + 1);
            const double f = flyr[lc];
            double pa = 1.0, pmt = 1.0, plm1 = 1.0, plm2 = 0.0;
            for (int k = 1; k <= nmom; ++k) {
                const double pl = ((double)(2 * k - 1) * ctheta * plm1 - (double)(k - 1) * plm2) / (double)k;
                plm2 = plm1;
                plm1 = pl;
                pa = pa + (double)(2 * k + 1) * pl * pm[k];
                if (k <= nstr - 1) pmt = pmt + (double)(2 * k + 1) * pl * (pm[k] - f) / (1.0 - f);
            }
            phast[pc * L + lc] = pa / (1.0 - f * ssalb[lc]);
            phasm[pc * L + lc] = pmt;
        }
        __syncthreads();
        // ---- the corrections at every output level ----
        for (int t = threadIdx.x; t < np * nlev; t += blockDim.x) {
            const int pc = t / nlev, ol = t % nlev;
            const int pair = p0 + pc, iu = pair / nphi, jp = pair % nphi;
            const int lev = P.all_levels ? ol : P.t.level_out[ol];
            const int lyu = layru[lev];
            if (lyrcut && !(lyu < ncut)) continue;
            const double umu = P.t.umu[iu];
            const double ussndm = sinsca(dither, lyu, ncut, phast + pc * L, ssalb, taucpr, umu, umu0, utaupr[lev], fbeam, pi);
            const double ussp = sinsca(dither, lyu, ncut, phasm + pc * L, oprim, taucpr, umu, umu0, utaupr[lev], fbeam, pi);
            double corr = ussndm - ussp;
            // second-order term: only looking up (umu < 0) within 10 degrees of the beam, not at the top
            if (umu < 0.0 && fabs(acos(-umu0) / rpd - acos(umu) / rpd) <= 10.0 && !(lev == 0 && tauc[0] <= dither)) {
                const double cphi = P.t.cosphi[jp];
                const double ctheta = -umu0 * umu + sqrt((1.0 - umu0 * umu0) * (1.0 - umu * umu)) * cphi;
                const double utau = tauc[lev];
                const double zero = (double)1e-4f;
                double dtau = utau - tauc[lyu - 1];
                double wbar = ssalb[lyu - 1] * dtau, fbar = flyr[lyu - 1] * wbar, stau = dtau;
                for (int lyr = 1; lyr <= lyu - 1; ++lyr) {
                    const double dt = fmax(dtauc[lyr - 1], 0.0);
                    wbar = wbar + ssalb[lyr - 1] * dt;
                    fbar = fbar + ssalb[lyr - 1] * dt * flyr[lyr - 1];
                    stau = stau + dt;
                }
                if (!(wbar <= zero || fbar <= zero || stau <= zero || fbeam <= zero)) {
                    fbar = fbar / wbar;
                    wbar = wbar / stau;
                    double pspike = 1.0, gbar = 1.0, plm1 = 1.0, plm2 = 0.0;
                    for (int k = 1; k <= nstr - 1; ++k) {
                        const double pl = ((double)(2 * k - 1) * ctheta * plm1 - (double)(k - 1) * plm2) / (double)k;
                        plm2 = plm1;
                        plm1 = pl;
                        pspike = pspike + (2.0 * gbar - gbar * gbar) * (double)(2 * k + 1) * pl;
                    }
                    for (int k = nstr; k <= nmom; ++k) {
                        const double pl = ((double)(2 * k - 1) * ctheta * plm1 - (double)(k - 1) * plm2) / (double)k;
                        plm2 = plm1;
                        plm1 = pl;
                        gbar = pmom[(size_t)(lyu - 1) * (nmom + 1) + k] * ssalb[lyu - 1] * dtau;
                        for (int lyr = 1; lyr <= lyu - 1; ++lyr)
                            gbar = gbar + pmom[(size_t)(lyr - 1) * (nmom + 1) + k] * ssalb[lyr - 1] * fmax(dtauc[lyr - 1], 0.0);
                        if (fbar * wbar * stau <= zero) gbar = 0.0;
                        else gbar = gbar / (fbar * wbar * stau);
                        pspike = pspike + (2.0 * gbar - gbar * gbar) * (double)(2 * k + 1) * pl;
                    }
                    const double umu0p = umu0 / (1.0 - fbar * wbar);
                    const double duims = fbeam / (4.0 * pi) * ((fbar * wbar) * (fbar * wbar)) / (1.0 - fbar * wbar) * pspike
                                         * xifunc(-umu, umu0p, umu0p, utau);
                    corr = corr - duims;
                }
            }
            double *u = uu + ((size_t)jp * nlev + ol) * numu + iu;
            *u = *u + corr;
        }
        __syncthreads();
    }
}

}  // namespace sbd
