// The scatterers' part of SBDART's band model for one wavelength, as host/device code: Rayleigh depths (rayleigh,
// spectra.f:179-247), the cloud deck's depths, single-scattering albedos and asymmetry factors from the Mie tables (taucloud
// with its look-up, taucloud.f:10-140, 6726-6768) and the boundary-layer and stratospheric aerosols (tauaero, tauaero.f:
// 1175-1359) -- the layer block of the compact batch form (include/sbdart_amd.h, sbd_mix_in::lay): channels DTAUC, DTAUA,
// DTAUR, their scattering depth, then asymmetry factor and the two factors of every scattering term.  SURVEY 8f row N1's
// other half on the device (the gas part: sbd_gas.hpp): north_star's "per-wavelength optical depths from taugas.f / tauaero.f /
// taucloud.f precomputed into coalesced HBM arrays" -- the blocks never exist on the host.
//
// Written from the Fortran host's restatement (sbdart_amd/fortran/sbd_bandmodel_mod.f90: rayleigh_depths;
// sbd_cloud_mod.f90: mie_lookup, cloud_depths; sbd_aerosol_mod.f90: boundary_layer_at, stratospheric_at, aerosol_depths --
// bit-equal to the live reference, tests/test_band_model.py) with the SAME sequence of roundings: every sum, product and
// quotient in the same order, no contraction, integer powers as products, the reference's bare literals as the REAL*4
// numbers they are (F(x) below), REAL*4 intrinsics where the Fortran has them (float(j)/n, log(2.)).  On the HOST the blocks
// are bit-equal to the Fortran host's (tests/test_scat_device.py through sbd_scatter_blocks_host); on the DEVICE log / pow
// are the device library's (a few ulps per call; the test states the bound).
// Covered: what the compact form covers -- a cloud deck with one cloud per layer, IAER 1..5 with one asymmetry factor per
// wavelength, stratospheric layers; not usrcld.dat, aerosol.dat, user moments, SPOWDER (those runs keep the host path).
#pragma once
#include <cmath>
#include <cstdint>
#pragma clang fp contract(off)      // one rounding per operation, like the reference's x86-64 object code

#if defined(__HIPCC__)
#define SBD_SC __host__ __device__ inline
#else
#define SBD_SC inline
#endif

namespace sbd {
namespace scat {

#define SCF(x) ((double)(x##f))      // a REAL*4 literal of the reference, widened

constexpr int NCLD = 5;             // cloud slots (params.f:12)
constexpr int NAERZ = 5;            // stratospheric aerosol layers (params.f)
constexpr int NAERW = 47;           // wavelengths of the stratospheric aerosol tables
constexpr int MXWV = 400, MRE = 13; // the Mie tables' grid: ln wavelength x log2 radius

struct Model {                      // per run; pointers into host or device memory
    int nz;
    const double *z, *p, *t;        // [nz] levels bottom-up: km, mb, K
    double xrsc;
    int cloud_term;                 // the run has a cloud deck: it is term 0
    int cld_nslot, cld_layer[NCLD];
    double cld_tcloud[NCLD], cld_lwp[NCLD], cld_nre[NCLD];
    const double *mie[6];           // q, w, g of water droplets, then of ice: [MRE][MXWV]
    int iaer, nosct, aer_nwl;
    const double *aer_wl, *aer_ext, *aer_absb, *aer_asym;   // [aer_nwl] boundary-layer spectrum
    double abaer;
    const double *aer_column;       // [nz] layers top-down: 0.55 um depth / extinction(0.55)
    int nstrat, jaer[NAERZ], strat_layer[NAERZ];
    double taerst[NAERZ];
    const double *strat_wl, *strat_tab;   // [NAERW], [4][3][NAERW]
};

SBD_SC int nterm(const Model &m)
{
    int n = m.cloud_term + ((m.iaer != 0) ? 1 : 0);
    for (int i = 0; i < m.nstrat; ++i)
        if (m.jaer[i] != 0 && m.taerst[i] > 0.0) ++n;
    return n;
}

// index j (1-based, 1..n-1) with xx(j) <= x < xx(j+1) for ascending xx (descending: mirrored); the ends clamp
SBD_SC int bracket(const double *xx, int n, double x)
{
    if (x == xx[0]) return 1;
    if (x == xx[n - 1]) return n - 1;
    const bool up = xx[n - 1] > xx[0];
    int lo = 1, hi = n;
    while (hi - lo > 1) {
        const int mid = (hi + lo) / 2;
        if (up == (x > xx[mid - 1])) lo = mid;
        else hi = mid;
    }
    return lo;
}

// Mie look-up: extinction efficiency, single-scattering albedo, asymmetry factor at (wl, re); re < 0: ice
SBD_SC void mie_lookup(const Model &m, double wl, double re, double &qc, double &wc, double &gc)
{
    const double wlmin = SCF(0.29), wlmax = SCF(333.33), eps = SCF(.000001);
    const double wmin = log(wlmin);
    const double wstep = (log(wlmax) - wmin) / (double)(MXWV - 1);
    double fw = 1.0 + (log(wl) - wmin) / wstep;
    fw = fmin(fmax(fw, 1.0), (double)(float)MXWV - eps);
    const int iw = (int)fw;
    fw = fw - (double)iw;
    double fr = 1.0 + (log(fabs(re)) / (double)0.6931471824645996f - 1.0) * 2.0;       // log(2.) is a REAL*4 there
    fr = fmin(fmax(fr, 1.0), (double)(float)MRE - eps);
    const int ir = (int)fr;
    fr = fr - (double)ir;
    const int o = (re < 0.0) ? 3 : 0;
    const int k = iw + (ir - 1) * MXWV;                     // 1-based
    auto bil = [&](const double *t) {
        return t[k - 1] * (1.0 - fw) * (1.0 - fr) + t[k] * fw * (1.0 - fr) + t[k + MXWV - 1] * (1.0 - fw) * fr + t[k + MXWV] * fw * fr;
    };
    qc = bil(m.mie[o]); wc = bil(m.mie[o + 1]); gc = bil(m.mie[o + 2]);
}

// boundary-layer aerosol at wl: extinction relative to 0.55 um, single-scattering albedo, asymmetry factor
SBD_SC void boundary_layer_at(const Model &a, double wl, double &extinc, double &wa, double &ga)
{
    extinc = 0.0; wa = 0.0; ga = 0.0;
    if (a.iaer == 0) return;
    const int n = a.aer_nwl;
    const double *w = a.aer_wl, *ext = a.aer_ext, *absb = a.aer_absb, *asym = a.aer_asym;
    const int l = bracket(w, n, wl);
    if (wl <= w[0]) {
        extinc = ext[0] * pow(w[0] / wl, a.abaer);
        wa = 1.0 - (absb[0] / ext[0]);
        ga = asym[0];
    } else if (wl >= w[n - 1]) {
        extinc = ext[n - 1] * pow(w[n - 1] / wl, a.abaer);
        wa = 1.0 - (absb[n - 1] / ext[n - 1]);
        ga = asym[n - 1];
    } else {
        const double wt = log(wl / w[l - 1]) / log(w[l] / w[l - 1]);
        extinc = ext[l - 1] * pow(ext[l] / ext[l - 1], wt);
        double absorp;
        if (absb[l - 1] > 0.0 && absb[l] > 0.0) absorp = absb[l - 1] * pow(absb[l] / absb[l - 1], wt);
        else absorp = absb[l - 1] * (1.0 - wt) + absb[l] * wt;
        if (extinc > 0.0) wa = fmax(0.0, fmin(1.0 - absorp / extinc, 1.0));
        ga = (1.0 - wt) * asym[l - 1] + wt * asym[l];
    }
}

// stratospheric model ja at wl
SBD_SC void stratospheric_at(const Model &a, int ja, double wl, double &qa, double &wa, double &ga)
{
    const double *awl = a.strat_wl, *t = a.strat_tab;
    auto s = [&](int iw, int q) { return t[(iw - 1) + (q - 1) * NAERW + (ja - 1) * 3 * NAERW]; };
    wa = 0.0;
    const int l = bracket(awl, NAERW, wl);
    if (wl <= awl[0]) {
        qa = s(1, 1) * pow(awl[0] / wl, a.abaer);
        wa = 1.0 - (s(1, 2) / s(1, 1));
        ga = s(l, 3);
    } else if (wl >= awl[NAERW - 1]) {
        qa = s(NAERW, 1) * pow(awl[0] / wl, a.abaer);          // (the reference scales from the FIRST wavelength here)
        wa = 1.0 - (s(NAERW, 2) / s(NAERW, 1));
        ga = s(NAERW, 3);
    } else {
        const double wt = log(wl / awl[l - 1]) / log(awl[l] / awl[l - 1]);
        qa = s(l, 1) * pow(s(l + 1, 1) / s(l, 1), wt);
        const double absorp = s(l, 2) * pow(s(l + 1, 2) / s(l, 2), wt);
        if (qa > 0.0) wa = fmax(0.0, fmin(1.0 - absorp / qa, 1.0));
        ga = (1.0 - wt) * s(l, 3) + wt * s(l + 1, 3);
    }
}

// The layer block of one wavelength: lay[ch * cs + l * ls], l = 0 .. nz-1 top-down, ch = 0 .. 3 + 3 nterm.
// (cs, ls): the host's blocks are [nch][nz] (cs = nz, ls = 1).
SBD_SC void point_block(const Model &m, double wl, double *lay, size_t cs, size_t ls)
{
    const int nz = m.nz;
    const int nt = nterm(m), nch = 4 + 3 * nt;
#define LAY(ch, l) lay[(size_t)(ch) * cs + (size_t)(l) * ls]           // l 0-based
    for (int ch = 0; ch < nch; ++ch)
        for (int l = 0; l < nz; ++l) LAY(ch, l) = 0.0;
    // ---- Rayleigh (rayleigh_depths): levels bottom-up in z/p/t, layer 1 = top ----
    {
        const double fit1 = SCF(9.38076e+18), fit2 = SCF(-1.08426e+09), pzero = SCF(1013.25), tzero = SCF(273.15);
        const double v = 10000.0 / wl;
        const double v2 = v * v;
        const double sig = (v * v * v * v) / (fit1 + fit2 * v2);      // (v**4 as the host's compiler forms it: left to right)
        LAY(2, 0) = sig * (m.p[nz - 1] / pzero) / (m.t[nz - 1] / tzero) * 5.0;
        for (int i = 2; i <= nz; ++i) {
            const int lev = nz - i + 1;                          // 1-based level
            const double lower = (m.p[lev - 1] / pzero) / (m.t[lev - 1] / tzero);
            const double upper = (m.p[lev] / pzero) / (m.t[lev] / tzero);
            const double dz = m.z[lev] - m.z[lev - 1];
            if (lower == upper) LAY(2, i - 1) = 0.5 * sig * dz * (lower + upper);
            else LAY(2, i - 1) = sig * dz * (upper - lower) / log(upper / lower);
        }
        if (m.xrsc != 1.0)
            for (int l = 0; l < nz; ++l) LAY(2, l) = m.xrsc * LAY(2, l);
    }
    // ---- the cloud deck (cloud_depths with one cloud per layer: a layer's sums have one term) ----
    int slot = 0;
    if (m.cloud_term) {
        const double rhoice = SCF(.917), wl55 = SCF(0.55);
        for (int i = 1; i <= m.cld_nslot; ++i) {
            if (m.cld_layer[i - 1] <= 0) continue;              // not the base of a cloud
            const int lbot = m.cld_layer[i - 1];
            int ltop = lbot;
            if (i != NCLD && m.cld_layer[i] < 0) ltop = -m.cld_layer[i];
            if (m.cld_tcloud[i - 1] == 0.0 && m.cld_lwp[i - 1] == 0.0) continue;
            for (int j = ltop; j <= lbot; ++j) {
                double reff, tcld, lwpth;
                if (ltop == lbot) {
                    reff = m.cld_nre[i - 1]; tcld = m.cld_tcloud[i - 1]; lwpth = m.cld_lwp[i - 1];
                } else {
                    auto spread_over = [&](double total, double grad) {
                        double part;
                        if (grad == 0.0) part = total / (double)(lbot - ltop + 1);
                        else {
                            part = 2.0 * total / ((double)(lbot - ltop + 1) * (1.0 + grad));
                            part = part + (double)(lbot - j) * part * (grad - 1.0) / (double)(lbot - ltop);
                        }
                        return part;
                    };
                    const double wt = (double)((float)(j - ltop) / (float)(lbot - ltop));      // float(j-ltop)/(lbot-ltop): REAL*4
                    reff = m.cld_nre[i] * pow(m.cld_nre[i - 1] / m.cld_nre[i], wt);
                    tcld = spread_over(m.cld_tcloud[i - 1], m.cld_tcloud[i]);
                    lwpth = spread_over(m.cld_lwp[i - 1], m.cld_lwp[i]);
                }
                double qc, wc, gc;
                mie_lookup(m, wl, reff, qc, wc, gc);
                double taucld = LAY(0, j - 1);                   // (0: one cloud per layer)
                if (m.cld_tcloud[i - 1] != 0.0) {               // optical depth given at 0.55 um: scale with the efficiency
                    double q550, w550, g550;
                    mie_lookup(m, wl55, reff, q550, w550, g550);
                    taucld = tcld * qc / q550 + taucld;
                } else if (lwpth != 0.0) {                      // water path and radius: tau = 3 Q LWP / (4 r rho)
                    if (reff < 0.0) taucld = -0.75 * qc * lwpth / reff / rhoice + taucld;
                    else taucld = 0.75 * qc * lwpth / reff + taucld;
                }
                const double wcld = (wc + 0.0) / 1.0;           // (wcld(j)/cnt(j), cnt = 1)
                LAY(0, j - 1) = taucld;
                LAY(4, j - 1) = gc;
                LAY(5, j - 1) = taucld * wcld;
                LAY(6, j - 1) = 1.0;
                // (the scattering depth's cloud share, dtauc * wcld, is formed below from these two: keep wcld in `scat`'s channel)
                LAY(3, j - 1) = wcld;
            }
        }
        slot = 1;
    }
    // ---- aerosols (aerosol_depths): boundary layer, then the stratospheric layers in their order ----
    // (channel 1 = DTAUA; WAER is carried in the last term channel written per layer: it is not part of the block, so it
    //  lives in a local array)
    double waer[66];
    for (int l = 0; l < nz; ++l) waer[l] = 0.0;
    if (m.iaer != 0) {
        double extinc, wa, ga;
        boundary_layer_at(m, wl, extinc, wa, ga);
        if (m.nosct == 1) extinc = extinc * (1.0 - wa);
        if (m.nosct == 3) extinc = extinc * (1.0 - wa * ga);
        if (m.nosct != 0) { wa = 0.0; ga = 0.0; }
        const int c0 = 4 + 3 * slot;
        for (int l = 0; l < nz; ++l) {
            const double dt = extinc * m.aer_column[l];
            LAY(1, l) = dt;
            waer[l] = wa;
            LAY(c0, l) = ga; LAY(c0 + 1, l) = dt; LAY(c0 + 2, l) = wa;
        }
        ++slot;
    }
    for (int i = 0; i < m.nstrat; ++i) {
        if (m.jaer[i] != 0 && m.taerst[i] > 0.0) {
            const int nl = m.strat_layer[i];                     // 1-based layer
            double extinc, wa, ga;
            stratospheric_at(m, m.jaer[i], wl, extinc, wa, ga);
            const double dt = m.taerst[i] * extinc;
            const int c0 = 4 + 3 * slot;
            LAY(c0, nl - 1) = ga; LAY(c0 + 1, nl - 1) = dt; LAY(c0 + 2, nl - 1) = wa;
            ++slot;
            const double da = LAY(1, nl - 1);
            waer[nl - 1] = (waer[nl - 1] * da + wa * dt) / (da + dt);
            LAY(1, nl - 1) = da + dt;
        }
    }
    // ---- the scattering depth: dtauc * wcld + dtaua * waer + dtaur ----
    for (int l = 0; l < nz; ++l) {
        const double wcld = m.cloud_term ? LAY(3, l) : 0.0;
        LAY(3, l) = LAY(0, l) * wcld + LAY(1, l) * waer[l] + LAY(2, l);
    }
#undef LAY
}

}  // namespace scat
}  // namespace sbd
