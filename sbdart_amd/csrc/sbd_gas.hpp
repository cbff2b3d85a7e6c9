// The gas part of SBDART's band model for one wavelength, as host/device code: LOWTRAN7's 20 cm-1 band model and
// continua along a vertical and a slant path, the three-term k-distribution fit, the Newton slant-path correction and
// depthscl's KDIST policy (reference: taugas.f:2236-2534 taugas with its spectral look-ups 2538-6821, 1802-1920 kdistr,
// 7392-7510 gasset, 7650-7692 taucor, 7512-7648 depthscl).  SURVEY 8f row N1 on the device: north_star's "per-wavelength
// optical depths precomputed into coalesced HBM arrays".
//
// Written from the Fortran host's restatement (sbdart_amd/fortran/sbd_gas_mod.f90, bit-equal to the live reference:
// tests/test_band_model.py) with the SAME sequence of roundings: every sum, product and quotient in the same order, no
// contraction (this header is compiled with fp-contract off), integer powers as products, and the reference's
// constants typed as it typed them -- a bare literal there is REAL*4, widened (F(x) below).  On the HOST (glibc's libm,
// what the Fortran runtime calls) the results are bit-equal to the Fortran host's: tests/test_gas_device.py pins that
// through sbd_gas_terms_host.  On the DEVICE exp / log / log10 / pow come from the device math library: the same
// formulas to within a few ulps per call (tests/test_gas_device.py states and checks the bound on the work items).
#pragma once
#include <cmath>
#include <cstdint>
#include "sbd_gas_types.hpp"
#pragma clang fp contract(off)      // one rounding per operation, like the reference's x86-64 object code

#if defined(__HIPCC__)
#define SBD_HD __host__ __device__
#else
#define SBD_HD
#endif

namespace sbd {
namespace gas {

#define F(x) ((double)(x##f))       // a REAL*4 literal of the reference, widened

// first absorber-amount slot of each molecule's bands minus one (slot - base = band number)
SBD_HD inline int slot_base(int m)
{
    constexpr int b[NMOL] = {16, 35, 30, 46, 43, 45, 49, 53, 55, 54, 51};
    return b[m];
}
// LOWTRAN7 band limits: molecule (1-based), first and last wavenumber (cm-1, 5 cm-1 grid), absorber-amount slot
constexpr int NRANGE = 64;
SBD_HD inline void band_range(int k, int &mol, int &v0, int &v1, int &slot)
{
    constexpr int r[NRANGE][4] = {
        {1, 0, 345, 17}, {1, 350, 1000, 18}, {1, 1005, 1640, 19}, {1, 1645, 2530, 20}, {1, 2535, 3420, 21},
        {1, 3425, 4310, 22}, {1, 4315, 6150, 23}, {1, 6155, 8000, 24}, {1, 8005, 9615, 25}, {1, 9620, 11540, 26},
        {1, 11545, 13070, 27}, {1, 13075, 14860, 28}, {1, 14865, 16045, 29}, {1, 16340, 17860, 30},
        {3, 0, 200, 31}, {3, 515, 1275, 32}, {3, 1630, 2295, 33}, {3, 2670, 2845, 34}, {3, 2850, 3260, 35},
        {2, 425, 835, 36}, {2, 840, 1440, 37}, {2, 1805, 2855, 38}, {2, 3070, 3755, 39}, {2, 3760, 4065, 40},
        {2, 4530, 5380, 41}, {2, 5905, 7025, 42}, {2, 7395, 7785, 43}, {2, 8030, 8335, 43}, {2, 9340, 9670, 43},
        {5, 0, 175, 44}, {5, 1940, 2285, 45}, {5, 4040, 4370, 45}, {6, 1065, 1775, 46}, {6, 2345, 3230, 46},
        {6, 4110, 4690, 46}, {6, 5865, 6135, 46}, {4, 0, 120, 47}, {4, 490, 775, 48}, {4, 865, 995, 48},
        {4, 1065, 1385, 48}, {4, 1545, 2040, 48}, {4, 2090, 2655, 48}, {4, 2705, 2865, 49}, {4, 3245, 3925, 49},
        {4, 4260, 4470, 49}, {4, 4540, 4785, 49}, {4, 4910, 5165, 49}, {7, 0, 265, 50}, {7, 7650, 8080, 51},
        {7, 9235, 9490, 51}, {7, 12850, 13220, 51}, {7, 14300, 14600, 51}, {7, 15695, 15955, 51},
        {7, 49600, 52710, 51}, {11, 0, 385, 52}, {11, 390, 2150, 53}, {8, 1700, 2005, 54}, {10, 580, 925, 55},
        {10, 1515, 1695, 55}, {10, 2800, 2970, 55}, {9, 0, 185, 56}, {9, 400, 650, 57}, {9, 950, 1460, 57},
        {9, 2415, 2580, 57}};
    mol = r[k][0]; v0 = r[k][1]; v1 = r[k][2]; slot = r[k][3];
}

struct Spectrum {                   // everything that depends on the wavelength only (sbd_gas_mod's gas_spectrum)
    double v;
    double self296, self260, foreign, radfn0, radfn1, far_wing;
    double n2, hno3, o2_herzberg, o2_s, o2_a, o2_b, o4;
    double oz[3];
    int slot[NMOL];                 // 1-based absorber slot of the band each molecule is in, -1 = none
    double cp[NMOL], bs[NMOL], ba[NMOL], bb[NMOL], bc[NMOL];
};

// value of a 10 cm-1 table that starts at v1: the entry at v, or the mean of the two around it (taugas.f:3854-3871)
SBD_HD inline double ten_wavenumber_table(const Tab &t, double v1, double v)
{
    const int i = (int)((v - v1) / F(10.) + F(1.00001));
    if (i >= t.n) return 0.0;
    double c = t.p[i - 1];
    if (((int)v) % 10 > 0) c = (t.p[i - 1] + t.p[i]) / F(2.);
    return c;
}

// band-model coefficient of molecule m (0-based) at wavenumber v; -20 outside every region (taugas.f:6416-6456)
SBD_HD inline double band_coefficient(const Tables &T, int m, double v)
{
    const int iv = (int)v;
    int before = 0;
    for (int r = 0; r < T.lo[m].n; ++r) {
        const int lo = T.lo[m].p[r], hi = T.hi[m].p[r];
        if (lo == -999) break;
        if (iv >= lo && iv <= hi) return T.cp[m].p[before + (iv - lo) / 5];
        before = before + (hi - lo) / 5 + 1;
    }
    return -F(20.0);
}

SBD_HD inline void spectrum_at(const Tables &T, double wl, double xo4, Spectrum &s)
{
    const double bigexp = F(87.), fraco2 = F(.209), fracn2 = F(.781), effn2 = F(.2);
    const double v = F(10000.) / wl;
    const int iv5 = 5 * ((int)(F(10000.0) / wl) / 5);
    s.v = v;
    s.self296 = s.self260 = s.foreign = s.radfn0 = s.radfn1 = s.far_wing = 0.0;
    s.n2 = s.hno3 = s.o2_herzberg = s.o2_s = s.o2_a = s.o2_b = s.o4 = 0.0;
    s.oz[0] = s.oz[1] = s.oz[2] = 0.0;
    for (int m = 0; m < NMOL; ++m) { s.slot[m] = -1; s.cp[m] = -F(20.); s.bs[m] = s.ba[m] = s.bb[m] = s.bc[m] = 0.0; }
    // ---- water-vapour continuum: self at 296 K and 260 K, foreign; radiation field factors ----
    s.self296 = ten_wavenumber_table(T.self296, -20.0, v);
    s.self260 = ten_wavenumber_table(T.self260, -20.0, v);
    s.foreign = ten_wavenumber_table(T.foreign, -20.0, v);
    if (s.self296 > 0.0) {
        const double alpha2 = (double)(200.f * 200.f);                         // 200.**2 in REAL*4
        const double xh2o = (F(1.) - F(0.2333) * (alpha2 / ((v - F(1050.)) * (v - F(1050.)) + alpha2)));
        s.self296 = s.self296 * xh2o;
        s.self260 = s.self260 * xh2o;
    }
    if ((v / F(0.6952)) / 260.0 <= bigexp) {
        double xd = exp(-v / (296.0 * F(0.6952)));
        s.radfn0 = v * (F(1.) - xd) / (F(1.) + xd);
        xd = exp(-v / (260.0 * F(0.6952)));
        s.radfn1 = v * (F(1.) - xd) / (F(1.) + xd);
    } else {
        s.radfn0 = v;
        s.radfn1 = v;
    }
    {
        // -log(1.025*3.159e-8) and -log(8.97e-6): REAL*4 constant expressions of the reference, folded by its compiler
        // to the correctly rounded single-precision values 17.2457332611084 (0x40313EE860000000) and 11.621624946594238
        // (0x40273E45A0000000) -- written out so that host and device hold the same bits whatever their logf does
        const double ya = exp(0x1.13ee86p+4 + F(2.75e-4) * v);
        const double yb = exp(0x1.73e45ap+3 + F(1.300e-3) * v);
        s.far_wing = F(1.) / (ya + yb);
    }
    // ---- nitrogen continuum, 2080-2740 cm-1 ----
    if (v >= F(2080.) && v <= F(2740.)) {
        const int i = (int)v;
        s.n2 = T.n2.p[(i - 2080) / 5];
    }
    // ---- nitric acid, three windows ----
    if (v >= F(850.0) && v <= F(920.0)) {
        const int i = (int)((v - F(845.)) / F(5.));
        s.hno3 = T.h1.p[i - 1];
    } else if (v >= F(1275.0) && v <= F(1350.0)) {
        const int i = (int)((v - F(1270.)) / F(5.));
        s.hno3 = T.h2.p[i - 1];
    } else if (v >= F(1675.0) && v <= F(1735.0)) {
        const int i = (int)((v - F(1670.)) / F(5.));
        s.hno3 = T.h3.p[i - 1];
    }
    // ---- oxygen: Herzberg continuum (analytic), 1395-1760 cm-1 collision-induced band ----
    if (v > F(36000.00)) {
        double corr = 0.0;
        if (v <= F(40000.)) corr = ((F(40000.) - v) / F(4000.)) * F(7.917e-27);
        const double rlosch = (double)(2.6868e24f * 1.0e-5f);
        const double yratio = v / F(48811.0);
        const double ly = log(yratio);
        s.o2_herzberg = (F(6.884e-24) * (yratio) * exp(-F(69.738) * (ly * ly)) - corr) * rlosch;
    }
    if (!(v < 1395 || v > 1760)) {
        const int i = (int)((v - 1395.0) / 5.0 + F(1.00001));
        double a = 0.0, b = 0.0, c = 0.0;
        if (i >= 1 && i <= T.o2s0.n) {
            c = T.o2s0.p[i - 1];
            a = T.o2a.p[i - 1];
            b = T.o2b.p[i - 1];
        }
        s.o2_a = a;
        s.o2_b = a * a / F(2.) + b;
        s.o2_s = c / F(0.20946);
    }
    // ---- O2-O2 / O2-N2 collision complexes, 1 nm table from 335 nm ----
    {
        const double wnm = F(1000.) * wl;
        int inm = (int)wnm;
        const double f = wnm - inm;
        inm = inm - 335 + 1;
        if (inm >= 1 && inm <= 1015) {
            double factor = fraco2 * fraco2;
            if (wl > F(1.2)) factor = fraco2 * (fraco2 + effn2 * fracn2);
            s.o4 = xo4 * factor * (T.o4.p[inm - 1] * (F(1.) - f) + T.o4.p[inm] * f);
        }
    }
    // ---- ozone: Hartley (UV), Hartley-Huggins with temperature terms, Chappuis ----
    if (v > 40800) {
        const int n = T.o3uv.n;
        double c = 0.0;
        int i = (int)((v - 40800.0) / 100.0 + F(1.00001));
        if (i >= 1 && i <= n) {
            const double vr = i * 100.0 + 40800.0;
            if (vr <= (v + F(.1)) && vr >= (v - F(.1))) {
                c = T.o3uv.p[i - 1];
            } else {
                if (i == n) i = n - 1;
                const double am = (T.o3uv.p[i] - T.o3uv.p[i - 1]) / 100.0;
                const double c0 = T.o3uv.p[i - 1] - am * vr;
                c = am * v + c0;
            }
        }
        s.oz[0] = F(.269) * c;
    } else if (v > 24370) {
        const int i = (int)((v - 27370.0) / 5.0 + F(1.00001));
        if (i >= 1 && i <= T.hh0.n) {
            const double t = T.hh0.p[i - 1];
            s.oz[0] = F(.269) * t;
            s.oz[1] = t * T.hh1.p[i - 1];
            s.oz[2] = t * T.hh2.p[i - 1];
        }
    } else if (v >= F(13000.) && v <= 24200) {
        const double xi = (v - F(13000.0)) / F(200.0) + F(1.);
        const int n = (int)(xi + F(1.001));
        s.oz[0] = T.chap.p[n - 1] + (xi - (double)(float)n) * (T.chap.p[n - 1] - T.chap.p[n - 2]);
    }
    // ---- band model: coefficient, band and band parameters of every molecule ----
    for (int m = 0; m < NMOL; ++m) s.cp[m] = band_coefficient(T, m, v);
    for (int k = 0; k < NRANGE; ++k) {
        int mol, v0, v1, slot;
        band_range(k, mol, v0, v1, slot);
        if (iv5 < v0 || iv5 > v1) continue;
        const int m = mol - 1;
        s.slot[m] = slot;
        const int band = slot - slot_base(m);
        s.bs[m] = T.bs[m].p[band - 1];
        s.ba[m] = T.ba[m].p[band - 1];
        s.bb[m] = T.bb[m].p[band - 1];
        s.bc[m] = T.bc[m].p[band - 1];
    }
    if (iv5 >= 49600 && iv5 <= 52710) s.bs[6] = F(.4704);           // Schumann-Runge: its own band-model exponent
    if (v > 49600) {                                                 // ... and coefficients
        s.cp[6] = -F(20.);
        const int i = (int)((v - 49600.0) / 5.0 + F(1.0001));
        if (i >= 1 && i <= T.schrun.n) s.cp[6] = T.schrun.p[i - 1];
    }
}

// Continuum and band-model ("line") optical depth of every layer for a path whose zenith cosine at the ground is amu0
// (spherical-shell air mass per layer); layer 1 is the top (taugas.f:2236-2534).  uu [nz][MXQ]: absorber amounts above
// each level (bottom-up levels), z [nz] altitudes.  dtau_cont / dtau_line: [nz], index 0 = top layer.
// Only the slots the wavelength uses are carried along: the continuum's thirteen and the band slot of each molecule.
// Work arrays are STRIDED (element i of an array a is a[i * st]): one thread per wavelength on the device keeps
// its arrays interleaved with its neighbours' (coalesced), the host walks them with st = 1.
SBD_HD inline void path_depths(const Spectrum &s, const double *uu, double amu0, const double *z, int nz, double re_earth,
                               double *dtau_cont, double *dtau_line, const size_t st)
{
    const double awlmax = F(20.), wfac = F(1.e-20);
    // continuum slots (1-based in the reference): 1 2 3 4 5 8 9 10 11 58 59 60 63
    constexpr int cs[13] = {1, 2, 3, 4, 5, 8, 9, 10, 11, 58, 59, 60, 63};
    double wc[13], wb[NMOL];
    double zim = z[nz - 1];
    double prev_c = 0.0, prev_l = 0.0;
    for (int i = nz; i >= 1; --i) {
        const int im = nz - i + 1;
        const double zi = z[i - 1];
        const double zbar = F(0.5) * (zi + zim);
        zim = zi;
        const double *ui = uu + (size_t)(i - 1) * MXQ;
        if (i == nz) {
            const double r = re_earth / (re_earth + zi);
            const double af = sqrt(F(1.) - (F(1.) - amu0 * amu0) * (r * r));
            for (int c = 0; c < 13; ++c) wc[c] = ui[cs[c] - 1] / af;
            for (int m = 0; m < NMOL; ++m) wb[m] = (s.slot[m] > 0) ? ui[s.slot[m] - 1] / af : 0.0;
        } else {
            const double *un = uu + (size_t)i * MXQ;
            const double r = re_earth / (re_earth + zbar);
            const double af = sqrt(F(1.) - (F(1.) - amu0 * amu0) * (r * r));
            for (int c = 0; c < 13; ++c) wc[c] = wc[c] + (ui[cs[c] - 1] - un[cs[c] - 1]) / af;
            for (int m = 0; m < NMOL; ++m)
                if (s.slot[m] > 0) wb[m] = wb[m] + (ui[s.slot[m] - 1] - un[s.slot[m] - 1]) / af;
        }
        const double w1 = wc[0], w2 = wc[1], w3 = wc[2], w4 = wc[3], w5 = wc[4], w8 = wc[5], w9 = wc[6], w10 = wc[7],
                     w11 = wc[8], w58 = wc[9], w59 = wc[10], w60 = wc[11], w63 = wc[12];
        const double uniform = +s.o4 * w3 + s.n2 * w4 + s.o2_s * (w63 + s.o2_a * (w1 - 220 * w63) + s.o2_b * w2) + s.o2_herzberg * w58;
        const double h2o = s.self296 * s.radfn0 * (wfac * w5) + ((s.self260 * s.radfn1) - (s.self296 * s.radfn0)) * (wfac * w9)
                           + (s.foreign + s.far_wing) * s.radfn0 * (wfac * w10);
        const double ozone = s.oz[0] * w8 + s.oz[1] * w59 + s.oz[2] * w60;
        const double trace = s.hno3 * w11;
        const double cum_c = uniform + h2o + ozone + trace;
        double cum_l = 0.0;
        for (int k = 0; k < NMOL; ++k) {
            if (s.slot[k] > 0) {
                if (s.cp[k] > -awlmax && wb[k] > F(1.e-20)) {
                    double awl = s.bs[k] * (s.cp[k] + log10(wb[k]));
                    awl = fmin(awl, awlmax);
                    cum_l = cum_l + pow(10.0, awl);
                }
            }
        }
        if (im == 1) {
            dtau_cont[0] = cum_c;
            dtau_line[0] = cum_l;
        } else {
            dtau_cont[(size_t)(im - 1) * st] = cum_c - prev_c;
            dtau_line[(size_t)(im - 1) * st] = cum_l - prev_l;
        }
        prev_c = cum_c;
        prev_l = cum_l;
    }
}

// LOWTRAN7 three-term exponential-sum fit of the band transmission: per layer (0 = top) the optical depth increments of
// the three terms dtk [MK][nz], their running sums at the bottom tk_bot [MK], the layer's weights wtk [MK][nz]
// (taugas.f:1802-1920)
SBD_HD inline void three_term_fit(const Spectrum &s, const double *uu, int nz, double *dtk, double *wtk, double *tk_bot, const size_t st)
{
    const double fac[MK] = {F(1.0), F(0.09), F(0.015)};
    double strength[MK][NMOL], share[MK][NMOL], cp1[NMOL];
    for (int m = 0; m < NMOL; ++m) {
        cp1[m] = pow(10.0, s.cp[m]);
        for (int k = 0; k < MK; ++k) { strength[k][m] = 0.0; share[k][m] = 0.0; }
        if (s.slot[m] > 0) {
            for (int k = 0; k < MK; ++k) strength[k][m] = fac[k] * s.bc[m];
            share[0][m] = s.ba[m];
            share[1][m] = s.bb[m];
            share[2][m] = F(1.) - s.ba[m] - s.bb[m];
        }
    }
    for (int k = 0; k < MK; ++k) {
        for (int n = 1; n <= nz; ++n) {
            const int lev = nz - n + 1;
            double d = 0.0, weighted = 0.0;
            for (int m = 0; m < NMOL; ++m) {
                const int ib = s.slot[m];
                if (ib < 0) continue;
                double duu;
                if (lev == nz) duu = uu[(size_t)(lev - 1) * MXQ + ib - 1];
                else duu = uu[(size_t)(lev - 1) * MXQ + ib - 1] - uu[(size_t)lev * MXQ + ib - 1];
                const double wpth = duu * strength[k][m];
                d = d + wpth * cp1[m];
                weighted = weighted + wpth * cp1[m] * share[k][m];
            }
            dtk[(size_t)(k * nz + n - 1) * st] = d;
            double w = (double)(1.f / 3.f);
            if (d != 0) w = weighted / d;
            wtk[(size_t)(k * nz + n - 1) * st] = w;
        }
    }
    double run[MK] = {0.0, 0.0, 0.0};
    for (int n = 0; n < nz; ++n) {
        const double w0 = wtk[(size_t)n * st], w1 = wtk[(size_t)(nz + n) * st], w2 = wtk[(size_t)(2 * nz + n) * st];
        const double total = w0 + w1 + w2;
        wtk[(size_t)n * st] = w0 / total;
        wtk[(size_t)(nz + n) * st] = w1 / total;
        wtk[(size_t)(2 * nz + n) * st] = w2 / total;
        for (int k = 0; k < MK; ++k) {
            const double d = dtk[(size_t)(k * nz + n) * st];
            run[k] = (n == 0) ? d : d + run[k];
        }
    }
    for (int k = 0; k < MK; ++k) tk_bot[k] = run[k];
}

// factor cf that makes the three-term transmission sum(g exp(-cf tau/amu)) equal exp(-utau): Newton iteration from
// cf = 1 (taugas.f:7650-7692).  Returns false when the iteration does not converge (the reference stops).
SBD_HD inline bool match_slant_transmission(const double *gwk, const double *tau, double amu, double utau, double &cf)
{
    cf = F(1.);
    if (utau > F(12.0)) return true;
    for (int it = 0; it < 20; ++it) {
        const double e0 = exp(-cf * tau[0] / amu), e1 = exp(-cf * tau[1] / amu), e2 = exp(-cf * tau[2] / amu);
        const double ff = gwk[0] * e0 + gwk[1] * e1 + gwk[2] * e2;
        const double f = log(ff) + utau;
        if (fabs(f) < F(0.000001)) return true;
        const double fp = -(gwk[0] * e0 * tau[0] + gwk[1] * e1 * tau[1] + gwk[2] * e2 * tau[2]) / (ff * amu);
        cf = cf + (-f / fp);
    }
    return false;
}

// The gas terms of one wavelength (gasset, taugas.f:7392-7510): number of k-terms nk (1 or 3), their weights gwk [MK],
// per layer the continuum depth dtau_cont [nz] and for each term the line depth dtauk [2 MK][nz] -- rows 0..2 as
// fitted, rows 3..5 corrected to reproduce the slant-path band transmission at the solar zenith angle.
// work: 9 nz doubles; dtau_cont, dtauk and work strided by st.  Returns 0, or 1 when the slant-path Newton iteration
// failed (TAUCOR's stop).
SBD_HD inline int gas_terms(int kdist, const Spectrum &s, const double *uu, double amu0, const double *z, int nz,
                            double re_earth, int &nk, double *gwk, double *dtauk, double *dtau_cont, double *work, const size_t st)
{
    double *dtcs = work, *dtls = work + (size_t)nz * st, *dtlv = work + (size_t)2 * nz * st, *dtk = work + (size_t)3 * nz * st,
           *wtk = work + (size_t)6 * nz * st;                                   // dtk, wtk: [MK][nz]
    path_depths(s, uu, F(1.), z, nz, re_earth, dtau_cont, dtlv, st);
    if (amu0 > 0.0) path_depths(s, uu, amu0, z, nz, re_earth, dtcs, dtls, st);
    else for (int j = 0; j < nz; ++j) dtls[(size_t)j * st] = dtlv[(size_t)j * st];
    for (int k = 0; k < MK; ++k) gwk[k] = 0.0;
    for (int j = 0; j < 2 * MK * nz; ++j) dtauk[(size_t)j * st] = 0.0;
    nk = 1;
    gwk[0] = F(1.);
    double sum_lv = 0.0;
    for (int j = 0; j < nz; ++j) sum_lv = sum_lv + dtlv[(size_t)j * st];
    if (!(kdist == 0 || sum_lv < F(.01))) {
        double tk_bot[MK];
        three_term_fit(s, uu, nz, dtk, wtk, tk_bot, st);
        if (!(fmax(fmax(tk_bot[0], tk_bot[1]), tk_bot[2]) < F(0.01))) {
            nk = MK;
            for (int k = 0; k < MK; ++k) {
                double g = 0.0;
                for (int j = 0; j < nz; ++j) g = g + dtlv[(size_t)j * st] * wtk[(size_t)(k * nz + j) * st];
                gwk[k] = g;
            }
            const double wnorm = gwk[0] + gwk[1] + gwk[2];
            if (wnorm == 0) gwk[0] = F(1.);
            else for (int k = 0; k < MK; ++k) gwk[k] = gwk[k] / wnorm;
        }
    }
    int rc = 0;
    if (kdist == 0 || nk == 1) {
        for (int j = 0; j < nz; ++j) {
            dtauk[(size_t)j * st] = dtlv[(size_t)j * st];
            dtauk[(size_t)(MK * nz + j) * st] = amu0 * dtls[(size_t)j * st];
        }
    } else {
        for (int k = 0; k < MK; ++k)
            for (int j = 0; j < nz; ++j) {
                const double d = dtk[(size_t)(k * nz + j) * st];
                dtauk[(size_t)(k * nz + j) * st] = d;
                dtauk[(size_t)((MK + k) * nz + j) * st] = d;
            }
        if (kdist >= 2 && amu0 > 0.0) {
            double slant = 0.0, corrected[MK] = {0.0, 0.0, 0.0};
            for (int j = 0; j < nz; ++j) {
                slant = slant + dtls[(size_t)j * st];
                double fit[MK], cf;
                for (int k = 0; k < MK; ++k) { fit[k] = dtk[(size_t)(k * nz + j) * st]; corrected[k] = fit[k] + corrected[k]; }
                if (!match_slant_transmission(gwk, corrected, amu0, slant, cf)) rc = 1;
                for (int k = 0; k < MK; ++k) {
                    dtauk[(size_t)((MK + k) * nz + j) * st] = corrected[k] * (cf - F(1.0)) + fit[k];
                    corrected[k] = cf * corrected[k];
                }
            }
        }
    }
    if (amu0 <= 0.0)
        for (int j = 0; j < nz; ++j) dtauk[(size_t)(MK * nz + j) * st] = dtlv[(size_t)j * st];
    return rc;
}

// weight that fades the slant-path correction out (rolloff, taugas.f:7625-7647)
SBD_HD inline double correction_weight(double wl, double tsc)
{
    const double wllo = F(3.9), wlhi = F(4.1);
    double ramp = (wlhi - wl) / (wlhi - wllo);
    ramp = fmax(fmin(1.0, ramp), 0.0);
    ramp = ramp * exp(F(1.) - fmax(tsc, 1.0));
    return ramp;
}

// depthscl's gas depth of k-term k (0-based) with the slant-path correction policy KDIST (taugas.f:7550-7590) and the
// term's weight wt: dtaug [nz] (contiguous).  dtaur, dtauc, dtaua [nz] (contiguous): the point's Rayleigh, cloud and
// aerosol depths -- their running sum tsc = tsc + dtaur + dtauc + dtaua drives the roll-off.  dtauk, dtau_cont strided.
SBD_HD inline void scaled_gas_depth(int kdist, int nk, int k, double wl, const double *dtaur, const double *dtauc, const double *dtaua,
                                    const double *dtauk, const double *dtau_cont, int nz, const size_t st, const double *gwk,
                                    double *dtaug, double &wt)
{
    wt = gwk[k];
    if (kdist == 0 || nk == 1) {
        wt = F(1.);
        double tsc = 0.0, tglv = 0.0, tgls = 0.0;
        for (int l = 0; l < nz; ++l) {
            tglv = tglv + dtauk[(size_t)l * st];
            tgls = tgls + dtauk[(size_t)(MK * nz + l) * st];
            tsc = tsc + dtaur[l] + dtauc[l] + dtaua[l];
            double afac = F(1.);
            if (tglv > F(.001)) afac = tgls / tglv;
            const double ramp = correction_weight(wl, tsc);
            afac = afac * ramp + F(1.) - ramp;
            dtaug[l] = dtau_cont[(size_t)l * st] + dtauk[(size_t)l * st] * afac;
        }
    } else if (kdist == 1) {
        for (int l = 0; l < nz; ++l) dtaug[l] = dtau_cont[(size_t)l * st] + dtauk[(size_t)(k * nz + l) * st];
    } else if (kdist == 2) {
        for (int l = 0; l < nz; ++l) dtaug[l] = dtau_cont[(size_t)l * st] + dtauk[(size_t)((k + MK) * nz + l) * st];
    } else {
        double tsc = 0.0;
        for (int l = 0; l < nz; ++l) {
            tsc = tsc + dtaur[l] + dtauc[l] + dtaua[l];
            const double ramp = correction_weight(wl, tsc);
            dtaug[l] = dtau_cont[(size_t)l * st] + dtauk[(size_t)(k * nz + l) * st] * (F(1.) - ramp) + dtauk[(size_t)((k + MK) * nz + l) * st] * ramp;
        }
    }
}

// everything for one wavelength: spectrum, gas terms, the terms' depths.  slots [MK][nz] contiguous (the terms' gas
// depths, rows beyond nk untouched), wt [MK]; ws: 16 nz doubles strided by st (dtauk 6 nz | dtau_cont nz | work 9 nz).
// lay: the point's layer block [nch][nz] (channels dtauc, dtaua, dtaur first).  Returns gas_terms' code.
SBD_HD inline int point_gas(const Tables &T, int kdist, double xo4, const double *uu, const double *z, int nz, double re_earth,
                            double wl, double amu0, const double *lay, double *ws, const size_t st, int &nk, double *wt, double *slots)
{
    Spectrum s;
    spectrum_at(T, wl, xo4, s);
    double gwk[MK];
    double *dtauk = ws, *dtau_cont = ws + (size_t)6 * nz * st, *work = ws + (size_t)7 * nz * st;
    const int rc = gas_terms(kdist, s, uu, amu0, z, nz, re_earth, nk, gwk, dtauk, dtau_cont, work, st);
    for (int k = 0; k < MK; ++k) wt[k] = 0.0;
    for (int k = 0; k < nk; ++k)
        scaled_gas_depth(kdist, nk, k, wl, lay + 2 * nz, lay, lay + nz, dtauk, dtau_cont, nz, st, gwk, slots + (size_t)k * nz, wt[k]);
    return rc;
}

#undef F

}  // namespace gas
}  // namespace sbd
