// Bidirectional surfaces (LAMBER = false): BDREF's three models and SURFAC's Fourier components.
//
//   BDREF (spectra.f:249-296) dispatches on the surface model of the run: 1 ocean (seabdrf, spectra.f:421-465:
//   foam + sun glint of a wind-roughened sea, slope distribution averaged over the wind direction, Fresnel
//   reflection of the facet + sub-surface reflectance), 2 Hapke's soil model (hapkbdrf, spectra.f:298-348),
//   3 Ross-thick / Li-sparse kernels (rtlsbdrf, spectra.f:350-419).  What the ocean model derives from the
//   wavelength alone -- the water's refractive index and its sub-surface reflectance, table look-ups of the
//   host's band model -- comes with the work item (sbd_batch_in::bitem); everything here is geometry.
//
//   SURFAC (disort.f:3639-3918), per (work item, azimuth mode m): BDR(iq, jq) = (2 - delta_m0)/2 * the 50-point
//   Gauss sum over the relative azimuth of BDREF(mu_iq, mu_jq, phi) cos(m phi) at the quadrature angles, column 0
//   for the beam; RMU(iu, iq) the same at the upward user angles; for m = 0 the directional emissivities
//   BEM(iq), EMU(iu) = 1 - the flux albedo at that reflection angle (a 50 x 25 double sum).  CHEKIN's test of the
//   surface (disort.f:5080-5096: the flux albedo DREF, disort.f:5178-5284, at 101 incidence cosines must lie in
//   [0,1]) runs here too and raises SBD_ST_ERR_INPUT.
//
// Mapping: one block of 256 threads per (item, mode) -- or per mode only, once per engine, when the model does
// not depend on the wavelength (Hapke, Ross-Li).  A thread owns one entry of BDR / RMU and walks its 50 azimuths
// in the reference's order; the emissivity and flux-albedo sums are cut into their 50 inner sums (one per thread,
// 25 terms each, reference order), parked in LDS and added in order by one thread per target.
#pragma once
#include "sbd_common.hpp"

namespace sbd {

constexpr int kSurfGauss = 50;   // NMUG of SURFAC / DREF

struct BrdfModel {
    int ibdrf;
    double bp[8];                // run parameters (include/sbdart_amd.h: sbd_run_cfg::bpar)
    double nr, ni, rsw;          // ocean: this wavelength's water constants
};

#define SBD_PI_PARAMS 3.1415926536   /* params.f:29 */

// Fresnel reflection coefficient of a water facet (fresnel, spectra.f:1320-1356)
__host__ __device__ inline double surf_fresnel(double nr, double ni, double coschi, double sinchi)
{
    const double t = nr * nr - ni * ni - sinchi * sinchi;
    const double a1 = fabs(t);
    const double a2 = sqrt(t * t + 4.0 * nr * nr * ni * ni);
    const double u = sqrt(0.5 * (a1 + a2));
    const double v = sqrt(0.5 * (-a1 + a2));
    const double rr2 = ((coschi - u) * (coschi - u) + v * v) / ((coschi + u) * (coschi + u) + v * v);
    const double b1 = (nr * nr - ni * ni) * coschi;
    const double b2 = 2.0 * nr * ni * coschi;
    const double rl2 = ((b1 - u) * (b1 - u) + (b2 + v) * (b2 + v)) / ((b1 + u) * (b1 + u) + (b2 - v) * (b2 - v));
    return (rr2 + rl2) / 2.0;
}

// sun glint (sunglint, spectra.f:1224-1318)
__host__ __device__ inline double surf_sunglint(double wndspd, double nr, double ni, double csin, double cvin, double phi)
{
    const double pi = SBD_PI_PARAMS;
    const double cs = fmax(csin, 0.05), cv = fmax(cvin, 0.05);
    const double ss = sqrt(1.0 - cs * cs), sv = sqrt(1.0 - cv * cv);
    const double zx = -sv * sin(pi - phi) / (cs + cv);
    const double zy = (ss + sv * cos(pi - phi)) / (cs + cv);
    const double tilt = atan(sqrt(zx * zx + zy * zy));
    const double sigmac = SBD_F32(0.003) + SBD_F32(0.00192) * wndspd;
    const double sigmau = SBD_F32(0.00316) * wndspd;
    const double c40 = SBD_F32(0.40), c22 = SBD_F32(0.12), c04 = SBD_F32(0.23);
    const double zx2 = zx * zx, zy2 = zy * zy, zx4 = zx2 * zx2, zy4 = zy2 * zy2;
    const double axe2 = 0.5 * (zx2 + zy2) / sigmac;
    const double axn2 = 0.5 * (zx2 + zy2) / sigmau;
    const double axe4 = (3.0 * zx4 + 6.0 * zx2 * zy2 + 3.0 * zy4) / (8.0 * (sigmac * sigmac));
    const double axn4 = (3.0 * zx4 + 6.0 * zx2 * zy2 + 3.0 * zy4) / (8.0 * (sigmau * sigmau));
    const double axe2xn2 = (zx4 + 10.0 * zx2 * zy2 + zy4) / (8.0 * sigmau * sigmac);
    double coef = 1.0;
    coef = coef + c40 / 24.0 * (axe4 - 6.0 * axe2 + 3.0);
    coef = coef + c04 / 24.0 * (axn4 - 6.0 * axn2 + 3.0);
    coef = coef + c22 / 4.0 * (axe2xn2 - axn2 - axe2 + 1.0);
    coef = coef / (2.0 * pi * sqrt(sigmau) * sqrt(sigmac));
    const double proba = coef * exp(-(axe2 + axn2) / 2.0);
    double cos2chi = cv * cs + sv * ss * cos(pi - phi);
    if (cos2chi > 1.0) cos2chi = SBD_F32(0.99999999999);
    if (cos2chi < -1.0) cos2chi = -SBD_F32(0.99999999999);
    const double coschi = sqrt(0.5 * (1.0 + cos2chi));
    const double sinchi = sqrt(0.5 * (1.0 - cos2chi));
    const double r1 = surf_fresnel(nr, ni, coschi, sinchi);
    const double ct = cos(tilt);
    return pi * r1 * proba / (4.0 * cs * cv * ((ct * ct) * (ct * ct)));
}

// Hapke (hapkbdrf, spectra.f:298-348): ui incidence, ur reflection
__host__ __device__ inline double surf_hapke(const double *bp, double ui, double ur, double phir)
{
    const double pi = SBD_PI_PARAMS, hssa = bp[0], hasym = bp[1], hotspt = bp[2], hotwdth = bp[3];
    const double coss = ui * ur + sqrt(1.0 - ur * ur) * sqrt(1.0 - ui * ui) * cos(pi - phir);
    const double s = acos(coss);
    const double pfun = (1.0 - hasym * hasym) / pow(1.0 + hasym * hasym + 2.0 * hasym * coss, 1.5);
    const double pfun0 = (1.0 - hasym * hasym) / ((1.0 + hasym) * (1.0 + hasym) * (1.0 + hasym));
    const double b0 = hotspt / (hssa * pfun0);
    const double bfun = b0 / (1.0 + tan(s / 2.0) / hotwdth);
    const double hfunr = (1.0 + 2.0 * ur) / (1.0 + 2.0 * ur * sqrt(1.0 - hssa));
    const double hfuni = (1.0 + 2.0 * ui) / (1.0 + 2.0 * ui * sqrt(1.0 - hssa));
    const double bdrf = (1.0 + bfun) * pfun + hfunr * hfuni - 1.0;
    return 0.25 * hssa * bdrf / (ur + ui);
}

// Ross-thick / Li-sparse (rtlsbdrf, spectra.f:350-419)
__host__ __device__ inline double surf_rossli(const double *bp, double mui, double mur, double phir)
{
    const double pi = SBD_PI_PARAMS, rliso = bp[0], rlvol = bp[1], rlgeo = bp[2], rlhot = bp[3], rlwdth = bp[4];
    const double ui = fmax(mui, 0.01), ur = fmax(mur, 0.01);
    const double cosra = cos(pi - phir);
    double coss = ui * ur + sqrt(1.0 - ur * ur) * sqrt(1.0 - ui * ui) * cosra;
    coss = fmax(-1.0, fmin(coss, 1.0));
    const double s = acos(coss), sins = sin(s);
    double f1 = (pi / 2.0 - s) * coss + sins;
    f1 = f1 / (ui + ur) - pi / 4.0;
    const double vza = acos(ur), sza = acos(ui);
    const double tanvzap = rlwdth * tan(vza), tanszap = rlwdth * tan(sza);
    double vzap = vza, szap = sza;
    if (rlwdth != 1.0) { vzap = atan(tanvzap); szap = atan(tanszap); }
    double cossp = cos(szap) * cos(vzap) + sin(szap) * sin(vzap) * cosra;
    cossp = fmax(-1.0, fmin(cossp, 1.0));
    const double dd = tanszap * tanszap + tanvzap * tanvzap - 2.0 * tanszap * tanvzap * cosra;
    const double secsum = 1.0 / cos(szap) + 1.0 / cos(vzap);
    const double tt = tanszap * tanvzap * sin(pi - phir);
    double cost = rlhot * sqrt(dd + tt * tt);
    cost = cost / secsum;
    cost = fmax(-1.0, fmin(cost, 1.0));
    const double t = acos(cost);
    double f2 = (t - sin(t) * cost) * secsum / pi;
    f2 = f2 - 1.0 / cos(vzap) + 0.5 * (1.0 + cossp) / (cos(szap) * cos(vzap));
    return rliso + rlvol * f1 + rlgeo * f2;
}

// BDREF(WVNMLO, WVNMHI, MUR, MUI, PHIR): reflection cosine first (spectra.f:249-296)
__host__ __device__ inline double surf_bdref(const BrdfModel &M, double mur, double mui, double phir)
{
    if (M.ibdrf == 1) {
        const double rgl = surf_sunglint(M.bp[0], M.nr, M.ni, mui, mur, phir);
        return M.bp[2] + (1.0 - M.bp[1]) * rgl + (1.0 - M.bp[2]) * M.rsw;   // rfoam + (1 - wndwt) rgl + (1 - rfoam) rsw
    }
    if (M.ibdrf == 2) return surf_hapke(M.bp, mui, mur, phir);
    return surf_rossli(M.bp, mui, mur, phir);
}

// where the tables of (item slot, mode) live: one set per item (ocean) or one per run (brdf_shared)
SBD_DEVICE size_t surf_index(const Params &P, int slot, int mazim) { return (size_t)(P.brdf_shared ? 0 : slot) * P.nmode + mazim; }
// BDR(iq, jq), jq = 0..nn (0: the beam), of table set `sidx`; BEM(iq); RMU(iu, iq); EMU(iu) -- 1-based like the reference
SBD_DEVICE const double *surf_bdr(const Params &P, size_t sidx) { return P.bdr + sidx * (size_t)P.nn * (P.nn + 1); }
SBD_DEVICE const double *surf_bem(const Params &P, size_t sidx) { return P.bem + sidx * (size_t)P.nn; }
SBD_DEVICE const double *surf_rmu(const Params &P, size_t sidx) { return P.rmu + sidx * (size_t)P.numu * (P.nn + 1); }
SBD_DEVICE const double *surf_emu(const Params &P, size_t sidx) { return P.emu + sidx * (size_t)P.numu; }
#define SBD_BDR(tab, iq, jq) (tab)[((iq) - 1) * (nn + 1) + (jq)]
#define SBD_RMU(tab, iu, iq) (tab)[((iu) - 1) * (nn + 1) + (iq)]

// SOLVE0's bottom-boundary entry B(N - nn + iq) (disort.f:3434-3599) for a surface that reflects (no LYRCUT):
// Lambertian (bdrt == nullptr: BDR = ALBEDO for m = 0, nothing for m > 0) or bidirectional (tables of this mode).
// zzl / zp0l / zp1l: the particular solutions of layer ncut, [n] each; eb = EXPBEA(ncut), tc = TAUCPR(ncut).
SBD_DEVICE double surf_bottom_rhs(const int iq, const int mazim, const bool beam, const double fbeam, const double umu0,
                                  const double pi, const double albedo, const double *bdrt, const double *bemt, const int nn,
                                  const double *cwt, const double *cmu, const double *zzl, const double *zp0l,
                                  const double *zp1l, const double eb, const double tc, const double bplank)
{
#define SBD_ZL(a, i) (a)[(i) - 1]
    if (mazim > 0) {
        if (!bdrt) return -SBD_ZL(zzl, iq + nn) * eb;                       // Lambertian: no coupling for m > 0
        double sum = 0.0;
        for (int jq = 1; jq <= nn; ++jq)
            sum = sum + cwt[jq - 1] * cmu[jq - 1] * SBD_BDR(bdrt, iq, jq) * SBD_ZL(zzl, nn + 1 - jq) * eb;
        double v = sum;
        if (beam) v = sum + (SBD_BDR(bdrt, iq, 0) * umu0 * fbeam / pi - SBD_ZL(zzl, iq + nn)) * eb;
        return v;
    }
    const double bem = bdrt ? bemt[iq - 1] : 1.0 - albedo;
    double sum = 0.0;
    if (beam) {
        for (int jq = 1; jq <= nn; ++jq) {
            const double bdr = bdrt ? SBD_BDR(bdrt, iq, jq) : albedo;
            sum = sum + cwt[jq - 1] * cmu[jq - 1] * bdr *
                            (SBD_ZL(zzl, nn + 1 - jq) * eb + SBD_ZL(zp0l, nn + 1 - jq) + SBD_ZL(zp1l, nn + 1 - jq) * tc);
        }
        const double bdr0 = bdrt ? SBD_BDR(bdrt, iq, 0) : albedo;
        return 2.0 * sum + (bdr0 * umu0 * fbeam / pi - SBD_ZL(zzl, iq + nn)) * eb
               + bem * bplank - SBD_ZL(zp0l, iq + nn) - SBD_ZL(zp1l, iq + nn) * tc;
    }
    for (int jq = 1; jq <= nn; ++jq) {
        const double bdr = bdrt ? SBD_BDR(bdrt, iq, jq) : albedo;
        sum = sum + cwt[jq - 1] * cmu[jq - 1] * bdr * (SBD_ZL(zp0l, nn + 1 - jq) + SBD_ZL(zp1l, nn + 1 - jq) * tc);
    }
    return 2.0 * sum + bem * bplank - SBD_ZL(zp0l, iq + nn) - SBD_ZL(zp1l, iq + nn) * tc;
#undef SBD_ZL
}

// grid: (items or 1) x nmode blocks of 256 threads; LDS: 50 * (nn + numu) doubles (+ 101 * 50 for the m = 0 block's
// CHEKIN pass, done in rounds) -- sized by the launcher as surf_lds_doubles()
inline int surf_lds_doubles(int nn, int numu) { return kSurfGauss * (nn + numu > 101 ? nn + numu : 101); }

static __global__ void __launch_bounds__(256) surfac_kernel(Params P, int32_t *bad_flag)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int nn = P.nn, numu = P.numu, nmode = P.nmode;
    const int slot = P.brdf_shared ? 0 : (int)(blockIdx.x / nmode);
    const int mazim = (int)(blockIdx.x % nmode);
    const int tid = threadIdx.x;
    BrdfModel M;
    M.ibdrf = P.ibdrf;
    for (int k = 0; k < 8; ++k) M.bp[k] = P.bpar[k];
    M.nr = M.ni = M.rsw = 0.0;
    double fbeam = 1.0;                       // (shared tables: the beam column is always filled; it only ever meets FBEAM > 0)
    int32_t *svi = nullptr;
    if (!P.brdf_shared) {
        svi = P.svi + (size_t)slot * P.svi_stride;
        const int st0 = svi[SBD_SVI_STATUS];
        fbeam = P.fbeam[slot];
        if (st0 & (0x20 | 0x10)) return;
        if (mazim > svi[SBD_SVI_NAZ]) return;
        M.nr = P.bitem[(size_t)slot * 4 + 0];
        M.ni = P.bitem[(size_t)slot * 4 + 1];
        M.rsw = P.bitem[(size_t)slot * 4 + 2];
    }
    const double pi = P.pi;
    const double *gmu = P.t.gmu50, *gwt = P.t.gwt50, *cmu = P.t.cmu, *umu = P.t.umu;
    const double umu0 = P.umu0;
    const size_t sidx = (size_t)slot * nmode + mazim;
    double *bdr = P.bdr + sidx * (size_t)nn * (nn + 1), *bem = P.bem + sidx * (size_t)nn;
    double *rmu = P.rmu + sidx * (size_t)numu * (nn + 1), *emu = P.emu + sidx * (size_t)numu;
    const double fac = 0.5 * (2.0 - ((mazim == 0) ? 1.0 : 0.0));
    // ---- BDR and RMU: one entry per thread, 50 azimuths in order (disort.f:3765-3792, 3853-3878) ----
    const int e1 = nn * (nn + 1), e2 = numu * (nn + 1);
    for (int e = tid; e < e1 + e2; e += blockDim.x) {
        const bool user = e >= e1;
        const int ee = user ? e - e1 : e;
        const int row = ee / (nn + 1), jq = ee % (nn + 1);
        const double mur = user ? umu[row] : cmu[row];
        const double mui = (jq == 0) ? umu0 : cmu[jq - 1];
        double val = 0.0;
        if (!(user && !(mur > 0.0)) && !(jq == 0 && !(fbeam > 0.0))) {
            double sum = 0.0;
            for (int k = 0; k < kSurfGauss; ++k)
                sum = sum + gwt[k] * surf_bdref(M, mur, mui, pi * gmu[k]) * cos((double)mazim * pi * gmu[k]);
            val = fac * sum;
        }
        (user ? rmu : bdr)[ee] = val;
    }
    if (mazim != 0) return;
    // ---- m = 0: directional emissivities (disort.f:3795-3818, 3880-3903): inner sums over the incidence cosine ----
    const int ntarget = nn + numu;
    for (int t = tid; t < ntarget * kSurfGauss; t += blockDim.x) {
        const int tg = t / kSurfGauss, jg = t % kSurfGauss;
        const double mur = (tg < nn) ? cmu[tg] : umu[tg - nn];
        double sum = 0.0;
        if (mur > 0.0)
            for (int k = 0; k < kSurfGauss / 2; ++k) sum = sum + gwt[k] * gmu[k] * surf_bdref(M, mur, gmu[k], pi * gmu[jg]);
        smem[t] = sum;
    }
    __syncthreads();
    for (int tg = tid; tg < ntarget; tg += blockDim.x) {
        double d = 0.0;
        for (int jg = 0; jg < kSurfGauss; ++jg) d = d + gwt[jg] * smem[tg * kSurfGauss + jg];
        if (tg < nn) bem[tg] = 1.0 - d;
        else emu[tg - nn] = (umu[tg - nn] > 0.0) ? 1.0 - d : 0.0;
    }
    __syncthreads();
    // ---- CHEKIN: the flux albedo DREF(mu) at mu = 0, 0.01 .. 1 must lie in [0,1] (disort.f:5080-5096, 5178-5284) ----
    for (int t = tid; t < 101 * kSurfGauss; t += blockDim.x) {
        const int im = t / kSurfGauss, jg = t % kSurfGauss;
        const double mu = (double)((float)im * 0.01f);           // RMU = IRMU*0.01 in REAL*4
        double sum = 0.0;
        for (int k = 0; k < kSurfGauss / 2; ++k) sum = sum + gwt[k] * gmu[k] * surf_bdref(M, gmu[k], mu, pi * gmu[jg]);
        smem[t] = sum;
    }
    __syncthreads();
    int bad = 0;
    for (int im = tid; im < 101; im += blockDim.x) {
        double d = 0.0;
        for (int jg = 0; jg < kSurfGauss; ++jg) d = d + gwt[jg] * smem[im * kSurfGauss + jg];
        if (d < 0.0 || d > 1.0) bad = 1;
    }
    if (bad) {
        if (svi) atomicOr(&svi[SBD_SVI_STATUS], 0x20);
        if (bad_flag) atomicOr(bad_flag, 1);
    }
}

}  // namespace sbd
