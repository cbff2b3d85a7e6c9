// The boundary-value system of the constants of integration of one (work item, azimuth mode) -- SETMTX + SOLVE0's
// right-hand side (disort.f:2702-2994, 3322-3637) -- as the band LU kernels that do not build their rows from GC's
// quarters see it (band_kernel, sbd_band.hpp; band_rows_kernel, sbd_bandr.hpp): pointers into the pass's workspace and
// the generators of the right-hand side and of the boundary rows.  The interface rows come matrix-ready from the layer
// kernels (ga / gb).
#pragma once
#include "sbd_common.hpp"
#include "sbd_surface.hpp"

namespace sbd {

// The boundary-value system of one (item, mode): right-hand side and matrix rows (SOLVE0 / SETMTX)
template <int NN>
struct BandSystem {
    static constexpr int n = 2 * NN, nn = NN;
    int N, ncut, mazim;
    bool lyrcut, beam, brdf, refl;
    double fbeam, albedo, delm0, umu0, pi, fisot, tplank, bplank;
    const double *gc, *kk, *ek, *zz, *zp0, *zp1, *ga, *gb, *taucpr, *expbea, *cmu, *cwt, *bdrt, *bemt;
    double *sbot;

    SBD_DEVICE double GC(int i, int j, int lc) const { return gc[((size_t)(lc - 1) * n + (i - 1)) * n + (j - 1)]; }
    SBD_DEVICE double KK(int i, int lc) const { return kk[(lc - 1) * n + (i - 1)]; }
    SBD_DEVICE double EK(int i, int lc) const { return ek[(lc - 1) * nn + (i - 1)]; }
    SBD_DEVICE double ZZ(int i, int lc) const { return zz[(lc - 1) * n + (i - 1)]; }
    SBD_DEVICE double ZP0(int i, int lc) const { return zp0[(lc - 1) * n + (i - 1)]; }
    SBD_DEVICE double ZP1(int i, int lc) const { return zp1[(lc - 1) * n + (i - 1)]; }

    SBD_DEVICE void init(const Params &P, int slot, int mazim_, long long ms, const int32_t *svi, double *sbot_)
    {
        const int L = P.L;
        mazim = mazim_;
        ncut = svi[SBD_SVI_NCUT];
        lyrcut = svi[SBD_SVI_LYRCUT] != 0;
        N = ncut * n;
        const SV o(L);
        const double *sv = P.sv + (size_t)slot * P.sv_stride;
        taucpr = sv + o.taucpr();
        expbea = sv + o.expbea();
        bplank = sv[o.bplank()];
        tplank = sv[o.tplank()];
        fbeam = P.fbeam[slot];
        beam = fbeam > 0.0;
        albedo = P.albedo[slot];
        delm0 = (mazim == 0) ? 1.0 : 0.0;
        umu0 = P.umu0; pi = P.pi; fisot = P.fisot;
        cmu = P.t.cmu; cwt = P.t.cwt;
        gc = P.gc + (size_t)ms * L * n * n;
        kk = P.kk + (size_t)ms * L * n;
        ek = P.ek + (size_t)ms * L * nn;
        zz = P.zz + (size_t)ms * L * n;
        zp0 = P.zp0 + (size_t)(ms - mazim) * L * n;   // thermal particular solutions exist for mode 0 only
        zp1 = P.zp1 + (size_t)(ms - mazim) * L * n;
        ga = P.ga + (size_t)ms * L * n * n;
        gb = P.gb + (size_t)ms * L * n * n + (size_t)n * n;    // block of layer lc+1
        // the surface: Lambertian (couples only for m = 0, disort.f:2925) or bidirectional (SURFAC's tables of this mode)
        brdf = P.ibdrf != 0;
        const size_t sidx = surf_index(P, slot, mazim);
        bdrt = brdf ? surf_bdr(P, sidx) : nullptr;
        bemt = brdf ? surf_bem(P, sidx) : nullptr;
        refl = !lyrcut && (brdf || delm0 != 0.0);
        sbot = sbot_;
    }
    // bottom-boundary reflection sums: S(IQ) = sum_k CWT(k) CMU(k) BDR GC(nn+1-k, IQ, ncut), Lambertian BDR = albedo
    // for every pair (SURFAC, disort.f:3746-3763); lane < n
    SBD_DEVICE void fill_sbot(int lane) const
    {
        if (lane < n) {
            double s = 0.0;
            if (refl && !brdf)
#pragma nounroll
                for (int k = 1; k <= nn; ++k) s = s + cwt[k - 1] * cmu[k - 1] * albedo * GC(nn + 1 - k, lane + 1, ncut);
            sbot[lane] = s;
        }
    }
    // right-hand side B (SOLVE0, disort.f:3434-3599), unknown index = (lc-1)*n + iq
    SBD_DEVICE double rhs(int it) const
    {
        double v;
        if (it <= nn) {   // top boundary
            const int iq = it;
            if (mazim == 0) {
                if (beam) v = -ZZ(nn + 1 - iq, 1) - ZP0(nn + 1 - iq, 1) + fisot + tplank;
                else v = -ZP0(nn + 1 - iq, 1) + fisot + tplank;
            } else {
                v = -ZZ(nn + 1 - iq, 1);
            }
        } else if (it > N - nn) {   // bottom boundary
            const int iq = it - (N - nn);
            if (lyrcut) {                                  // nothing comes back from below the cut (disort.f:3441-3452)
                if (mazim > 0) v = -ZZ(iq + nn, ncut) * expbea[ncut];
                else if (beam) v = -ZZ(iq + nn, ncut) * expbea[ncut] - ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
                else v = -ZP0(iq + nn, ncut) - ZP1(iq + nn, ncut) * taucpr[ncut];
            } else {
                v = surf_bottom_rhs(iq, mazim, beam, fbeam, umu0, pi, albedo, bdrt, bemt, nn, cwt, cmu,
                                    zz + (ncut - 1) * n, zp0 + (ncut - 1) * n, zp1 + (ncut - 1) * n,
                                    expbea[ncut], taucpr[ncut], bplank);
            }
        } else {   // interface lc | lc+1
            const int q = it - nn - 1;
            const int lc = q / n + 1, iq = q % n + 1;
            if (mazim > 0) {
                v = (ZZ(iq, lc + 1) - ZZ(iq, lc)) * expbea[lc];
            } else if (beam) {
                v = (ZZ(iq, lc + 1) - ZZ(iq, lc)) * expbea[lc] + ZP0(iq, lc + 1) - ZP0(iq, lc)
                    + (ZP1(iq, lc + 1) - ZP1(iq, lc)) * taucpr[lc];
            } else {
                v = ZP0(iq, lc + 1) - ZP0(iq, lc) + (ZP1(iq, lc + 1) - ZP1(iq, lc)) * taucpr[lc];
            }
        }
        return v;
    }
    // element (r, col) of any row: interface rows from the matrix-ready blocks (unit stride), boundary rows generated
    SBD_DEVICE double row_elem(int r, int col) const
    {
        if (col > N) return 0.0;
        if (r > nn && r <= N - nn) {
            const int qq = r - nn - 1;                   // row jq = qq % n of interface lc = qq / n + 1
            const int d = col - (qq / n) * n;            // 1..2n inside the row's support
            if (d >= 1 && d <= n) return ga[(size_t)qq * n + d - 1];
            if (d > n && d <= 2 * n) return gb[(size_t)qq * n + d - n - 1];
            return 0.0;
        }
        return boundary_elem(r, col);
    }
    // element (r, col) of a boundary row (SETMTX, disort.f:2844-2990): a GC element times its STWJ factor
    SBD_DEVICE double boundary_elem(int r, int col) const
    {
        if (col < 1 || col > N) return 0.0;
        double g = 0.0, f = 1.0;
        if (r <= nn) {                       // top boundary: GC(nn+1-r, j, 1) * exp(KK(j,1)*TAUCPR(1))
            if (col <= n) {
                g = GC(nn + 1 - r, col, 1);
                if (col <= nn) f = exp(KK(col, 1) * taucpr[1]);
            }
        } else {                             // bottom boundary, the surface's reflection folded in
            const int iq = col - (N - n);
            if (iq >= 1) {
                g = GC(nn + (r - (N - nn)), iq, ncut);
                if (refl && brdf) {                        // row r - (N - nn) of BDR meets the downward streams (disort.f:2946-2952)
                    double sr = 0.0;
#pragma nounroll
                    for (int k = 1; k <= nn; ++k)
                        sr = sr + cwt[k - 1] * cmu[k - 1] * SBD_BDR(bdrt, r - (N - nn), k) * GC(nn + 1 - k, iq, ncut);
                    g = g - (1.0 + delm0) * sr;
                } else if (refl) g = g - (1.0 + delm0) * sbot[iq - 1];
                if (iq > nn) f = EK(n + 1 - iq, ncut);
            }
        }
        return g * f;
    }
};

}  // namespace sbd
