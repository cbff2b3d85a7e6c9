// TERPEV on the matrix cores (round 6): the eigenvectors interpolated to the user angles (disort.f:3920-4020) for the
// radiance runs of NSTR 32 -- GU(iu, iq, lc) = 1/2 sum_l GL(l) Ylm(mu_u) sum_jq CWT(jq) Ylm(+-mu_jq) EVECC(jq, iq) -- as two
// chained fp64 matrix products per layer and mode instead of 2 x (32 x 16 + 20 x 32) multiply-adds per lane inside the
// layer kernel (28 % of its ticks at NSTR 32, profiles/r06_layer_phases.txt).  The step is bound by the vector ALUs'
// ISSUE slots; v_mfma_f64_16x16x4_f64 has the vector peak, not more, but retires 1 024 multiply-adds per
// issue slot on a pipe of its own.
//
// With E11 / E21 the two independent quarters of the eigenvector matrix (Params::gcc: what the layer kernels leave for
// the band LU), S = E11 + E21, D = E11 - E21 and Y(l, -mu) = (-1)^(l-m) Y(l, mu):
//     even l - m:  T(l, j) = sum_jq w_jq Y(l, jq) S(jq, j)       odd:  T(l, j) = sum_jq w_jq Y(l, jq) D(jq, j)
//     P_e(iu, j) = sum_{l even} [1/2 GL(l) Yu(iu, l)] T(l, j)     P_o likewise over the odd l
//     GU(iu, column j + nn) = P_e + P_o   (k_j > 0)               GU(iu, column nn - 1 - j) = P_o - P_e   (k_j < 0)
// -- the same terms as the layer kernel's, summed in another order (the products of four l at a time inside the matrix
// instruction): agreement to rounding, the parity gates are the same.
//
// One wave per (work item, azimuth mode) walks the item's layers: the quadrature- and user-angle Ylm operands of the mode
// stay in registers, per layer the quarters are read in the B-operand layout (lane: row 4 s + lane / 16, column lane % 16),
// T leaves the first product in exactly the layout the second one reads (C/D row = lane / 16 + 4 reg: register r IS the
// B operand of k-step r).  Layers the fast layer kernel handed to the reference-algorithm kernel are overwritten by that
// kernel afterwards (it runs behind this one and interpolates its own eigenvectors).
#pragma once
#include "sbd_common.hpp"

namespace sbd {

typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NN>
__global__ void __launch_bounds__(64) terpev_mfma_kernel(Params P)
{
    constexpr int nn = NN, n = 2 * NN;
    static_assert(NN == 16, "terpev_mfma_kernel: one 16 x 16 tile per quarter");
    const int lane = threadIdx.x, j = lane & 15, q = lane >> 4;
    const int nmode = P.nmode, L = P.L, numu = P.numu;
    // (blocks in mode-major order, as the band kernels')
    const int mazim = (int)(blockIdx.x / (unsigned)P.nslot);
    const int slot = (int)(blockIdx.x % (unsigned)P.nslot);
    if (mazim >= nmode) return;
    const int32_t *svi = P.svi + (size_t)slot * P.svi_stride;
    const int st0 = svi[SBD_SVI_STATUS];
    if (st0 & (0x20 | 0x10)) return;                      // (the layer kernel's exits)
    if (mazim > svi[SBD_SVI_NAZ]) return;
    const int ncut = svi[SBD_SVI_NCUT];
    const long long ms = (long long)slot * nmode + mazim;
    const SV o(L);
    const double *sv = P.sv + (size_t)slot * P.sv_stride;
    const double *ylmc = P.t.ylmc + (size_t)mazim * n * (n + 1);
    const double *ylmu = P.t.ylmu + (size_t)mazim * numu * (n + 1);
    const int nmom = P.nmom;

    // ---- the mode's operands (the same for every layer) ----
    // first product, A: row i = lane % 16 <-> l = m + par + 2 i, k = 4 s + q <-> jq
    double yw[2][4];
    // second product, A: row i <-> iu = 16 tile + i, k = 4 r + q <-> l = m + par + 2 (4 r + q)
    double yu[2][2][4];
    int l2[2][4];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int l1 = mazim + par + 2 * j;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int jq = 4 * s + q;
            yw[par][s] = (l1 < n) ? ylmc[jq * (n + 1) + l1] * P.t.cwt[jq] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l = mazim + par + 2 * (4 * r + q);
            l2[par][r] = l;
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
                const int iu = 16 * tile + j;
                yu[tile][par][r] = (l < n && iu < numu) ? ylmu[iu * (n + 1) + l] : 0.0;
            }
        }
    }
    const bool two_tiles = numu > 16;

    for (int lc = 1; lc <= ncut; ++lc) {
        const size_t lidx = (size_t)ms * L + (lc - 1);
        // ---- this layer's loads, in one batch ----
        const double *cc = P.gcc + lidx * 2 * nn * nn;
        double e11[4], e21[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int jq = 4 * s + q;
            e11[s] = cc[jq * nn + j];
            e21[s] = cc[nn * nn + jq * nn + j];
        }
        const double *pm = P.pmom + (pmom_item(P, slot) * L + (lc - 1)) * (nmom + 1);
        double pk[2][4];
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int l = l2[par][r];
                pk[par][r] = (l == 0) ? 1.0 : ((l < n && l <= nmom) ? pm[l] : 0.0);
            }
        const double oprim = sv[o.oprim() + lc - 1], f = sv[o.flyr() + lc - 1];
        const double hs = 0.5 * oprim / (1.0 - f);          // 1/2 GL(l) = hs (2 l + 1) (PMOM(l) - F)  (disort.f:2583-2585)

        // ---- T = Yw S (even l - m), Yw D (odd) ----
        v4f64 te = {0.0, 0.0, 0.0, 0.0}, to = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            te = __builtin_amdgcn_mfma_f64_16x16x4f64(yw[0][s], e11[s] + e21[s], te, 0, 0, 0);
            to = __builtin_amdgcn_mfma_f64_16x16x4f64(yw[1][s], e11[s] - e21[s], to, 0, 0, 0);
        }
        // ---- P_e, P_o per tile of user angles; GU ----
        double *guout = P.gu + lidx * n * numu;
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
            if (tile == 1 && !two_tiles) break;
            v4f64 pe = {0.0, 0.0, 0.0, 0.0}, po = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ae = hs * (double)(2 * l2[0][r] + 1) * (pk[0][r] - f) * yu[tile][0][r];
                const double ao = hs * (double)(2 * l2[1][r] + 1) * (pk[1][r] - f) * yu[tile][1][r];
                pe = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, te[r], pe, 0, 0, 0);
                po = __builtin_amdgcn_mfma_f64_16x16x4f64(ao, to[r], po, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iu = 16 * tile + q + 4 * r;        // C/D: row = lane / 16 + 4 reg, column = lane % 16
                if (iu < numu) {
                    guout[(size_t)(j + nn) * numu + iu] = pe[r] + po[r];
                    guout[(size_t)(nn - 1 - j) * numu + iu] = po[r] - pe[r];
                }
            }
        }
    }
}

}  // namespace sbd
