// The reference's OWN formulation of the boundary-value system of one (work item, azimuth mode), for ONE purpose: LINPACK's
// reciprocal condition estimate of it, on which the reference raises errmsg 2 ("SOLVE0--SGBCO says matrix near singular",
// disort.f:3607-3610: 1 + RCOND == 1).
//
// The band kernels of this engine (sbd_band4.hpp, sbd_band1.hpp, sbd_bandr.hpp) solve an equivalent system -- the layers'
// eigenvector columns in Jacobi's order and normalisation, rows and columns arranged by layer block, no L factor kept --
// and SGBCO's estimate is invariant under none of that: it is a property of the reference's matrix as the reference
// builds it.  RCOND falls below eps only next to layers a few ulps from conservative scattering, whose smallest
// eigenvalue is rounding noise of the reduced eigenproblem (k ~ 1e-8): ASYMTX's unnormalised eigenvector, divided by that
// k (disort.f:3273-3286), scales two columns of the layer by 1e8 against the neighbours'.  To say what the reference says
// there, one has to compute what the reference computes: this file restates, statement for statement and with one rounding
// per operation (no fused multiply-add: the reference's x86-64 object code has none),
//
//   SETDIS's delta-M scaling of a layer           disort.f:2570-2592
//   SOLEIG (CC, AMB, APB, ARRAY, eigenvectors)    disort.f:3197-3314
//   ASYMTX (balance, Hessenberg, double QR)       disort.f:873-1656
//   SETMTX (LINPACK band storage)                 disort.f:2702-2994
//   SGBFA / SGBCO with ISAMAX, SSCAL, SAXPY,      disutil.f:426-918, 1611-2102
//     SASUM, SDOT (left-to-right sums)
//
// as plain serial code that compiles for the host and for the device.  On the device one wave serves one flagged system
// (band_rcond_kernel, sbd_k_refband.hip): a lane per layer for the eigenproblems, then the band matrix.  On the host the
// same source stands behind sbd_band_rcond_host (no GPU): with the host's exp it returns the oracle's RCOND bit for bit
// (tests/test_refband_host.py) -- the pin of the kernel's source, like sbd_gas.hpp's.  Only systems the cheap filter lists
// come here (an item with a layer within 1e-12 of conservative scattering -- setup_kernel's mark, sbd_setup.hpp -- or a pivot
// ratio <= 1e-10 in the band LU): none on the headline sweep.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SBD_RB __host__ __device__ inline
#else
#define SBD_RB inline
#endif

namespace sbd {
namespace refband {

SBD_RB double dsign(double a, double b) { return (b >= 0.0) ? fabs(a) : -fabs(a); }
SBD_RB int imin(int a, int b) { return a < b ? a : b; }
SBD_RB int imax(int a, int b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------------------------------
// ASYMTX (disort.f:873-1656): eigenvalues and (unnormalised) eigenvectors of a real M x M matrix with real spectrum.
// aa (leading dimension ia) is destroyed; evec has leading dimension ievec; wk holds 2 M doubles.
// Returns IER: 0, the index of an eigenvalue that did not converge in 30 iterations, or -1 (complex pair, M = 2).
// ---------------------------------------------------------------------------------------------------------------------
SBD_RB int asymtx(double *aa, double *evec, double *eval, int m, int ia, int ievec, double *wk)
{
#pragma clang fp contract(off)
#define AA(i, j) aa[((size_t)(j) - 1) * ia + ((i) - 1)]
#define EV(i, j) evec[((size_t)(j) - 1) * ievec + ((i) - 1)]
#define WK(i) wk[(i) - 1]
    const double c1 = 0.4375, c2 = 0.5, c3 = 0.75, c4 = 0.95, c5 = 16.0, c6 = 256.0;
    const double tol = 2.220446049250313e-16;       // R1MACH(4) of the reference's double-precision build
    double p = 0.0, q = 0.0, r = 0.0;
    double col, discri, f, g, h, repl, rnorm, row, s, scale, sgn, t, uu, vv, w, x, y, z;
    int i, in, j, k, ka, kkk, l, lb = 0, lll, n, n1, n2;

    if (m == 1) { eval[0] = AA(1, 1); EV(1, 1) = 1.0; return 0; }
    if (m == 2) {                                   // closed form, disort.f:989-1023
        discri = (AA(1, 1) - AA(2, 2)) * (AA(1, 1) - AA(2, 2)) + 4.0 * AA(1, 2) * AA(2, 1);
        if (discri < 0.0) return -1;
        sgn = 1.0;
        if (AA(1, 1) < AA(2, 2)) sgn = -1.0;
        eval[0] = 0.5 * (AA(1, 1) + AA(2, 2) + sgn * sqrt(discri));
        eval[1] = 0.5 * (AA(1, 1) + AA(2, 2) - sgn * sqrt(discri));
        EV(1, 1) = 1.0;
        EV(2, 2) = 1.0;
        if (AA(1, 1) == AA(2, 2) && (AA(2, 1) == 0.0 || AA(1, 2) == 0.0)) {
            rnorm = fabs(AA(1, 1)) + fabs(AA(1, 2)) + fabs(AA(2, 1)) + fabs(AA(2, 2));
            w = tol * rnorm;
            EV(2, 1) = AA(2, 1) / w;
            EV(1, 2) = -AA(1, 2) / w;
        } else {
            EV(2, 1) = AA(2, 1) / (eval[0] - AA(2, 2));
            EV(1, 2) = AA(1, 2) / (eval[1] - AA(1, 1));
        }
        return 0;
    }
    for (i = 1; i <= m; ++i) {
        eval[i - 1] = 0.0;
        for (j = 1; j <= m; ++j) EV(i, j) = 0.0;
        EV(i, i) = 1.0;
    }
    // balance: rows isolating an eigenvalue go down (disort.f:1043-1092) ...
    rnorm = 0.0;
    l = 1;
    k = m;
    for (bool again = true; again;) {
        again = false;
        kkk = k;
        for (j = kkk; j >= 1; --j) {
            row = 0.0;
            for (i = 1; i <= k; ++i)
                if (i != j) row = row + fabs(AA(j, i));
            if (row == 0.0) {
                WK(k) = (double)j;
                if (j != k) {
                    for (i = 1; i <= k; ++i) { repl = AA(i, j); AA(i, j) = AA(i, k); AA(i, k) = repl; }
                    for (i = l; i <= m; ++i) { repl = AA(j, i); AA(j, i) = AA(k, i); AA(k, i) = repl; }
                }
                k = k - 1;
                again = true;
                break;
            }
        }
    }
    // ... columns isolating one go left (disort.f:1095-1135)
    for (bool again = true; again;) {
        again = false;
        lll = l;
        for (j = lll; j <= k; ++j) {
            col = 0.0;
            for (i = l; i <= k; ++i)
                if (i != j) col = col + fabs(AA(i, j));
            if (col == 0.0) {
                WK(l) = (double)j;
                if (j != l) {
                    for (i = 1; i <= k; ++i) { repl = AA(i, j); AA(i, j) = AA(i, l); AA(i, l) = repl; }
                    for (i = l; i <= m; ++i) { repl = AA(j, i); AA(j, i) = AA(l, i); AA(l, i) = repl; }
                }
                l = l + 1;
                again = true;
                break;
            }
        }
    }
    // balance the sub-matrix in rows l..k with powers of 16 (disort.f:1138-1188)
    for (i = l; i <= k; ++i) WK(i) = 1.0;
    for (;;) {
        bool noconv = false;
        for (i = l; i <= k; ++i) {
            col = 0.0;
            row = 0.0;
            for (j = l; j <= k; ++j)
                if (j != i) { col = col + fabs(AA(j, i)); row = row + fabs(AA(i, j)); }
            f = 1.0;
            g = row / c5;
            h = col + row;
            while (col < g) { f = f * c5; col = col * c6; }
            g = row * c5;
            while (col >= g) { f = f / c5; col = col / c6; }
            if ((col + row) / f < c4 * h) {
                WK(i) = WK(i) * f;
                noconv = true;
                for (j = l; j <= m; ++j) AA(i, j) = AA(i, j) / f;
                for (j = 1; j <= k; ++j) AA(j, i) = AA(j, i) * f;
            }
        }
        if (!noconv) break;
    }
    // Householder reduction to upper Hessenberg form, transformations accumulated (disort.f:1191-1286)
    if (!(k - 1 < l + 1)) {
        for (n = l + 1; n <= k - 1; ++n) {
            h = 0.0;
            WK(n + m) = 0.0;
            scale = 0.0;
            for (i = n; i <= k; ++i) scale = scale + fabs(AA(i, n - 1));
            if (scale != 0.0) {
                for (i = k; i >= n; --i) {
                    WK(i + m) = AA(i, n - 1) / scale;
                    h = h + WK(i + m) * WK(i + m);
                }
                g = -dsign(sqrt(h), WK(n + m));
                h = h - WK(n + m) * g;
                WK(n + m) = WK(n + m) - g;
                for (j = n; j <= m; ++j) {
                    f = 0.0;
                    for (i = k; i >= n; --i) f = f + WK(i + m) * AA(i, j);
                    for (i = n; i <= k; ++i) AA(i, j) = AA(i, j) - WK(i + m) * f / h;
                }
                for (i = 1; i <= k; ++i) {
                    f = 0.0;
                    for (j = k; j >= n; --j) f = f + WK(j + m) * AA(i, j);
                    for (j = n; j <= k; ++j) AA(i, j) = AA(i, j) - WK(j + m) * f / h;
                }
                WK(n + m) = scale * WK(n + m);
                AA(n, n - 1) = scale * g;
            }
        }
        for (n = k - 2; n >= l; --n) {
            f = AA(n + 1, n);
            if (f != 0.0) {
                f = f * WK(n + 1 + m);
                for (i = n + 2; i <= k; ++i) WK(i + m) = AA(i, n);
                if (n + 1 <= k) {
                    for (j = 1; j <= m; ++j) {
                        g = 0.0;
                        for (i = n + 1; i <= k; ++i) g = g + WK(i + m) * EV(i, j);
                        g = g / f;
                        for (i = n + 1; i <= k; ++i) EV(i, j) = EV(i, j) + g * WK(i + m);
                    }
                }
            }
        }
    }
    // norm of the Hessenberg matrix, isolated eigenvalues (disort.f:1289-1300)
    n = 1;
    for (i = 1; i <= m; ++i) {
        for (j = n; j <= m; ++j) rnorm = rnorm + fabs(AA(i, j));
        n = i;
        if (i < l || i > k) eval[i - 1] = AA(i, i);
    }
    n = k;
    t = 0.0;
    // the eigenvalues, last to first (disort.f:1305-1546)
    while (n >= l) {
        in = 0;
        n1 = n - 1;
        n2 = n - 2;
        for (;;) {                                  // iterations for the eigenvalue(s) at the bottom of rows l..n
            for (i = l; i <= n; ++i) {
                lb = n + l - i;
                if (lb == l) break;
                s = fabs(AA(lb - 1, lb - 1)) + fabs(AA(lb, lb));
                if (s == 0.0) s = rnorm;
                if (fabs(AA(lb, lb - 1)) <= tol * s) break;
            }
            x = AA(n, n);
            if (lb == n) {                          // one eigenvalue
                AA(n, n) = x + t;
                eval[n - 1] = AA(n, n);
                n = n1;
                break;
            }
            y = AA(n1, n1);
            w = AA(n, n1) * AA(n1, n);
            if (lb == n1) {                         // two eigenvalues
                p = (y - x) * c2;
                q = p * p + w;
                z = sqrt(fabs(q));
                AA(n, n) = x + t;
                x = AA(n, n);
                AA(n1, n1) = y + t;
                z = p + dsign(z, p);
                eval[n1 - 1] = x + z;
                eval[n - 1] = eval[n1 - 1];
                if (z != 0.0) eval[n - 1] = x - w / z;
                x = AA(n, n1);
                r = sqrt(x * x + z * z);
                p = x / r;
                q = z / r;
                for (j = n1; j <= m; ++j) {
                    z = AA(n1, j);
                    AA(n1, j) = q * z + p * AA(n, j);
                    AA(n, j) = q * AA(n, j) - p * z;
                }
                for (i = 1; i <= n; ++i) {
                    z = AA(i, n1);
                    AA(i, n1) = q * z + p * AA(i, n);
                    AA(i, n) = q * AA(i, n) - p * z;
                }
                for (i = l; i <= k; ++i) {
                    z = EV(i, n1);
                    EV(i, n1) = q * z + p * EV(i, n);
                    EV(i, n) = q * EV(i, n) - p * z;
                }
                n = n2;
                break;
            }
            if (in == 30) return n;                 // no convergence
            if (in == 10 || in == 20) {             // exceptional shift
                t = t + x;
                for (i = l; i <= n; ++i) AA(i, i) = AA(i, i) - x;
                s = fabs(AA(n, n1)) + fabs(AA(n1, n2));
                x = c3 * s;
                y = x;
                w = -c1 * (s * s);
            }
            in = in + 1;
            for (j = lb; j <= n2; ++j) {            // two consecutive small sub-diagonal elements
                i = n2 + lb - j;
                z = AA(i, i);
                r = x - z;
                s = y - z;
                p = (r * s - w) / AA(i + 1, i) + AA(i, i + 1);
                q = AA(i + 1, i + 1) - z - r - s;
                r = AA(i + 2, i + 1);
                s = fabs(p) + fabs(q) + fabs(r);
                p = p / s;
                q = q / s;
                r = r / s;
                if (i == lb) break;
                uu = fabs(AA(i, i - 1)) * (fabs(q) + fabs(r));
                vv = fabs(p) * (fabs(AA(i - 1, i - 1)) + fabs(z) + fabs(AA(i + 1, i + 1)));
                if (uu <= tol * vv) break;
            }
            AA(i + 2, i) = 0.0;
            for (j = i + 3; j <= n; ++j) { AA(j, j - 2) = 0.0; AA(j, j - 3) = 0.0; }
            for (ka = i; ka <= n1; ++ka) {          // double QR step on rows i..n
                const bool notlas = (ka != n1);
                if (ka == i) {
                    s = dsign(sqrt(p * p + q * q + r * r), p);
                    if (lb != i) AA(ka, ka - 1) = -AA(ka, ka - 1);
                } else {
                    p = AA(ka, ka - 1);
                    q = AA(ka + 1, ka - 1);
                    r = 0.0;
                    if (notlas) r = AA(ka + 2, ka - 1);
                    x = fabs(p) + fabs(q) + fabs(r);
                    if (x == 0.0) continue;
                    p = p / x;
                    q = q / x;
                    r = r / x;
                    s = dsign(sqrt(p * p + q * q + r * r), p);
                    AA(ka, ka - 1) = -s * x;
                }
                p = p + s;
                x = p / s;
                y = q / s;
                z = r / s;
                q = q / p;
                r = r / p;
                for (j = ka; j <= m; ++j) {         // rows
                    p = AA(ka, j) + q * AA(ka + 1, j);
                    if (notlas) { p = p + r * AA(ka + 2, j); AA(ka + 2, j) = AA(ka + 2, j) - p * z; }
                    AA(ka + 1, j) = AA(ka + 1, j) - p * y;
                    AA(ka, j) = AA(ka, j) - p * x;
                }
                const int itop = imin(n, ka + 3);
                for (int ii = 1; ii <= itop; ++ii) {  // columns
                    p = x * AA(ii, ka) + y * AA(ii, ka + 1);
                    if (notlas) { p = p + z * AA(ii, ka + 2); AA(ii, ka + 2) = AA(ii, ka + 2) - p * r; }
                    AA(ii, ka + 1) = AA(ii, ka + 1) - p * q;
                    AA(ii, ka) = AA(ii, ka) - p;
                }
                for (int ii = l; ii <= k; ++ii) {   // accumulate
                    p = x * EV(ii, ka) + y * EV(ii, ka + 1);
                    if (notlas) { p = p + z * EV(ii, ka + 2); EV(ii, ka + 2) = EV(ii, ka + 2) - p * r; }
                    EV(ii, ka + 1) = EV(ii, ka + 1) - p * q;
                    EV(ii, ka) = EV(ii, ka) - p;
                }
            }
        }
    }
    // eigenvectors of the triangular matrix, back-transformed, un-balanced, un-permuted (disort.f:1549-1652)
    if (rnorm != 0.0) {
        for (n = m; n >= 1; --n) {
            n2 = n;
            AA(n, n) = 1.0;
            for (i = n - 1; i >= 1; --i) {
                w = AA(i, i) - eval[n - 1];
                if (w == 0.0) w = tol * rnorm;
                r = AA(i, n);
                for (j = n2; j <= n - 1; ++j) r = r + AA(i, j) * AA(j, n);
                AA(i, n) = -r / w;
                n2 = i;
            }
        }
        for (i = 1; i <= m; ++i)
            if (i < l || i > k)
                for (j = i; j <= m; ++j) EV(i, j) = AA(i, j);
        if (k != 0) {
            for (j = m; j >= l; --j)
                for (i = l; i <= k; ++i) {
                    z = 0.0;
                    const int ntop = imin(j, k);
                    for (n = l; n <= ntop; ++n) z = z + EV(i, n) * AA(n, j);
                    EV(i, j) = z;
                }
        }
    }
    for (i = l; i <= k; ++i)
        for (j = 1; j <= m; ++j) EV(i, j) = EV(i, j) * WK(i);
    for (i = l - 1; i >= 1; --i) {
        j = (int)WK(i);
        if (i != j)
            for (n = 1; n <= m; ++n) { repl = EV(i, n); EV(i, n) = EV(j, n); EV(j, n) = repl; }
    }
    for (i = k + 1; i <= m; ++i) {
        j = (int)WK(i);
        if (i != j)
            for (n = 1; n <= m; ++n) { repl = EV(i, n); EV(i, n) = EV(j, n); EV(j, n) = repl; }
    }
    return 0;
#undef AA
#undef EV
#undef WK
}

// Scratch of ONE layer's eigenproblem (doubles): amb, apb, array nn x nn; evecc n x n; cc nn x n; gl n; eval nn; wk 2 nn
SBD_RB size_t layer_work_doubles(int n) { const size_t nn = n / 2; return 3 * nn * nn + (size_t)n * n + nn * n + n + nn + 2 * nn; }

// SETDIS's scaling of a layer + SOLEIG (disort.f:2577-2585, 3197-3314) for azimuth mode mazim: GC(.,.,lc) column-major
// (gc[(j-1) n + i-1] = GC(i,j)) and KK(.,lc) as the reference stores them.  ssalb: the layer's albedo AFTER DISORT's
// dither (disort.f:486).  pm: the layer's moments PMOM(0:nmom).  ylmc: YLMC(l, iq) = ylmc[(iq-1)(n+1) + l] of this mode,
// mirrored for iq > nn.  *dtaucp receives the scaled optical depth.  Returns ASYMTX's IER.
SBD_RB int soleig_layer(int n, int mazim, double dtauc, double ssalb, const double *pm, int nmom, const double *cmu,
                        const double *cwt, const double *ylmc, double *work, double *gc, double *kk, double *dtaucp)
{
#pragma clang fp contract(off)
    const int nn = n / 2;
    double *amb = work, *apb = amb + nn * nn, *arr = apb + nn * nn, *evecc = arr + nn * nn, *cc = evecc + n * n;
    double *gl = cc + nn * n, *eval = gl + n, *wk = eval + nn;
#define AMB(i, j) amb[((j) - 1) * nn + ((i) - 1)]
#define APB(i, j) apb[((j) - 1) * nn + ((i) - 1)]
#define ARR(i, j) arr[((j) - 1) * nn + ((i) - 1)]
#define EVECC(i, j) evecc[((j) - 1) * n + ((i) - 1)]
#define CCH(i, j) cc[((j) - 1) * nn + ((i) - 1)]
#define YLMC(l, iq) ylmc[((iq) - 1) * (n + 1) + (l)]
#define GCL(i, j) gc[((size_t)(j) - 1) * n + ((i) - 1)]
    const double f = (n <= nmom) ? pm[n] : 0.0;
    const double oprim = ssalb * (1.0 - f) / (1.0 - f * ssalb);
    *dtaucp = (1.0 - f * ssalb) * dtauc;
    for (int k = 0; k <= n - 1; ++k) {
        const double pk = (k == 0) ? 1.0 : ((k <= nmom) ? pm[k] : 0.0);
        gl[k] = (double)(2 * k + 1) * oprim * (pk - f) / (1.0 - f);
    }
    for (int iq = 1; iq <= nn; ++iq) {
        for (int jq = 1; jq <= n; ++jq) {
            double sum = 0.0;
            for (int l = mazim; l <= n - 1; ++l) sum = sum + gl[l] * YLMC(l, iq) * YLMC(l, jq);
            CCH(iq, jq) = 0.5 * sum * cwt[jq - 1];
        }
        for (int jq = 1; jq <= nn; ++jq) {
            const double alpha = CCH(iq, jq) / cmu[iq - 1];
            const double beta = CCH(iq, jq + nn) / cmu[iq - 1];
            AMB(iq, jq) = alpha - beta;
            APB(iq, jq) = alpha + beta;
        }
        AMB(iq, iq) = AMB(iq, iq) - 1.0 / cmu[iq - 1];
        APB(iq, iq) = APB(iq, iq) - 1.0 / cmu[iq - 1];
    }
    for (int iq = 1; iq <= nn; ++iq)
        for (int jq = 1; jq <= nn; ++jq) {
            double sum = 0.0;
            for (int kq = 1; kq <= nn; ++kq) sum = sum + APB(iq, kq) * AMB(kq, jq);
            ARR(iq, jq) = sum;
        }
    const int ier = asymtx(arr, evecc, eval, nn, nn, n, wk);
    if (ier != 0) return ier;
    for (int iq = 1; iq <= nn; ++iq) {
        eval[iq - 1] = sqrt(fabs(eval[iq - 1]));
        kk[iq + nn - 1] = eval[iq - 1];
        kk[nn + 1 - iq - 1] = -eval[iq - 1];
    }
    for (int jq = 1; jq <= nn; ++jq)
        for (int iq = 1; iq <= nn; ++iq) {
            double sum = 0.0;
            for (int kq = 1; kq <= nn; ++kq) sum = sum + AMB(iq, kq) * EVECC(kq, jq);
            APB(iq, jq) = sum / eval[jq - 1];
        }
    for (int jq = 1; jq <= nn; ++jq)
        for (int iq = 1; iq <= nn; ++iq) {
            double gpplgm = APB(iq, jq);
            const double gpmigm = EVECC(iq, jq);
            const double e11 = 0.5 * (gpplgm + gpmigm), e21 = 0.5 * (gpplgm - gpmigm);
            gpplgm = -gpplgm;
            const double e12 = 0.5 * (gpplgm + gpmigm), e22 = 0.5 * (gpplgm - gpmigm);
            GCL(iq + nn, jq + nn) = e11;
            GCL(nn + 1 - iq, jq + nn) = e21;
            GCL(iq + nn, nn + 1 - jq) = e12;
            GCL(nn + 1 - iq, nn + 1 - jq) = e22;
        }
    return 0;
#undef AMB
#undef APB
#undef ARR
#undef EVECC
#undef CCH
#undef YLMC
#undef GCL
}

// SETMTX (disort.f:2702-2994): the coefficient matrix in LINPACK's band storage CBAND(lda, n ncut), lda = 9 nn - 2.
// gc / kk: [layer][..] as soleig_layer leaves them; dtaucp [L], taucpr [L+1].  The surface: Lambertian with `albedo` when
// bdr == NULL (BDR = ALBEDO for m = 0, SURFAC disort.f:3746-3763), else SURFAC's table of this mode, BDR(iq, jq) =
// bdr[(iq-1)(nn+1) + jq], jq = 0..nn (sbd_surface.hpp).  wk: nn doubles.  exp() is the caller's (host: libm, device: the device library).
SBD_RB void setmtx(int n, int ncut, bool lyrcut, bool lamber, double delm0, double albedo, const double *bdr, const double *cmu,
                   const double *cwt, const double *gc, const double *kk, const double *dtaucp, const double *taucpr,
                   double *cband, int lda, double *wk)
{
#pragma clang fp contract(off)
    const int nn = n / 2;
#define CB(i, j) cband[((size_t)(j) - 1) * lda + ((i) - 1)]
#define GC(i, j, lc) gc[(((size_t)(lc) - 1) * n + ((j) - 1)) * n + ((i) - 1)]
#define KK(i, lc) kk[((size_t)(lc) - 1) * n + ((i) - 1)]
#define BDRV(iq, jq) (bdr ? bdr[((size_t)(iq) - 1) * (nn + 1) + (jq)] : albedo)
    const size_t total = (size_t)lda * n * ncut;
    for (size_t e = 0; e < total; ++e) cband[e] = 0.0;
    const int ncd = 3 * nn - 1;
    const int nshift = (3 * ncd + 1) - 2 * n + 1;
    int ncol = 0, jcol, irow;
    for (int lc = 1; lc <= ncut; ++lc) {
        for (int iq = 1; iq <= nn; ++iq) wk[iq - 1] = exp(KK(iq, lc) * dtaucp[lc - 1]);
        jcol = 0;
        for (int iq = 1; iq <= nn; ++iq) {
            ncol = ncol + 1;
            irow = nshift - jcol;
            for (int jq = 1; jq <= n; ++jq) {
                CB(irow + n, ncol) = GC(jq, iq, lc);
                CB(irow, ncol) = -GC(jq, iq, lc) * wk[iq - 1];
                irow = irow + 1;
            }
            jcol = jcol + 1;
        }
        for (int iq = nn + 1; iq <= n; ++iq) {
            ncol = ncol + 1;
            irow = nshift - jcol;
            for (int jq = 1; jq <= n; ++jq) {
                CB(irow + n, ncol) = GC(jq, iq, lc) * wk[n + 1 - iq - 1];
                CB(irow, ncol) = -GC(jq, iq, lc);
                irow = irow + 1;
            }
            jcol = jcol + 1;
        }
    }
    // top boundary
    jcol = 0;
    for (int iq = 1; iq <= nn; ++iq) {
        const double expa = exp(KK(iq, 1) * taucpr[1]);
        irow = nshift - jcol + nn;
        for (int jq = nn; jq >= 1; --jq) { CB(irow, jcol + 1) = GC(jq, iq, 1) * expa; irow = irow + 1; }
        jcol = jcol + 1;
    }
    for (int iq = nn + 1; iq <= n; ++iq) {
        irow = nshift - jcol + nn;
        for (int jq = nn; jq >= 1; --jq) { CB(irow, jcol + 1) = GC(jq, iq, 1); irow = irow + 1; }
        jcol = jcol + 1;
    }
    // bottom boundary (wk still holds exp(KK DTAUCP) of layer ncut)
    const bool plain = lyrcut || (lamber && delm0 == 0.0);
    int nncol = ncol - n;
    jcol = 0;
    for (int iq = 1; iq <= nn; ++iq) {
        nncol = nncol + 1;
        irow = nshift - jcol + n;
        for (int jq = nn + 1; jq <= n; ++jq) {
            if (plain) {
                CB(irow, nncol) = GC(jq, iq, ncut);
            } else {
                double sum = 0.0;
                for (int k = 1; k <= nn; ++k) sum = sum + cwt[k - 1] * cmu[k - 1] * BDRV(jq - nn, k) * GC(nn + 1 - k, iq, ncut);
                CB(irow, nncol) = GC(jq, iq, ncut) - (1.0 + delm0) * sum;
            }
            irow = irow + 1;
        }
        jcol = jcol + 1;
    }
    for (int iq = nn + 1; iq <= n; ++iq) {
        nncol = nncol + 1;
        irow = nshift - jcol + n;
        const double expa = wk[n + 1 - iq - 1];
        for (int jq = nn + 1; jq <= n; ++jq) {
            if (plain) {
                CB(irow, nncol) = GC(jq, iq, ncut) * expa;
            } else {
                double sum = 0.0;
                for (int k = 1; k <= nn; ++k) sum = sum + cwt[k - 1] * cmu[k - 1] * BDRV(jq - nn, k) * GC(nn + 1 - k, iq, ncut);
                CB(irow, nncol) = (GC(jq, iq, ncut) - (1.0 + delm0) * sum) * expa;
            }
            irow = irow + 1;
        }
        jcol = jcol + 1;
    }
#undef CB
#undef GC
#undef KK
#undef BDRV
}

// ---- BLAS-1 as LINPACK uses it here (unit stride).  SASUM and SDOT add left to right (the reference's unrolled loops
//      associate that way, disutil.f:1651-1666, 1812-1828); SAXPY and SSCAL are element-wise ----
SBD_RB int isamax(int n, const double *sx)              // first index of the largest |x| (strict <), 0 when none is > 0
{
    if (n <= 0) return 0;
    if (n == 1) return 1;
    double smax = 0.0;
    int idx = 0;
    for (int i = 1; i <= n; ++i) {
        const double xmag = fabs(sx[i - 1]);
        if (smax < xmag) { smax = xmag; idx = i; }
    }
    return idx;
}
SBD_RB void saxpy(int n, double sa, const double *__restrict__ sx, double *__restrict__ sy)
{
#pragma clang fp contract(off)
    if (n <= 0 || sa == 0.0) return;
    for (int i = 0; i < n; ++i) sy[i] = sy[i] + sa * sx[i];
}
SBD_RB void sscal(int n, double sa, double *sx)
{
    for (int i = 0; i < n; ++i) sx[i] = sa * sx[i];
}
SBD_RB double sasum(int n, const double *sx)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + fabs(sx[i]);
    return s;
}
SBD_RB double sdot(int n, const double *sx, const double *sy)
{
#pragma clang fp contract(off)
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + sx[i] * sy[i];
    return s;
}

// SGBFA (disutil.f:771-918): LU of the band matrix, partial pivoting by ISAMAX's rule, multipliers stored negated
SBD_RB void sgbfa(double *abd, int lda, int n, int ml, int mu, int32_t *ipvt, int *info)
{
#pragma clang fp contract(off)
#define ABD(i, j) abd[((size_t)(j) - 1) * lda + ((i) - 1)]
    const int m = ml + mu + 1;
    *info = 0;
    const int j0 = mu + 2, j1 = imin(n, m) - 1;
    for (int jz = j0; jz <= j1; ++jz) {
        const int i0 = m + 1 - jz;
        for (int i = i0; i <= ml; ++i) ABD(i, jz) = 0.0;
    }
    int jz = j1, ju = 0;
    for (int k = 1; k <= n - 1; ++k) {
        const int kp1 = k + 1;
        jz = jz + 1;
        if (jz <= n)
            for (int i = 1; i <= ml; ++i) ABD(i, jz) = 0.0;
        const int lm = imin(ml, n - k);
        int l = isamax(lm + 1, &ABD(m, k)) + m - 1;
        ipvt[k - 1] = l + k - m;
        if (ABD(l, k) == 0.0) {
            *info = k;
        } else {
            if (l != m) { const double t = ABD(l, k); ABD(l, k) = ABD(m, k); ABD(m, k) = t; }
            double t = -1.0 / ABD(m, k);
            sscal(lm, t, &ABD(m + 1, k));
            ju = imin(imax(ju, mu + ipvt[k - 1]), n);
            int mm = m;
            for (int j = kp1; j <= ju; ++j) {
                l = l - 1;
                mm = mm - 1;
                t = ABD(l, j);
                if (l != mm) { ABD(l, j) = ABD(mm, j); ABD(mm, j) = t; }
                saxpy(lm, t, &ABD(m + 1, k), &ABD(mm + 1, j));
            }
        }
    }
    ipvt[n - 1] = n;
    if (ABD(m, n) == 0.0) *info = n;
#undef ABD
}

// SGBCO (disutil.f:426-769): SGBFA + the 1-norm reciprocal condition estimate.  z: n doubles.
SBD_RB double sgbco(double *abd, int lda, int n, int ml, int mu, int32_t *ipvt, double *z)
{
#pragma clang fp contract(off)
#define ABD(i, j) abd[((size_t)(j) - 1) * lda + ((i) - 1)]
    double anorm = 0.0;
    int l = ml + 1, is = l + mu, info;
    for (int j = 1; j <= n; ++j) {
        const double s = sasum(l, &ABD(is, j));
        if (s > anorm) anorm = s;
        if (is > ml + 1) is = is - 1;
        if (j <= mu) l = l + 1;
        if (j >= n - ml) l = l - 1;
    }
    sgbfa(abd, lda, n, ml, mu, ipvt, &info);

    double ek = 1.0, s, sm, t, wk, wkm, ynorm;
    for (int j = 0; j < n; ++j) z[j] = 0.0;
    const int m = ml + mu + 1;
    int ju = 0;
    for (int k = 1; k <= n; ++k) {                  // solve trans(U) w = e
        if (z[k - 1] != 0.0) ek = dsign(ek, -z[k - 1]);
        if (fabs(ek - z[k - 1]) > fabs(ABD(m, k))) {
            s = fabs(ABD(m, k)) / fabs(ek - z[k - 1]);
            sscal(n, s, z);
            ek = s * ek;
        }
        wk = ek - z[k - 1];
        wkm = -ek - z[k - 1];
        s = fabs(wk);
        sm = fabs(wkm);
        if (ABD(m, k) != 0.0) { wk = wk / ABD(m, k); wkm = wkm / ABD(m, k); }
        else { wk = 1.0; wkm = 1.0; }
        const int kp1 = k + 1;
        ju = imin(imax(ju, mu + ipvt[k - 1]), n);
        int mm = m;
        if (kp1 <= ju) {
            for (int j = kp1; j <= ju; ++j) {
                mm = mm - 1;
                sm = sm + fabs(z[j - 1] + wkm * ABD(mm, j));
                z[j - 1] = z[j - 1] + wk * ABD(mm, j);
                s = s + fabs(z[j - 1]);
            }
            if (s < sm) {
                t = wkm - wk;
                wk = wkm;
                mm = m;
                for (int j = kp1; j <= ju; ++j) { mm = mm - 1; z[j - 1] = z[j - 1] + t * ABD(mm, j); }
            }
        }
        z[k - 1] = wk;
    }
    s = 1.0 / sasum(n, z);
    sscal(n, s, z);
    for (int kb = 1; kb <= n; ++kb) {               // solve trans(L) y = w
        const int k = n + 1 - kb;
        const int lm = imin(ml, n - k);
        if (k < n) z[k - 1] = z[k - 1] + sdot(lm, &ABD(m + 1, k), &z[k]);
        if (fabs(z[k - 1]) > 1.0) { s = 1.0 / fabs(z[k - 1]); sscal(n, s, z); }
        const int lp = ipvt[k - 1];
        t = z[lp - 1]; z[lp - 1] = z[k - 1]; z[k - 1] = t;
    }
    s = 1.0 / sasum(n, z);
    sscal(n, s, z);
    ynorm = 1.0;
    for (int k = 1; k <= n; ++k) {                  // solve L v = y
        const int lp = ipvt[k - 1];
        t = z[lp - 1]; z[lp - 1] = z[k - 1]; z[k - 1] = t;
        const int lm = imin(ml, n - k);
        if (k < n) saxpy(lm, t, &ABD(m + 1, k), &z[k]);
        if (fabs(z[k - 1]) > 1.0) { s = 1.0 / fabs(z[k - 1]); sscal(n, s, z); ynorm = s * ynorm; }
    }
    s = 1.0 / sasum(n, z);
    sscal(n, s, z);
    ynorm = s * ynorm;
    for (int kb = 1; kb <= n; ++kb) {               // solve U z = v
        const int k = n + 1 - kb;
        if (fabs(z[k - 1]) > fabs(ABD(m, k))) {
            s = fabs(ABD(m, k)) / fabs(z[k - 1]);
            sscal(n, s, z);
            ynorm = s * ynorm;
        }
        if (ABD(m, k) != 0.0) z[k - 1] = z[k - 1] / ABD(m, k);
        if (ABD(m, k) == 0.0) z[k - 1] = 1.0;
        const int lm = imin(k, m) - 1;
        const int la = m - lm, lz = k - lm;
        t = -z[k - 1];
        saxpy(lm, t, &ABD(la, k), &z[lz - 1]);
    }
    s = 1.0 / sasum(n, z);
    sscal(n, s, z);
    ynorm = s * ynorm;
    return (anorm != 0.0) ? ynorm / anorm : 0.0;
#undef ABD
}

// Scratch of one system (doubles), everything but the per-layer eigenproblem work areas: GC, KK, DTAUCP, TAUCPR, CBAND, Z,
// IPVT (int32, counted in doubles), nn doubles for SETMTX
struct SystemScratch {
    size_t gc, kk, dtaucp, taucpr, cband, z, ipvt, wk, total;
    int lda;
    SBD_RB SystemScratch(int n, int L)
    {
        const size_t nn = n / 2;
        lda = (int)(9 * nn - 2);
        gc = 0;
        kk = gc + (size_t)L * n * n;
        dtaucp = kk + (size_t)L * n;
        taucpr = dtaucp + L;
        cband = taucpr + L + 1;
        z = cband + (size_t)lda * n * L;
        ipvt = z + (size_t)n * L;
        wk = ipvt + ((size_t)n * L + 1) / 2;
        total = wk + nn;
        total = (total + 1) & ~(size_t)1;
    }
};

// TAUCPR (disort.f:2581: a running sum, layer after layer), SETMTX, SGBCO: the band system's RCOND from the layers' GC /
// KK / DTAUCP that soleig_layer left in the scratch block s (laid out by SystemScratch).
SBD_RB double band_rcond_from_layers(int n, int L, int ncut, bool lyrcut, bool lamber, int mazim, double albedo,
                                     const double *bdr, const double *cmu, const double *cwt, double *s)
{
#pragma clang fp contract(off)
    const SystemScratch o(n, L);
    const int nn = n / 2, ncd = 3 * nn - 1;
    double *taucpr = s + o.taucpr;
    const double *dtaucp = s + o.dtaucp;
    taucpr[0] = 0.0;
    for (int lc = 1; lc <= L; ++lc) taucpr[lc] = taucpr[lc - 1] + dtaucp[lc - 1];
    const double delm0 = (mazim == 0) ? 1.0 : 0.0;
    setmtx(n, ncut, lyrcut, lamber, delm0, albedo, bdr, cmu, cwt, s + o.gc, s + o.kk, dtaucp, taucpr, s + o.cband, o.lda, s + o.wk);
    return sgbco(s + o.cband, o.lda, n * ncut, ncd, ncd, (int32_t *)(s + o.ipvt), s + o.z);
}

}  // namespace refband
}  // namespace sbd
