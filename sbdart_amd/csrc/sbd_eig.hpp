// Cooperative real-spectrum eigen-solver for the reduced DISORT eigenproblem
// (the role ASYMTX plays in the reference, disort.f:873-1656): balance ->
// Householder/Hessenberg -> shifted double-QR -> back-substitution.
//
// MI355X mapping: one GROUP of G lanes (G = power of two >= NSTR, a sub-wave) owns
// one M x M matrix (M = NSTR/2) that lives in LDS.  Scalar control flow (shifts,
// deflation tests, reflector construction) is evaluated redundantly by every
// lane of the group from broadcast LDS reads, so it stays uniform inside the
// group; every O(M) row / column / accumulation update is spread over the lanes
// (lane j owns column j or row j of the update) and keeps the element-wise
// operation order of the sequential algorithm, so the result matches a serial
// evaluation up to FMA contraction.  Groups of one wave may diverge from each
// other (different deflation histories); lanes of a group never do.
// Synchronisation is wave-level only (LDS ops of a wave are ordered).
#pragma once
#include "sbd_common.hpp"

namespace sbd {

// aa: M x M (leading dim ia), destroyed.  evec: M x M block of a matrix with leading
// dim iev.  eval[M].  wk[2M].  xs: M x M scratch (leading dim ia) for the
// back-substituted triangular eigenvectors.  g = lane index inside the group.
// Returns IER (0 = ok, >0 = eigenvalue IER did not converge, -1 complex 2x2).
SBD_DEVICE int eig_group(double *aa, int ia, double *evec, int iev, double *eval, int M,
                         double *wk, double *xs, int g)
{
#define AA(i, j) aa[((j) - 1) * ia + ((i) - 1)]
#define EV(i, j) evec[((j) - 1) * iev + ((i) - 1)]
#define XS(i, j) xs[((j) - 1) * ia + ((i) - 1)]
#define WK(i) wk[(i) - 1]
#define GSYNC() wave_lds_sync()
    const double c1 = 0.4375, c2 = 0.5, c3 = 0.75, c4 = 0.95, c5 = 16.0, c6 = 256.0;
    const double tol = 2.220446049250313e-16;
    const int me = g + 1;  // 1-based index this lane owns
    const bool act = me <= M;

    if (M == 1) {
        if (g == 0) { eval[0] = AA(1, 1); EV(1, 1) = 1.0; }
        GSYNC();
        return 0;
    }
    if (M == 2) {  // closed form (disort.f:989-1023)
        int ier = 0;
        const double a11 = AA(1, 1), a12 = AA(1, 2), a21 = AA(2, 1), a22 = AA(2, 2);
        const double discri = (a11 - a22) * (a11 - a22) + 4.0 * a12 * a21;
        if (discri < 0.0) ier = -1;
        GSYNC();
        if (g == 0 && ier == 0) {
            const double sgn = (a11 < a22) ? -1.0 : 1.0;
            const double e1 = 0.5 * (a11 + a22 + sgn * sqrt(discri));
            const double e2 = 0.5 * (a11 + a22 - sgn * sqrt(discri));
            eval[0] = e1;
            eval[1] = e2;
            EV(1, 1) = 1.0;
            EV(2, 2) = 1.0;
            if (a11 == a22 && (a21 == 0.0 || a12 == 0.0)) {
                const double w = tol * (fabs(a11) + fabs(a12) + fabs(a21) + fabs(a22));
                EV(2, 1) = a21 / w;
                EV(1, 2) = -a12 / w;
            } else {
                EV(2, 1) = a21 / (e1 - a22);
                EV(1, 2) = a12 / (e2 - a11);
            }
        }
        GSYNC();
        return ier;
    }

    // ---- initialise outputs: lane j owns column j ----
    if (act) {
        eval[me - 1] = 0.0;
        for (int i = 1; i <= M; ++i) EV(i, me) = (i == me) ? 1.0 : 0.0;
    }
    GSYNC();

    // ---- isolate eigenvalues (rows pushed down, columns pushed left) ----
    int l = 1, k = M;
    for (;;) {  // rows
        bool again = false;
        for (int j = k; j >= 1; --j) {
            double row = 0.0;
            for (int i = 1; i <= k; ++i)
                if (i != j) row += fabs(AA(j, i));
            if (row == 0.0) {
                if (g == 0) WK(k) = (double)j;
                if (j != k) {
                    double r1 = 0, r2 = 0, q1 = 0, q2 = 0;
                    const bool a1 = me <= k, a2 = (me >= l && me <= M);
                    if (a1) { r1 = AA(me, j); r2 = AA(me, k); }
                    GSYNC();
                    if (a1) { AA(me, j) = r2; AA(me, k) = r1; }
                    GSYNC();
                    if (a2) { q1 = AA(j, me); q2 = AA(k, me); }
                    GSYNC();
                    if (a2) { AA(j, me) = q2; AA(k, me) = q1; }
                }
                GSYNC();
                k = k - 1;
                again = true;
                break;
            }
        }
        if (!again) break;
    }
    for (;;) {  // columns
        bool again = false;
        for (int j = l; j <= k; ++j) {
            double col = 0.0;
            for (int i = l; i <= k; ++i)
                if (i != j) col += fabs(AA(i, j));
            if (col == 0.0) {
                if (g == 0) WK(l) = (double)j;
                if (j != l) {
                    double r1 = 0, r2 = 0, q1 = 0, q2 = 0;
                    const bool a1 = me <= k, a2 = (me >= l && me <= M);
                    if (a1) { r1 = AA(me, j); r2 = AA(me, l); }
                    GSYNC();
                    if (a1) { AA(me, j) = r2; AA(me, l) = r1; }
                    GSYNC();
                    if (a2) { q1 = AA(j, me); q2 = AA(l, me); }
                    GSYNC();
                    if (a2) { AA(j, me) = q2; AA(l, me) = q1; }
                }
                GSYNC();
                l = l + 1;
                again = true;
                break;
            }
        }
        if (!again) break;
    }

    // ---- balance rows l..k (sequential in i, like the serial algorithm) ----
    if (g == 0)
        for (int i = l; i <= k; ++i) WK(i) = 1.0;
    GSYNC();
    for (;;) {
        bool noconv = false;
        for (int i = l; i <= k; ++i) {
            double col = 0.0, row = 0.0;
            for (int j = l; j <= k; ++j)
                if (j != i) { col += fabs(AA(j, i)); row += fabs(AA(i, j)); }
            double f = 1.0, gg = row / c5;
            const double h = col + row;
            // col/row are > 0 here (zero rows/cols were isolated); guard NaN anyway
            if (!(col > 0.0) || !(row > 0.0)) continue;
            while (col < gg) { f *= c5; col *= c6; }
            gg = row * c5;
            while (col >= gg) { f /= c5; col /= c6; }
            if ((col + row) / f < c4 * h) {
                noconv = true;
                GSYNC();
                if (g == 0) WK(i) = WK(i) * f;
                if (me >= l && me <= M) AA(i, me) = AA(i, me) / f;
                GSYNC();
                if (me <= k) AA(me, i) = AA(me, i) * f;
                GSYNC();
            }
        }
        if (!noconv) break;
    }

    // ---- Hessenberg reduction (Householder), lane j = column j / lane i = row i ----
    if (!(k - 1 < l + 1)) {
        for (int n = l + 1; n <= k - 1; ++n) {
            double h = 0.0, scale = 0.0;
            for (int i = n; i <= k; ++i) scale += fabs(AA(i, n - 1));
            GSYNC();
            WK(n + M) = 0.0;  // every lane writes the same value
            if (scale != 0.0) {
                double wn = 0.0;
                for (int i = k; i >= n; --i) {
                    const double v = AA(i, n - 1) / scale;
                    if (i == n) wn = v;
                    h += v * v;
                    WK(i + M) = v;
                }
                const double gg = -dsign(sqrt(h), wn);
                h = h - wn * gg;
                WK(n + M) = wn - gg;
                GSYNC();
                if (me >= n && me <= M) {  // (I - u u^T / h) A, column me
                    double f = 0.0;
                    for (int i = k; i >= n; --i) f += WK(i + M) * AA(i, me);
                    for (int i = n; i <= k; ++i) AA(i, me) = AA(i, me) - WK(i + M) * f / h;
                }
                GSYNC();
                if (me <= k) {  // ... A (I - u u^T / h), row me
                    double f = 0.0;
                    for (int j = k; j >= n; --j) f += WK(j + M) * AA(me, j);
                    for (int j = n; j <= k; ++j) AA(me, j) = AA(me, j) - WK(j + M) * f / h;
                }
                GSYNC();
                WK(n + M) = scale * (wn - gg);
                AA(n, n - 1) = scale * gg;
            }
            GSYNC();
        }
        for (int n = k - 2; n >= l; --n) {  // accumulate transformations
            double f = AA(n + 1, n);
            if (f != 0.0) {
                f = f * WK(n + 1 + M);
                GSYNC();
                for (int i = n + 2; i <= k; ++i) WK(i + M) = AA(i, n);
                GSYNC();
                if (act) {
                    double gg = 0.0;
                    for (int i = n + 1; i <= k; ++i) gg += WK(i + M) * EV(i, me);
                    gg = gg / f;
                    for (int i = n + 1; i <= k; ++i) EV(i, me) = EV(i, me) + gg * WK(i + M);
                }
                GSYNC();
            }
        }
    }

    // ---- norm, isolated eigenvalues ----
    double rnorm = 0.0;
    {
        int n = 1;
        for (int i = 1; i <= M; ++i) {
            for (int j = n; j <= M; ++j) rnorm += fabs(AA(i, j));
            n = i;
        }
        if (act && (me < l || me > k)) eval[me - 1] = AA(me, me);
    }
    GSYNC();

    // ---- QR iterations ----
    int n = k;
    double t = 0.0;
    double p = 0.0, q = 0.0, r = 0.0, s, x, y, z, w;
    while (n >= l) {
        int in = 0;
        const int n1 = n - 1, n2 = n - 2;
        for (;;) {
            int lb = l;
            for (int i = l; i <= n; ++i) {
                lb = n + l - i;
                if (lb == l) break;
                s = fabs(AA(lb - 1, lb - 1)) + fabs(AA(lb, lb));
                if (s == 0.0) s = rnorm;
                if (fabs(AA(lb, lb - 1)) <= tol * s) break;
            }
            x = AA(n, n);
            if (lb == n) {  // one eigenvalue
                GSYNC();
                if (g == 0) { AA(n, n) = x + t; eval[n - 1] = x + t; }
                GSYNC();
                n = n1;
                break;
            }
            y = AA(n1, n1);
            w = AA(n, n1) * AA(n1, n);
            if (lb == n1) {  // two eigenvalues
                p = (y - x) * c2;
                q = p * p + w;
                z = sqrt(fabs(q));
                const double xx = x + t;
                z = p + dsign(z, p);
                const double e1 = xx + z;
                double e2 = e1;
                if (z != 0.0) e2 = xx - w / z;
                x = AA(n, n1);
                r = sqrt(x * x + z * z);
                p = x / r;
                q = z / r;
                GSYNC();
                if (g == 0) { AA(n, n) = xx; AA(n1, n1) = y + t; eval[n1 - 1] = e1; eval[n - 1] = e2; }
                GSYNC();
                if (me >= n1 && me <= M) {  // row modification, column me
                    z = AA(n1, me);
                    AA(n1, me) = q * z + p * AA(n, me);
                    AA(n, me) = q * AA(n, me) - p * z;
                }
                GSYNC();
                if (me <= n) {  // column modification, row me
                    z = AA(me, n1);
                    AA(me, n1) = q * z + p * AA(me, n);
                    AA(me, n) = q * AA(me, n) - p * z;
                }
                if (me >= l && me <= k) {  // accumulate
                    z = EV(me, n1);
                    EV(me, n1) = q * z + p * EV(me, n);
                    EV(me, n) = q * EV(me, n) - p * z;
                }
                GSYNC();
                n = n2;
                break;
            }
            if (in == 30) return n;  // no convergence
            if (in == 10 || in == 20) {  // exceptional shift
                t = t + x;
                GSYNC();
                if (me >= l && me <= n) AA(me, me) = AA(me, me) - x;
                GSYNC();
                s = fabs(AA(n, n1)) + fabs(AA(n1, n2));
                x = c3 * s;
                y = x;
                w = -c1 * (s * s);
            }
            in = in + 1;
            int i = lb;
            for (int j = lb; j <= n2; ++j) {
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
                const double uu = fabs(AA(i, i - 1)) * (fabs(q) + fabs(r));
                const double vv = fabs(p) * (fabs(AA(i - 1, i - 1)) + fabs(z) + fabs(AA(i + 1, i + 1)));
                if (uu <= tol * vv) break;
            }
            GSYNC();
            if (g == 0) {
                AA(i + 2, i) = 0.0;
                for (int j = i + 3; j <= n; ++j) { AA(j, j - 2) = 0.0; AA(j, j - 3) = 0.0; }
            }
            GSYNC();
            for (int ka = i; ka <= n1; ++ka) {  // double QR sweep
                const bool notlas = (ka != n1);
                double newsub = 0.0;
                bool setsub = false, negsub = false;
                if (ka == i) {
                    s = dsign(sqrt(p * p + q * q + r * r), p);
                    if (lb != i) negsub = true;
                } else {
                    p = AA(ka, ka - 1);
                    q = AA(ka + 1, ka - 1);
                    r = notlas ? AA(ka + 2, ka - 1) : 0.0;
                    x = fabs(p) + fabs(q) + fabs(r);
                    if (x == 0.0) continue;
                    p = p / x;
                    q = q / x;
                    r = r / x;
                    s = dsign(sqrt(p * p + q * q + r * r), p);
                    newsub = -s * x;
                    setsub = true;
                }
                p = p + s;
                x = p / s;
                y = q / s;
                z = r / s;
                q = q / p;
                r = r / p;
                GSYNC();
                if (g == 0) {
                    if (negsub) AA(ka, ka - 1) = -AA(ka, ka - 1);
                    if (setsub) AA(ka, ka - 1) = newsub;
                }
                if (me >= ka && me <= M) {  // row modification, column me
                    double pp = AA(ka, me) + q * AA(ka + 1, me);
                    if (notlas) {
                        pp = pp + r * AA(ka + 2, me);
                        AA(ka + 2, me) = AA(ka + 2, me) - pp * z;
                    }
                    AA(ka + 1, me) = AA(ka + 1, me) - pp * y;
                    AA(ka, me) = AA(ka, me) - pp * x;
                }
                GSYNC();
                const int iimax = (n < ka + 3) ? n : ka + 3;
                if (me <= iimax) {  // column modification, row me
                    double pp = x * AA(me, ka) + y * AA(me, ka + 1);
                    if (notlas) {
                        pp = pp + z * AA(me, ka + 2);
                        AA(me, ka + 2) = AA(me, ka + 2) - pp * r;
                    }
                    AA(me, ka + 1) = AA(me, ka + 1) - pp * q;
                    AA(me, ka) = AA(me, ka) - pp;
                }
                if (me >= l && me <= k) {  // accumulate, row me of EVEC
                    double pp = x * EV(me, ka) + y * EV(me, ka + 1);
                    if (notlas) {
                        pp = pp + z * EV(me, ka + 2);
                        EV(me, ka + 2) = EV(me, ka + 2) - pp * r;
                    }
                    EV(me, ka + 1) = EV(me, ka + 1) - pp * q;
                    EV(me, ka) = EV(me, ka) - pp;
                }
                GSYNC();
            }
        }
    }

    // ---- back-substitution: lane n owns eigenvector column n (into XS) ----
    GSYNC();
    if (rnorm != 0.0) {
        if (act) {
            const int nc = me;
            const double ev_n = eval[nc - 1];
            XS(nc, nc) = 1.0;
            int n2 = nc;
            for (int i = nc - 1; i >= 1; --i) {
                double ww = AA(i, i) - ev_n;
                if (ww == 0.0) ww = tol * rnorm;
                double rr = AA(i, nc);
                for (int j = n2; j <= nc - 1; ++j) rr = rr + AA(i, j) * XS(j, nc);
                XS(i, nc) = -rr / ww;
                n2 = i;
            }
        }
        GSYNC();
        if (act && (me < l || me > k))  // vectors of isolated eigenvalues, row me
            for (int j = me; j <= M; ++j) EV(me, j) = XS(me, j);
        if (k != 0 && me >= l && me <= k) {  // multiply by the transformation matrix, row me
            for (int j = M; j >= l; --j) {
                double zz = 0.0;
                const int nmax = (j < k) ? j : k;
                for (int nn_ = l; nn_ <= nmax; ++nn_) zz = zz + EV(me, nn_) * XS(nn_, j);
                EV(me, j) = zz;
            }
        }
        GSYNC();
    }
    if (me >= l && me <= k) {
        const double sc = WK(me);
        for (int j = 1; j <= M; ++j) EV(me, j) = EV(me, j) * sc;
    }
    GSYNC();
    for (int i = l - 1; i >= 1; --i) {  // undo permutations (rare)
        const int j = (int)WK(i);
        if (i != j) {
            double a = 0, b = 0;
            if (act) { a = EV(i, me); b = EV(j, me); }
            GSYNC();
            if (act) { EV(i, me) = b; EV(j, me) = a; }
            GSYNC();
        }
    }
    for (int i = k + 1; i <= M; ++i) {
        const int j = (int)WK(i);
        if (i != j) {
            double a = 0, b = 0;
            if (act) { a = EV(i, me); b = EV(j, me); }
            GSYNC();
            if (act) { EV(i, me) = b; EV(j, me) = a; }
            GSYNC();
        }
    }
    return 0;
#undef AA
#undef EV
#undef XS
#undef WK
#undef GSYNC
}

}  // namespace sbd
