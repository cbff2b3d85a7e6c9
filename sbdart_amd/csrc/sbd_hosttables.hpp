// Per-run tables on the host: the quadrature (QGAUSN, disort.f:5984-6157) and the normalised associated Legendre
// functions (LEPOLY, disort.f:5286-5408), fp64 with the reference's fp32-widened constants.  Used by sbd_engine_create
// and by sbd_band_rcond_host (sbd_k_refband.hip).
#pragma once
#include <cmath>
#include <cstddef>

namespace sbd {
namespace hosttab {

inline double ref_pi() { return (double)(2.0f * asinf(1.0f)); }              // disort.f:441
inline double ref_sqt(int k) { return (double)sqrtf((float)k); }             // disort.f:452-454

// Gauss-Legendre rule on (0,1), Newton with cubic correction (QGAUSN, disort.f:5984-6157)
inline void gauss01(int m, double *gmu, double *gwt)
{
    const double pi = ref_pi(), tol = 10.0 * 2.220446049250313e-16;
    if (m == 1) { gmu[0] = 0.5; gwt[0] = 1.0; return; }
    const double en = m, nnp1 = (double)(m * (m + 1));
    const double cona = (double)((float)(m - 1) / (float)(8 * m * m * m));
    const int lim = m / 2;
    for (int k = 1; k <= lim; ++k) {
        const double t = (double)(4 * k - 1) * pi / (double)(4 * m + 2);
        double x = cos(t + cona / tan(t)), p = 0, pm1, pm2, tmp, ppr;
        for (;;) {
            pm2 = 1.0;
            pm1 = x;
            for (int nn = 2; nn <= m; ++nn) {
                p = ((double)(2 * nn - 1) * x * pm1 - (double)(nn - 1) * pm2) / (double)nn;
                pm2 = pm1;
                pm1 = p;
            }
            tmp = 1.0 / (1.0 - x * x);
            ppr = en * (pm2 - x * p) * tmp;
            const double p2pri = (2.0 * x * ppr - nnp1 * p) * tmp;
            const double xi = x - (p / ppr) * (1.0 + (p / ppr) * p2pri / (2.0 * ppr));
            if (fabs(xi - x) > tol) { x = xi; continue; }
            break;
        }
        const double ep = en * pm2;
        gmu[k - 1] = -x;
        gwt[k - 1] = 2.0 / (tmp * (ep * ep));
        gmu[m - k] = x;
        gwt[m - k] = gwt[k - 1];
    }
    if (m % 2) {
        gmu[lim] = 0.0;
        double prod = 1.0;
        for (int k = 3; k <= m; k += 2) prod = prod * (double)k / (double)(k - 1);
        gwt[lim] = 2.0 / (prod * prod);
    }
    for (int k = 0; k < m; ++k) { gmu[k] = 0.5 * gmu[k] + 0.5; gwt[k] = 0.5 * gwt[k]; }
}

// normalised associated Legendre functions, degree recurrence per order m (LEPOLY,
// disort.f:5286-5408); ylm[i*(maxl+1)+l]; needs order m-1 in place for m > 0.
inline void legendre_norm(int nmu, int m, int maxl, int twonm1, const double *mu, double *ylm)
{
    auto Y = [&](int l, int i) -> double & { return ylm[(size_t)i * (maxl + 1) + l]; };
    if (m == 0) {
        for (int i = 0; i < nmu; ++i) { Y(0, i) = 1.0; Y(1, i) = mu[i]; }
        for (int l = 2; l <= twonm1; ++l)
            for (int i = 0; i < nmu; ++i)
                Y(l, i) = ((double)(2 * l - 1) * mu[i] * Y(l - 1, i) - (double)(l - 1) * Y(l - 2, i)) / (double)l;
    } else {
        for (int i = 0; i < nmu; ++i) {
            Y(m, i) = -ref_sqt(2 * m - 1) / ref_sqt(2 * m) * sqrt(1.0 - mu[i] * mu[i]) * Y(m - 1, i);
            Y(m + 1, i) = ref_sqt(2 * m + 1) * mu[i] * Y(m, i);
        }
        for (int l = m + 2; l <= twonm1; ++l) {
            const double t1 = ref_sqt(l - m) * ref_sqt(l + m), t2 = ref_sqt(l - m - 1) * ref_sqt(l + m - 1);
            for (int i = 0; i < nmu; ++i)
                Y(l, i) = ((double)(2 * l - 1) * mu[i] * Y(l - 1, i) - t2 * Y(l - 2, i)) / t1;
        }
    }
}

}  // namespace hosttab
}  // namespace sbd
