// The gas part of the band model (sbd_gas.hpp) as a kernel -- one thread per wavelength, work arrays interleaved over
// the wavelengths of a launch -- and as plain host code behind sbd_gas_terms_host: the SAME source on both sides.
// Compiled without contraction (sbd_gas.hpp sets it for everything that follows it in this file).
#include "../../include/sbdart_amd.h"
#include "sbd_gas_types.hpp"
#include "sbd_gas.hpp"

#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace sbd {

// sbdart_amd/data/sbdart_tables.bin: "SBDTBL1" + pad, int32 count, then per table char name[24], int32 kind (1 = f64),
// int32 n, the values (int32 tables padded to an even count).  The values sit at odd multiples of four bytes: they are
// repacked into one aligned array per type.
bool GasTablesPacked::parse(const void *image, size_t bytes, std::string &err)
{
    const unsigned char *b = (const unsigned char *)image;
    if (!b || bytes < 12 || memcmp(b, "SBDTBL1", 7) != 0) { err = "gas tables: not an image of sbdart_tables.bin"; return false; }
    int32_t ntab = 0;
    memcpy(&ntab, b + 8, 4);
    size_t o = 12;
    d.clear(); i.clear(); dir.clear();
    for (int k = 0; k < ntab; ++k) {
        if (o + 32 > bytes) { err = "gas tables: truncated image"; return false; }
        char name[25];
        memcpy(name, b + o, 24);
        name[24] = 0;
        for (int c = 23; c >= 0 && (name[c] == ' ' || name[c] == 0); --c) name[c] = 0;
        int32_t kind = 0, n = 0;
        memcpy(&kind, b + o + 24, 4);
        memcpy(&n, b + o + 28, 4);
        o += 32;
        if (n < 0) { err = "gas tables: negative length"; return false; }
        Entry e;
        e.kind = kind; e.n = n;
        if (kind == 1) {
            if (o + 8 * (size_t)n > bytes) { err = "gas tables: truncated image"; return false; }
            e.off = d.size();
            d.resize(d.size() + (size_t)n);
            memcpy(d.data() + e.off, b + o, 8 * (size_t)n);
            o += 8 * (size_t)n;
        } else {
            const size_t np = (size_t)n + (n & 1);
            if (o + 4 * np > bytes) { err = "gas tables: truncated image"; return false; }
            e.off = i.size();
            i.resize(i.size() + (size_t)n);
            memcpy(i.data() + e.off, b + o, 4 * (size_t)n);
            o += 4 * np;
        }
        dir[name] = e;
    }
    // every table the gas model reads must be there
    gas::Tables T;
    return view(d.data(), i.data(), T, err);
}

bool GasTablesPacked::view(const double *dbase, const int32_t *ibase, gas::Tables &T, std::string &err) const
{
    bool ok = true;
    auto rt = [&](const std::string &name) {
        gas::Tab t{nullptr, 0};
        auto it = dir.find(name);
        if (it == dir.end() || it->second.kind != 1) { err = "gas tables: table " + name + " is missing"; ok = false; return t; }
        t.p = dbase + it->second.off;
        t.n = it->second.n;
        return t;
    };
    auto it_ = [&](const std::string &name) {
        gas::TabI t{nullptr, 0};
        auto it = dir.find(name);
        if (it == dir.end() || it->second.kind == 1) { err = "gas tables: table " + name + " is missing"; ok = false; return t; }
        t.p = ibase + it->second.off;
        t.n = it->second.n;
        return t;
    };
    T.self296 = rt("h2o.self296"); T.self260 = rt("h2o.self260"); T.foreign = rt("h2o.foreign");
    T.n2 = rt("n2.cont"); T.h1 = rt("hno3.h1"); T.h2 = rt("hno3.h2"); T.h3 = rt("hno3.h3");
    T.o2s0 = rt("o2.s0"); T.o2a = rt("o2.a"); T.o2b = rt("o2.b"); T.o4 = rt("o4.sig");
    T.o3uv = rt("o3.uv"); T.hh0 = rt("o3.hh0"); T.hh1 = rt("o3.hh1"); T.hh2 = rt("o3.hh2");
    T.chap = rt("o3.chappuis"); T.schrun = rt("o2.schrun");
    static const char *mol[gas::NMOL] = {"h2o", "co2", "o3", "n2o", "co", "ch4", "o2", "no", "so2", "no2", "nh3"};
    for (int m = 0; m < gas::NMOL; ++m) {
        const std::string s = mol[m];
        T.cp[m] = rt("cp." + s); T.bs[m] = rt("bs." + s); T.ba[m] = rt("ba." + s); T.bb[m] = rt("bb." + s); T.bc[m] = rt("bc." + s);
        T.lo[m] = it_("iwl." + s); T.hi[m] = it_("iwh." + s);
    }
    return ok;
}

// One thread per wavelength.  ws: [16 nz][npad] doubles, lane-interleaved (element i of thread p's arrays at
// ws[i * npad + p]); slots [npoint][MK][nz]: the terms' gas depths; wt [npoint][MK]; nk, fail [npoint].
// first_is_run_first: point 0 of this launch is the run's first wavelength (the SZA >= 90 quirk, drt.f:433-455).
__global__ void __launch_bounds__(64) gas_kernel(GasRun R, int npoint, int first_is_run_first, const double *wl, const double *lay,
                                                 int nch, double *ws, size_t npad, int32_t *nk, double *wt, int32_t *fail, double *slots)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npoint) return;
    const int nz = R.nz;
    const double amu0 = (p == 0 && first_is_run_first) ? R.amu0_first : R.amu0_rest;
    int nkp = 1;
    double w3[gas::MK];
    const int rc = gas::point_gas(R.T, R.kdist, R.xo4, R.uu, R.z, nz, R.re_earth, wl[p], amu0, lay + (size_t)p * nch * nz,
                                  ws + p, npad, nkp, w3, slots + (size_t)p * gas::MK * nz);
    nk[p] = nkp;
    fail[p] = rc;
    for (int k = 0; k < gas::MK; ++k) wt[(size_t)p * gas::MK + k] = w3[k];
}

void launch_gas(hipStream_t st, const GasRun &R, int npoint, int first_is_run_first, const double *wl, const double *lay, int nch,
                double *ws, size_t npad, int32_t *nk, double *wt, int32_t *fail, double *slots)
{
    if (npoint <= 0) return;
    hipLaunchKernelGGL(gas_kernel, dim3((unsigned)((npoint + 63) / 64)), dim3(64), 0, st, R, npoint, first_is_run_first, wl, lay, nch,
                       ws, npad, nk, wt, fail, slots);
}

}  // namespace sbd

extern "C" int sbd_gas_terms_host(const sbd_gas_model *g, int32_t nlyr, int32_t npoint, const double *wl, const double *lay, int32_t nch,
                                  int32_t *nk, double *wt, int32_t *fail, double *dtaug_out)
{
    if (!g || !wl || !lay || !nk || !wt || !g->uu || !g->z || !g->tables) return SBD_E_INVALID;
    if (g->nz < 1 || g->nz > sbd::gas::MAXLYR || nlyr != g->nz || nch < 3 || npoint < 0) return SBD_E_INVALID;
    sbd::GasTablesPacked P;
    std::string err;
    if (!P.parse(g->tables, g->tables_bytes, err)) return SBD_E_INVALID;
    sbd::gas::Tables T;
    if (!P.view(P.d.data(), P.i.data(), T, err)) return SBD_E_INVALID;
    const int nz = g->nz;
    std::vector<double> ws((size_t)16 * nz), slots((size_t)sbd::gas::MK * nz);
    for (int p = 0; p < npoint; ++p) {
        const double amu0 = (p == 0) ? g->amu0_first : g->amu0_rest;
        int nkp = 1;
        double w3[sbd::gas::MK];
        std::fill(slots.begin(), slots.end(), 0.0);
        const int rc = sbd::gas::point_gas(T, g->kdist, g->xo4, g->uu, g->z, nz, sbd::kReEarth, wl[p], amu0, lay + (size_t)p * nch * nz,
                                           ws.data(), 1, nkp, w3, slots.data());
        nk[p] = nkp;
        if (fail) fail[p] = rc;
        for (int k = 0; k < sbd::gas::MK; ++k) wt[(size_t)p * sbd::gas::MK + k] = w3[k];
        if (dtaug_out) memcpy(dtaug_out + (size_t)p * sbd::gas::MK * nz, slots.data(), sizeof(double) * sbd::gas::MK * nz);
    }
    return SBD_OK;
}
