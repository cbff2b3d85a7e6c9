// Types shared by the gas model's source (sbd_gas.hpp, compiled without contraction in its own translation unit) and the
// host side of the C ABI (sbd_engine.hip): table views, the per-run block a launch takes, the repacked table image.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>

namespace sbd {
namespace gas {

constexpr int NMOL = 11;            // h2o co2 o3 n2o co ch4 o2 no so2 no2 nh3
constexpr int MK = 3;               // k-distribution terms
constexpr int MXQ = 63;             // absorber-amount slots (params.f:14)
constexpr int MAXLYR = 66;          // mxly + 1 (SPOWDER's extra layer)

struct Tab { const double *p; int n; };
struct TabI { const int32_t *p; int n; };
struct Tables {                     // pointers into one repacked image of sbdart_tables.bin (host or device copy)
    Tab self296, self260, foreign, n2, h1, h2, h3, o2s0, o2a, o2b, o4, o3uv, hh0, hh1, hh2, chap, schrun;
    Tab cp[NMOL], bs[NMOL], ba[NMOL], bb[NMOL], bc[NMOL];
    TabI lo[NMOL], hi[NMOL];
};

}  // namespace gas

constexpr double kReEarth = (double)6371.2f;        // params.f:22 (a REAL*4 literal, widened)

struct GasTablesPacked {            // the tables of an image, one aligned array per type + a directory by name
    struct Entry { int kind = 0; size_t off = 0; int n = 0; };
    std::vector<double> d;
    std::vector<int32_t> i;
    std::map<std::string, Entry> dir;
    bool parse(const void *image, size_t bytes, std::string &err);
    bool view(const double *dbase, const int32_t *ibase, gas::Tables &T, std::string &err) const;
};

struct GasRun {                     // what gas_kernel takes by value: table views and the run's profile, device pointers
    gas::Tables T;
    const double *uu, *z;
    int nz, kdist;
    double amu0_first, amu0_rest, xo4, re_earth;
};

void launch_gas(hipStream_t st, const GasRun &R, int npoint, int first_is_run_first, const double *wl, const double *lay, int nch,
                double *ws, size_t npad, int32_t *nk, double *wt, int32_t *fail, double *slots);

}  // namespace sbd
