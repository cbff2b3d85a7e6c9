// Launch interface of the scatterers' kernel (sbd_k_scat.hip, source sbd_scat.hpp compiled without contraction there) for
// the host side of the C ABI (sbd_engine.hip).
#pragma once
#include <string>
#include <hip/hip_runtime.h>
#include "../../include/sbdart_amd.h"
#include "sbd_gas_types.hpp"

namespace sbd {
namespace scat { struct Model; }
// the C ABI's model -> the source's; profile / spectrum pointers are taken as they are (the caller points them at host or
// device copies first), the Mie and stratospheric tables are found in the repacked image whose doubles start at dbase
bool scat_model_view(const sbd_scat_model *sm, const GasTablesPacked &pk, const double *dbase, scat::Model &m, std::string &err);
int scat_model_terms(const scat::Model &m);
void launch_scatter(hipStream_t st, const scat::Model &M, int npoint, const double *wl, double *lay, int nch);
// ... for a caller that does not see scat::Model (sbd_engine.hip keeps its own contraction setting): `dsm` is the ABI's
// model with its profile / spectrum pointers already pointing at DEVICE copies; returns false with `err` set when the model
// is inconsistent or nch is not 4 + 3 x its terms
bool launch_scatter_abi(hipStream_t st, const sbd_scat_model *dsm, const GasTablesPacked &pk, const double *dbase, int npoint,
                        const double *wl, double *lay, int nch, std::string &err);
}  // namespace sbd
