// The scatterers' part of the band model (sbd_scat.hpp) as a kernel -- one thread per wavelength writes its layer block
// where the gas kernel and the solves read it -- and as plain host code behind sbd_scatter_blocks_host: the SAME source.
// Compiled without contraction (sbd_scat.hpp sets it for everything that follows it in this file).
#include "../../include/sbdart_amd.h"
#include "sbd_gas_types.hpp"
#include "sbd_scat_types.hpp"
#include "sbd_scat.hpp"

#include <cstring>
#include <string>

namespace sbd {

// the C ABI's model -> the source's, tables resolved in a repacked image (dbase: host or device copy of pk.d)
bool scat_model_view(const sbd_scat_model *sm, const GasTablesPacked &pk, const double *dbase, scat::Model &m, std::string &err)
{
    if (!sm || !sm->z || !sm->p || !sm->t) { err = "scatter model: null profile"; return false; }
    if (sm->nz < 1 || sm->nz > 66) { err = "scatter model: nz outside 1..66"; return false; }
    m.nz = sm->nz; m.z = sm->z; m.p = sm->p; m.t = sm->t; m.xrsc = sm->xrsc;
    m.cloud_term = sm->cloud_term ? 1 : 0;
    m.cld_nslot = sm->cld_nslot;
    if (m.cld_nslot < 0 || m.cld_nslot > scat::NCLD) { err = "scatter model: cloud slots outside 0..5"; return false; }
    for (int i = 0; i < scat::NCLD; ++i) {
        m.cld_layer[i] = sm->cld_layer[i]; m.cld_tcloud[i] = sm->cld_tcloud[i]; m.cld_lwp[i] = sm->cld_lwp[i]; m.cld_nre[i] = sm->cld_nre[i];
        if (m.cld_layer[i] > m.nz || m.cld_layer[i] < -m.nz) { err = "scatter model: cloud layer outside the atmosphere"; return false; }
    }
    m.iaer = sm->iaer; m.nosct = sm->nosct; m.aer_nwl = sm->aer_nwl;
    m.aer_wl = sm->aer_wl; m.aer_ext = sm->aer_ext; m.aer_absb = sm->aer_absb; m.aer_asym = sm->aer_asym;
    m.abaer = sm->abaer; m.aer_column = sm->aer_column;
    if (m.iaer != 0 && (m.aer_nwl < 2 || !m.aer_wl || !m.aer_ext || !m.aer_absb || !m.aer_asym || !m.aer_column)) {
        err = "scatter model: boundary-layer aerosol without its spectrum / column"; return false;
    }
    m.nstrat = sm->nstrat;
    if (m.nstrat < 0 || m.nstrat > scat::NAERZ) { err = "scatter model: stratospheric layers outside 0..5"; return false; }
    for (int i = 0; i < scat::NAERZ; ++i) {
        m.jaer[i] = sm->jaer[i]; m.strat_layer[i] = sm->strat_layer[i]; m.taerst[i] = sm->taerst[i];
        if (i < m.nstrat && m.jaer[i] != 0 && m.taerst[i] > 0.0 && (m.jaer[i] < 1 || m.jaer[i] > 4 || m.strat_layer[i] < 1 || m.strat_layer[i] > m.nz)) {
            err = "scatter model: stratospheric aerosol model / layer out of range"; return false;
        }
    }
    static const char *names[6] = {"cloud.q", "cloud.w", "cloud.g", "cloud.qi", "cloud.wi", "cloud.gi"};
    auto find = [&](const char *name, int need) -> const double * {
        auto it = pk.dir.find(name);
        if (it == pk.dir.end() || it->second.kind != 1 || it->second.n < need) { err = std::string("scatter model: table ") + name + " is missing"; return nullptr; }
        return dbase + it->second.off;
    };
    for (int k = 0; k < 6; ++k) { m.mie[k] = find(names[k], scat::MXWV * scat::MRE); if (!m.mie[k]) return false; }
    m.strat_wl = find("aer.wl", scat::NAERW);
    m.strat_tab = find("aer.strat", 4 * 3 * scat::NAERW);
    return m.strat_wl && m.strat_tab;
}

int scat_model_terms(const scat::Model &m) { return scat::nterm(m); }

// One thread per wavelength: lay [npoint][nch][nz].
__global__ void __launch_bounds__(64) scatter_kernel(scat::Model M, int npoint, const double *wl, double *lay, int nch)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npoint) return;
    scat::point_block(M, wl[p], lay + (size_t)p * nch * M.nz, (size_t)M.nz, 1);
}

void launch_scatter(hipStream_t st, const scat::Model &M, int npoint, const double *wl, double *lay, int nch)
{
    if (npoint <= 0) return;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((npoint + 63) / 64)), dim3(64), 0, st, M, npoint, wl, lay, nch);
}

bool launch_scatter_abi(hipStream_t st, const sbd_scat_model *dsm, const GasTablesPacked &pk, const double *dbase, int npoint,
                        const double *wl, double *lay, int nch, std::string &err)
{
    scat::Model m;
    if (!scat_model_view(dsm, pk, dbase, m, err)) return false;
    if (nch != 4 + 3 * scat::nterm(m)) { err = "scatter model: nch is not 4 + 3 x the model's terms"; return false; }
    launch_scatter(st, m, npoint, wl, lay, nch);
    return true;
}

}  // namespace sbd

extern "C" int sbd_scatter_blocks_host(const sbd_scat_model *sm, int32_t npoint, const double *wl, int32_t nch, double *lay_out)
{
    if (!sm || !wl || !lay_out || !sm->tables || npoint < 0) return SBD_E_INVALID;
    sbd::GasTablesPacked pk;
    std::string err;
    if (!pk.parse(sm->tables, sm->tables_bytes, err)) return SBD_E_INVALID;
    sbd::scat::Model m;
    if (!sbd::scat_model_view(sm, pk, pk.d.data(), m, err)) return SBD_E_INVALID;
    if (nch != 4 + 3 * sbd::scat::nterm(m)) return SBD_E_INVALID;
    for (int32_t p = 0; p < npoint; ++p) sbd::scat::point_block(m, wl[p], lay_out + (size_t)p * nch * m.nz, (size_t)m.nz, 1);
    return SBD_OK;
}
