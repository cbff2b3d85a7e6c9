// band LU, a row per lane (NSTR 34..40)
#include "sbd_launch.hpp"
#include "sbd_bandr.hpp"
namespace sbd {
#define SBD_NN_CASES(M) M(17) M(18) M(19) M(20)
bool has_band_rows(int nn) { return nn >= 17 && nn <= 20; }
int band_rows_lds_bytes(int nn) { return (int)sizeof(double) * BandRowsLds(2 * nn, nn).total; }
void launch_band_rows(int nn, unsigned grid, hipStream_t st, const Params &P, bool fused)
{
#define SBD_C(NNv)                                                                                                             \
    if (nn == NNv) {                                                                                                           \
        if (fused) hipLaunchKernelGGL((band_rows_kernel<NNv, true>), dim3(grid), dim3(64), band_rows_lds_bytes(nn), st, P);     \
        else hipLaunchKernelGGL((band_rows_kernel<NNv, false>), dim3(grid), dim3(64), band_rows_lds_bytes(nn), st, P);          \
    }
    SBD_NN_CASES(SBD_C)
#undef SBD_C
}
}
