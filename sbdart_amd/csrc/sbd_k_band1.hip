// band LU, one system per wave, block form, window in registers (16 < NSTR <= 32)
#include "sbd_launch.hpp"
#include "sbd_band1.hpp"
namespace sbd {
#define SBD_NN_CASES(M) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)
void launch_band1(int nn, unsigned grid, hipStream_t st, const Params &P, bool fused)
{
#define SBD_C(NNv) if (nn == NNv) { if (fused) hipLaunchKernelGGL((band1_kernel<NNv, true>), dim3(grid), dim3(64), kBand1LdsDoubles * sizeof(double), st, P); \
                                    else hipLaunchKernelGGL((band1_kernel<NNv, false>), dim3(grid), dim3(64), kBand1LdsDoubles * sizeof(double), st, P); }
    SBD_NN_CASES(SBD_C)
#undef SBD_C
}
}
