// band LU, register window, one system per wave (NSTR <= 20)
#include "sbd_launch.hpp"
#include "sbd_band.hpp"
namespace sbd {
#define SBD_NN_CASES(M) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10)
void launch_band_reg(int nn, unsigned grid, int lds, hipStream_t st, const Params &P)
{
#define SBD_C(NNv) if (nn == NNv) hipLaunchKernelGGL((band_kernel<NNv, true>), dim3(grid), dim3(64), lds, st, P);
    SBD_NN_CASES(SBD_C)
#undef SBD_C
}
}
