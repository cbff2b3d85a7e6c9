// band LU, LDS window (NSTR > 20, or SBD_BAND_LDS=1)
#include "sbd_launch.hpp"
#include "sbd_band.hpp"
namespace sbd {
#define SBD_NN_CASES(M) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20)
hipError_t prepare_band_lds(int nn, int lds)
{
#define SBD_C(NNv) if (nn == NNv) return raise_lds((const void *)band_kernel<NNv, false>, lds);
    SBD_NN_CASES(SBD_C)
#undef SBD_C
    return hipSuccess;
}
void launch_band_lds(int nn, unsigned grid, int lds, hipStream_t st, const Params &P)
{
#define SBD_C(NNv) if (nn == NNv) hipLaunchKernelGGL((band_kernel<NNv, false>), dim3(grid), dim3(64), lds, st, P);
    SBD_NN_CASES(SBD_C)
#undef SBD_C
}
}
