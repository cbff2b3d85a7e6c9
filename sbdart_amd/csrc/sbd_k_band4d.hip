// band LU, four systems per wave: the instantiation that records its pivot choices (tests only)
#include "sbd_launch.hpp"
#include "sbd_band4.hpp"
namespace sbd {
#ifndef SBD_BAND4_CASES
#define SBD_BAND4_CASES(M) M(2) M(3) M(4) M(5) M(6) M(7) M(8)
#endif
void launch_band4_pivdbg(int nn, unsigned grid, hipStream_t st, const Params &P)
{
#define SBD_C(NNv) if (nn == NNv) hipLaunchKernelGGL((band4_kernel<NNv, false, true>), dim3(grid), dim3(64), 0, st, P);
    SBD_BAND4_CASES(SBD_C)
#undef SBD_C
}
}
