// back-substitution + fluxes, four systems per wave (NSTR <= 16)
#include "sbd_launch.hpp"
#include "sbd_solve4.hpp"
namespace sbd {
#ifndef SBD_BAND4_CASES
#define SBD_BAND4_CASES(M) M(2) M(3) M(4) M(5) M(6) M(7) M(8)
#endif
void launch_backsolve4(int nn, unsigned grid, hipStream_t st, const Params &P)
{
#define SBD_C(NNv) if (nn == NNv) hipLaunchKernelGGL((backsolve4_kernel<NNv>), dim3(grid), dim3(64), 0, st, P);
    SBD_BAND4_CASES(SBD_C)
#undef SBD_C
}
}
