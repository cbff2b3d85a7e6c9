// back-substitution + fluxes
#include "sbd_launch.hpp"
#include "sbd_solve.hpp"
#include "sbd_solve1.hpp"
namespace sbd {
#define SBD_NN_CASES(M) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20)
hipError_t prepare_backsolve(int nn, int lds)
{
#define SBD_C(NNv) if (nn == NNv) return raise_lds((const void *)backsolve_kernel<NNv>, lds);
    SBD_NN_CASES(SBD_C)
#undef SBD_C
    return hipSuccess;
}
// 16 < NSTR <= 32 (band1_kernel's factor): the row-oriented kernel, LDS for FLUXES' staging only
void launch_backsolve1(int nn, unsigned grid, hipStream_t st, const Params &P)
{
    const int lds = (int)sizeof(double) * 2 * 16 * 2 * nn;
#define SBD_C(NNv) if (nn == NNv) hipLaunchKernelGGL(backsolve1_kernel<NNv>, dim3(grid), dim3(64), lds, st, P);
    SBD_C(9) SBD_C(10) SBD_C(11) SBD_C(12) SBD_C(13) SBD_C(14) SBD_C(15) SBD_C(16)
#undef SBD_C
}
void launch_backsolve(int nn, unsigned grid, int lds, hipStream_t st, const Params &P)
{
#define SBD_C(NNv) if (nn == NNv) hipLaunchKernelGGL(backsolve_kernel<NNv>, dim3(grid), dim3(64), lds, st, P);
    SBD_NN_CASES(SBD_C)
#undef SBD_C
}
}
