// fast layer kernel (symmetrised eigenproblem, one-sided Jacobi), radiance mode = true
#include "sbd_launch.hpp"
#include "sbd_layer2.hpp"
namespace sbd {
hipError_t prepare_layer2_r(int nn, int lds)
{
#define SBD_C(NNv, Gv) if (nn == NNv) return raise_lds((const void *)layer_kernel2<NNv, Gv, true>, lds);
    SBD_L2_CASES(SBD_C)
#undef SBD_C
    return hipSuccess;
}
void launch_layer2_r(int nn, unsigned grid, int lds, hipStream_t st, const Params &P, int32_t *eigflag)
{
#define SBD_C(NNv, Gv) if (nn == NNv) hipLaunchKernelGGL((layer_kernel2<NNv, Gv, true>), dim3(grid), dim3(64), lds, st, P, eigflag);
    SBD_L2_CASES(SBD_C)
#undef SBD_C
}
}
#ifdef SBD_PHASE_TICKS
extern "C" int sbd_debug_layer2r_ticks(unsigned long long *out, int reset)
{
    static unsigned long long h[1024 * 16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(sbd::layer2_ticks), sizeof h) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out[i] = 0;
    for (int b = 0; b < 1024; ++b)
        for (int i = 0; i < 16; ++i) out[i] += h[b * 16 + i];
    if (reset) {
        for (auto &x : h) x = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(sbd::layer2_ticks), h, sizeof h) != hipSuccess) return 1;
    }
    return 0;
}
#endif
