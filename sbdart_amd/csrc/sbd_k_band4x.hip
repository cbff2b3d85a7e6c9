// band LU, four systems per wave, LINPACK's exact pivot rule (sbd_run_cfg::pivot_exact): fused, stored-factor and the
// instantiation that records its pivot choices
#include "sbd_launch.hpp"
#include "sbd_band4.hpp"
namespace sbd {
#ifndef SBD_BAND4_CASES
#define SBD_BAND4_CASES(M) M(2) M(3) M(4) M(5) M(6) M(7) M(8)
#endif
void launch_band4_exact(int nn, unsigned grid, hipStream_t st, const Params &P, bool fused, bool pivdbg)
{
    if (pivdbg) {
#define SBD_C(NNv) if (nn == NNv) hipLaunchKernelGGL((band4_kernel<NNv, false, true, true>), dim3(grid), dim3(64), 0, st, P);
        SBD_BAND4_CASES(SBD_C)
#undef SBD_C
        return;
    }
    if (fused) {
#define SBD_C(NNv) if constexpr (NNv >= 3) { if (nn == NNv) hipLaunchKernelGGL((band4_kernel<NNv, true, false, true>), dim3(grid), dim3(64), 0, st, P); }
        SBD_BAND4_CASES(SBD_C)
#undef SBD_C
        return;
    }
#define SBD_C(NNv) if (nn == NNv) hipLaunchKernelGGL((band4_kernel<NNv, false, false, true>), dim3(grid), dim3(64), 0, st, P);
    SBD_BAND4_CASES(SBD_C)
#undef SBD_C
}
}
