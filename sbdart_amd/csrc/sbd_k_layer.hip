// reference-algorithm layer kernel (QR eigen-solver): fallback list / SBD_LAYER_V1
#include "sbd_launch.hpp"
#include "sbd_layer.hpp"
namespace sbd {
#define SBD_G_CASES(M) M(4) M(8) M(16) M(32) M(64)
hipError_t prepare_layer_v1(int G, int lds)
{
#define SBD_C(Gv) if (G == Gv) { const hipError_t r_ = raise_lds((const void *)layer_kernel<Gv, true>, lds); \
                                 return r_ != hipSuccess ? r_ : raise_lds((const void *)layer_kernel<Gv, false>, lds); }
    SBD_G_CASES(SBD_C)
#undef SBD_C
    return hipSuccess;
}
void launch_layer_v1(int G, unsigned grid, int lds, hipStream_t st, const Params &P, int32_t *only_flagged)
{
#define SBD_C(Gv) if (G == Gv) { if (only_flagged) hipLaunchKernelGGL((layer_kernel<Gv, true>), dim3(grid), dim3(64), lds, st, P, only_flagged); \
                                 else hipLaunchKernelGGL((layer_kernel<Gv, false>), dim3(grid), dim3(64), lds, st, P, only_flagged); }
    SBD_G_CASES(SBD_C)
#undef SBD_C
}
}
