// setup, user-angle intensities, azimuth sum
#include "sbd_launch.hpp"
#include "sbd_setup.hpp"
#include "sbd_usrint.hpp"
#include "sbd_intcor.hpp"
namespace sbd {
void launch_setup(unsigned grid, hipStream_t st, const Params &P) { hipLaunchKernelGGL(setup_kernel, dim3(grid), dim3(64), 0, st, P); }
void launch_usrint(unsigned grid, int lds, hipStream_t st, const Params &P) { hipLaunchKernelGGL(usrint_kernel, dim3(grid), dim3(64), lds, st, P); }
void launch_cmpint(unsigned grid, hipStream_t st, const Params &P) { hipLaunchKernelGGL(cmpint_kernel, dim3(grid), dim3(64), 0, st, P); }
void launch_intcor(unsigned grid, hipStream_t st, const Params &P, int naz_run)
{
    hipLaunchKernelGGL(intcor_kernel, dim3(grid), dim3(256), sizeof(double) * 2 * kIntcorPairs * P.L, st, P, naz_run);
}
void launch_azimuth(unsigned grid, hipStream_t st, const Params &P, int naz_run) { hipLaunchKernelGGL(azimuth_kernel, dim3(grid), dim3(256), 0, st, P, naz_run); }
}
