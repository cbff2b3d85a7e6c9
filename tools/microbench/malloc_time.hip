// hipMalloc / hipFree wall time against the size asked for (the engine's workspace is ONE allocation: 8.8 GB for the
// headline shape, 47 GB at NSTR 32 x 50 layers -- where does sbd_engine_create's time go?).  GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o bin/malloc_time malloc_time.hip && bin/malloc_time
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(char *p, size_t n, size_t stride) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }
int main()
{
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    hipInit(0);
    int n = 0;
    hipGetDeviceCount(&n);
    hipSetDevice(0);
    hipFree(nullptr);
    printf("runtime bring-up (hipInit + first hipFree): %.1f ms\n", std::chrono::duration<double, std::milli>(now() - t0).count());
    hipStream_t st;
    t0 = now();
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    printf("first stream: %.1f ms\n", std::chrono::duration<double, std::milli>(now() - t0).count());
    const double gb[] = {0.25, 1, 2, 4, 8, 9, 16, 24, 32, 47, 64, 8, 47};
    for (double g : gb) {
        const size_t bytes = (size_t)(g * (1ull << 30));
        char *p = nullptr;
        t0 = now();
        hipError_t e = hipMalloc(&p, bytes);
        const double tm = std::chrono::duration<double, std::milli>(now() - t0).count();
        if (e != hipSuccess) { printf("%6.2f GB: %s\n", g, hipGetErrorString(e)); continue; }
        t0 = now();
        hipLaunchKernelGGL(touch, dim3((unsigned)((bytes / (2u << 20)) / 256 + 1)), dim3(256), 0, st, p, bytes, (size_t)(2u << 20));
        hipStreamSynchronize(st);
        const double tt = std::chrono::duration<double, std::milli>(now() - t0).count();
        t0 = now();
        hipFree(p);
        const double tf = std::chrono::duration<double, std::milli>(now() - t0).count();
        printf("%6.2f GB: hipMalloc %8.2f ms  first touch (one byte per 2 MB) %8.2f ms  hipFree %8.2f ms\n", g, tm, tt, tf);
    }
    return 0;
}
