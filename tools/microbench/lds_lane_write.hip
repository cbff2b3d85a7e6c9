// Microbenchmark (round 4): what does a ONE-lane ds_write2_b64 cost the CU's LDS pipe, against a full-wave one?
// band1_kernel transposes its pivot column with 16-24 such writes per sub-step; 8 waves per CU do it at once.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_lane_write tools/microbench/lds_lane_write.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, int iters, unsigned long long *cyc)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    double a0 = lane * 1.5, a1 = lane * 2.5;
    unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) double *)sm;
    unsigned long long bit = 1ull << 5;
    if (MODE == 1) addr += lane * 16;                              // full wave, conflict-free (dump layout)
    if (MODE == 3) addr = (lane == 5) ? addr : addr + 512 + lane * 16;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0)        // masked: one lane
                asm volatile("s_mov_b64 exec, %3\n\tds_write2_b64 %0, %1, %2 offset0:0 offset1:1\n\ts_mov_b64 exec, -1"
                             :: "v"(addr), "v"(a0), "v"(a1), "s"(bit) : "memory");
            else if (MODE == 1 || MODE == 3)   // every lane writes (its own 16 bytes)
                asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" :: "v"(addr), "v"(a0), "v"(a1) : "memory");
            else if (MODE == 2)   // every lane, same address
                asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" :: "v"(addr), "v"(a0), "v"(a1) : "memory");
            else if (MODE == 4) { // 4 x readlane + nothing (the SGPR route's price per two rows)
                int s0, s1, s2, s3;
                asm volatile("v_readlane_b32 %0, %4, 5\n\tv_readlane_b32 %1, %5, 5\n\tv_readlane_b32 %2, %4, 6\n\tv_readlane_b32 %3, %5, 6"
                             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(__double2hiint(a0)), "v"(__double2loint(a0)));
                a1 += s0 + s1 + s2 + s3;
            }
            a0 += 1.0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = a0 + a1 + sm[lane];
}
template <int MODE>
void run(const char *name, int wavesPerCU)
{
    const int nb = 256 * wavesPerCU, iters = 2000;
    double *out; unsigned long long *cyc;
    hipMalloc(&out, nb * 64 * 8); hipMalloc(&cyc, nb * 8);
    const size_t lds = 160 * 1024 / wavesPerCU - 256;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nb, 64, lds>>>(out, 10, cyc);
    hipEventRecord(e0);
    k<MODE><<<nb, 64, lds>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nb); hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= nb;
    printf("%-34s waves/CU %d: %.3f ms, %.1f wave-clocks per write, %.1f ns per write per wave (x%d waves share the CU)\n",
           name, wavesPerCU, ms, avg / (iters * 16.0), ms * 1e6 / (iters * 16.0), wavesPerCU);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w : {1, 4, 8}) {
        run<0>("one lane (exec-masked) write2_b64", w);
        run<1>("full wave, own 16 B each", w);
        run<2>("full wave, same address", w);
        run<3>("full wave, lane 5 real + dump", w);
        run<4>("4 x v_readlane_b32", w);
    }
    return 0;
}
