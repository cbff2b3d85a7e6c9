// Microbenchmark (round 4): VALU price per window row of the candidate band1 inner loops, 2 waves per SIMD.
//  mode 0: v_fmac_f64_dpp row_newbcast (multiplier from a lane of the row)      -- today's FMA
//  mode 1: 2 x v_readlane_b32 + v_fmac_f64 with the multiplier in SGPRs          -- no LDS transposition
//  mode 2: mode 1 + pivot keys (v_and_or_b32 per row, v_max3_u32 per two rows)
//  mode 3: mode 0 + v_max3_f32 on leading words per two rows
// build: hipcc -O3 --offload-arch=gfx950 -o tools/microbench/bin/valu_row_cost tools/microbench/valu_row_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int R = 32;
template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, int iters, unsigned long long *cyc)
{
    const int lane = threadIdx.x;
    double a[R];
#pragma unroll
    for (int p = 0; p < R; ++p) a[p] = 1.0 + 1e-3 * (lane + p);
    double tp = 1e-9 * lane, m = 0.5 + lane;
    unsigned key = 0; int mx = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int J = it & 31;
        asm volatile("" : "+s"(J));
#pragma unroll
        for (int p = 0; p < R; ++p) {
            if (MODE == 0 || MODE == 3) {
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[p]) : "v"(m), "v"(tp));
                if (MODE == 3 && (p & 1))
                    asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mx) : "v"(__double2hiint(a[p ^ 1 ? p - 1 : p])), "v"(__double2hiint(a[(p + 8) % R])));
            } else {
                int lo, hi;
                asm volatile("v_readlane_b32 %0, %2, %4\n\tv_readlane_b32 %1, %3, %4"
                             : "=s"(lo), "=s"(hi) : "v"(__double2loint(a[p])), "v"(__double2hiint(a[p])), "s"(J));
                const double sm = __hiloint2double(hi, lo);
                asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[p]) : "s"(sm), "v"(tp));
                if (MODE == 2) {
                    unsigned kk;
                    asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(kk) : "v"(__double2hiint(a[(p + 8) % R])), "s"(0x7fffffc0), "v"(63 - p));
                    if (p & 1) asm volatile("v_max3_u32 %0, %1, %2, %0" : "+v"(key) : "v"(kk), "v"(kk));
                }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    double s = 0;
#pragma unroll
    for (int p = 0; p < R; ++p) s += a[p];
    out[blockIdx.x * 64 + lane] = s + key + mx;
}
template <int MODE>
void run(const char *name, int wavesPerSimd)
{
    const int nb = 256 * 4 * wavesPerSimd, iters = 4000;
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, nb * 64 * 8); (void)hipMalloc(&cyc, nb * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<nb, 64>>>(out, 10, cyc);
    (void)hipEventRecord(e0);
    k<MODE><<<nb, 64>>>(out, iters, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nb); (void)hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= nb;
    printf("%-52s waves/SIMD %d: %.3f ms; per row: %.2f clocks in the wave, %.2f ns of SIMD time\n", name, wavesPerSimd, ms,
           avg / (iters * (double)R), ms * 1e6 / (iters * (double)R) / wavesPerSimd);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main()
{
    for (int w : {1, 2, 3}) {
        run<0>("DPP FMA", w);
        run<3>("DPP FMA + max3_f32 per two rows", w);
        run<1>("2 readlane + FMA(SGPR)", w);
        run<2>("2 readlane + FMA(SGPR) + key (and_or, max3/2)", w);
    }
    return 0;
}
