// Issue cost of the instructions band_rows_kernel is made of (gfx950): cycles per instruction of one wave alone on
// its SIMD and of two waves sharing it.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, long long *cyc, int lane_sel, int iters)
{
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, m = a0 * 0.5, b0 = a0, b1 = a1, b2 = a0, b3 = a1, b4 = a0, b5 = a1; int x0 = 1, x1 = 2;
    int P = __builtin_amdgcn_readfirstlane(lane_sel);
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // readlane, readlane, fmac (sgpr operand), pipelined on two pairs
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_readlane_b32 s92, %[l0], %[P]\n v_readlane_b32 s93, %[h0], %[P]\n v_fmac_f64_e32 %[a2], s[94:95], %[m]\n"
                             "v_readlane_b32 s94, %[l1], %[P]\n v_readlane_b32 s95, %[h1], %[P]\n v_fmac_f64_e32 %[a3], s[92:93], %[m]\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3) : [l0] "v"(__double2loint(a0)), [h0] "v"(__double2hiint(a0)), [l1] "v"(__double2loint(a1)), [h1] "v"(__double2hiint(a1)), [P] "s"(P), [m] "v"(m) : "s92", "s93", "s94", "s95");
        } else if (MODE == 1) { // fmac with VGPR operands only
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_fmac_f64_e32 %[a2], %[a0], %[m]\n v_fmac_f64_e32 %[a3], %[a1], %[m]\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3) : [a0] "v"(a0), [a1] "v"(a1), [m] "v"(m));
        } else if (MODE == 2) { // readlanes only
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_readlane_b32 s92, %[l0], %[P]\n v_readlane_b32 s93, %[h0], %[P]\n v_readlane_b32 s94, %[l1], %[P]\n v_readlane_b32 s95, %[h1], %[P]\n"
                             :: [l0] "v"(__double2loint(a0)), [h0] "v"(__double2hiint(a0)), [l1] "v"(__double2loint(a1)), [h1] "v"(__double2hiint(a1)), [P] "s"(P) : "s92", "s93", "s94", "s95");
        } else if (MODE == 3) { // fmac with an SGPR pair operand, no readlane
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_fmac_f64_e32 %[a2], s[94:95], %[m]\n v_fmac_f64_e32 %[a3], s[92:93], %[m]\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3) : [m] "v"(m) : "s92", "s93", "s94", "s95");
        } else if (MODE == 4) { // DPP fmac, row_newbcast
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_fmac_f64_dpp %[a2], %[a0], %[m] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %[a3], %[a1], %[m] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3) : [a0] "v"(a0), [a1] "v"(a1), [m] "v"(m));
        } else if (MODE == 5) { // v_mov_b64
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_mov_b64_e32 %[a2], %[a0]\n v_mov_b64_e32 %[a3], %[a1]\n" : [a2] "+v"(a2), [a3] "+v"(a3) : [a0] "v"(a0), [a1] "v"(a1));
        } else if (MODE == 6) { // one readlane per fmac (32-bit half only): what a packed broadcast would cost
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_readlane_b32 s92, %[l0], %[P]\n v_fmac_f64_e32 %[a2], s[94:95], %[m]\n v_readlane_b32 s94, %[l1], %[P]\n v_fmac_f64_e32 %[a3], s[92:93], %[m]\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3) : [l0] "v"(__double2loint(a0)), [h0] "v"(__double2hiint(a0)), [l1] "v"(__double2loint(a1)), [h1] "v"(__double2hiint(a1)), [P] "s"(P), [m] "v"(m) : "s92", "s93", "s94", "s95");
        } else if (MODE == 8) { // readlanes to 16 distinct SGPRs
#pragma unroll
            for (int r = 0; r < REP / 8; ++r)
                asm volatile("v_readlane_b32 s80, %[l0], %[P]\n v_readlane_b32 s81, %[h0], %[P]\n v_readlane_b32 s82, %[l1], %[P]\n v_readlane_b32 s83, %[h1], %[P]\n"
                             "v_readlane_b32 s84, %[l0], %[P]\n v_readlane_b32 s85, %[h0], %[P]\n v_readlane_b32 s86, %[l1], %[P]\n v_readlane_b32 s87, %[h1], %[P]\n"
                             "v_readlane_b32 s88, %[l0], %[P]\n v_readlane_b32 s89, %[h0], %[P]\n v_readlane_b32 s90, %[l1], %[P]\n v_readlane_b32 s91, %[h1], %[P]\n"
                             "v_readlane_b32 s92, %[l0], %[P]\n v_readlane_b32 s93, %[h0], %[P]\n v_readlane_b32 s94, %[l1], %[P]\n v_readlane_b32 s95, %[h1], %[P]\n"
                             :: [l0] "v"(__double2loint(a0)), [h0] "v"(__double2hiint(a0)), [l1] "v"(__double2loint(a1)), [h1] "v"(__double2hiint(a1)), [P] "s"(P)
                             : "s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92", "s93", "s94", "s95");
        } else if (MODE == 9) { // readfirstlane
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_readfirstlane_b32 s92, %[l0]\n v_readfirstlane_b32 s93, %[h0]\n v_readfirstlane_b32 s94, %[l1]\n v_readfirstlane_b32 s95, %[h1]\n"
                             :: [l0] "v"(__double2loint(a0)), [h0] "v"(__double2hiint(a0)), [l1] "v"(__double2loint(a1)), [h1] "v"(__double2hiint(a1)), [P] "s"(P) : "s92", "s93", "s94", "s95");
        } else if (MODE == 10) { // readlane, constant lane
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_readlane_b32 s92, %[l0], 5\n v_readlane_b32 s93, %[h0], 5\n v_readlane_b32 s94, %[l1], 5\n v_readlane_b32 s95, %[h1], 5\n"
                             :: [l0] "v"(__double2loint(a0)), [h0] "v"(__double2hiint(a0)), [l1] "v"(__double2loint(a1)), [h1] "v"(__double2hiint(a1)), [P] "s"(P) : "s92", "s93", "s94", "s95");
        } else if (MODE == 11) { // 8 independent fmac chains (vgpr)
#pragma unroll
            for (int r = 0; r < REP / 8; ++r)
                asm volatile("v_fmac_f64_e32 %[a2], %[a0], %[m]\n v_fmac_f64_e32 %[a3], %[a1], %[m]\n v_fmac_f64_e32 %[b0], %[a0], %[m]\n v_fmac_f64_e32 %[b1], %[a1], %[m]\n"
                             "v_fmac_f64_e32 %[b2], %[a0], %[m]\n v_fmac_f64_e32 %[b3], %[a1], %[m]\n v_fmac_f64_e32 %[b4], %[a0], %[m]\n v_fmac_f64_e32 %[b5], %[a1], %[m]\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3), [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2), [b3] "+v"(b3), [b4] "+v"(b4), [b5] "+v"(b5) : [a0] "v"(a0), [a1] "v"(a1), [m] "v"(m));
        } else if (MODE == 12) { // 8 independent dpp fmac chains
#pragma unroll
            for (int r = 0; r < REP / 8; ++r)
                asm volatile("v_fmac_f64_dpp %[a2], %[a0], %[m] row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %[a3], %[a1], %[m] row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %[b0], %[a0], %[m] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %[b1], %[a1], %[m] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %[b2], %[a0], %[m] row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %[b3], %[a1], %[m] row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %[b4], %[a0], %[m] row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %[b5], %[a1], %[m] row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3), [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2), [b3] "+v"(b3), [b4] "+v"(b4), [b5] "+v"(b5) : [a0] "v"(a0), [a1] "v"(a1), [m] "v"(m));
        } else if (MODE == 13) { // v_mov_b32 dpp row_bcast15 / bcast31 (wave-level broadcast steps)
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_mov_b32_dpp %[x0], %[l0] row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 0\n v_mov_b32_dpp %[x1], %[l1] row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 0\n"
                             : [x0] "+v"(x0), [x1] "+v"(x1) : [l0] "v"(__double2loint(a0)), [l1] "v"(__double2loint(a1)));
        } else if (MODE == 14) { // the group sequence of band_rows_kernel: EXEC narrowed, 32 readfirstlane, EXEC back, 16 fmac
#pragma unroll
            for (int r = 0; r < REP / 16; ++r)
                asm volatile("s_mov_b64 exec, %[mask]\n"
                             "v_readfirstlane_b32 s60, %[l0]\n v_readfirstlane_b32 s61, %[h0]\n"
                             "v_readfirstlane_b32 s62, %[l1]\n v_readfirstlane_b32 s63, %[h1]\n"
                             "v_readfirstlane_b32 s64, %[l0]\n v_readfirstlane_b32 s65, %[h0]\n"
                             "v_readfirstlane_b32 s66, %[l1]\n v_readfirstlane_b32 s67, %[h1]\n"
                             "v_readfirstlane_b32 s68, %[l0]\n v_readfirstlane_b32 s69, %[h0]\n"
                             "v_readfirstlane_b32 s70, %[l1]\n v_readfirstlane_b32 s71, %[h1]\n"
                             "v_readfirstlane_b32 s72, %[l0]\n v_readfirstlane_b32 s73, %[h0]\n"
                             "v_readfirstlane_b32 s74, %[l1]\n v_readfirstlane_b32 s75, %[h1]\n"
                             "v_readfirstlane_b32 s76, %[l0]\n v_readfirstlane_b32 s77, %[h0]\n"
                             "v_readfirstlane_b32 s78, %[l1]\n v_readfirstlane_b32 s79, %[h1]\n"
                             "v_readfirstlane_b32 s80, %[l0]\n v_readfirstlane_b32 s81, %[h0]\n"
                             "v_readfirstlane_b32 s82, %[l1]\n v_readfirstlane_b32 s83, %[h1]\n"
                             "v_readfirstlane_b32 s84, %[l0]\n v_readfirstlane_b32 s85, %[h0]\n"
                             "v_readfirstlane_b32 s86, %[l1]\n v_readfirstlane_b32 s87, %[h1]\n"
                             "v_readfirstlane_b32 s88, %[l0]\n v_readfirstlane_b32 s89, %[h0]\n"
                             "v_readfirstlane_b32 s90, %[l1]\n v_readfirstlane_b32 s91, %[h1]\n"
                             "s_mov_b64 exec, -1\n"
                             "v_fmac_f64_e32 %[a2], s[60:61], %[m]\n"
                             "v_fmac_f64_e32 %[a3], s[62:63], %[m]\n"
                             "v_fmac_f64_e32 %[b0], s[64:65], %[m]\n"
                             "v_fmac_f64_e32 %[b1], s[66:67], %[m]\n"
                             "v_fmac_f64_e32 %[b2], s[68:69], %[m]\n"
                             "v_fmac_f64_e32 %[b3], s[70:71], %[m]\n"
                             "v_fmac_f64_e32 %[b4], s[72:73], %[m]\n"
                             "v_fmac_f64_e32 %[b5], s[74:75], %[m]\n"
                             "v_fmac_f64_e32 %[a2], s[76:77], %[m]\n"
                             "v_fmac_f64_e32 %[a3], s[78:79], %[m]\n"
                             "v_fmac_f64_e32 %[b0], s[80:81], %[m]\n"
                             "v_fmac_f64_e32 %[b1], s[82:83], %[m]\n"
                             "v_fmac_f64_e32 %[b2], s[84:85], %[m]\n"
                             "v_fmac_f64_e32 %[b3], s[86:87], %[m]\n"
                             "v_fmac_f64_e32 %[b4], s[88:89], %[m]\n"
                             "v_fmac_f64_e32 %[b5], s[90:91], %[m]\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3), [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2), [b3] "+v"(b3), [b4] "+v"(b4), [b5] "+v"(b5)
                             : [l0] "v"(__double2loint(a0)), [h0] "v"(__double2hiint(a0)), [l1] "v"(__double2loint(a1)), [h1] "v"(__double2hiint(a1)), [mask] "s"(1ull << P), [m] "v"(m)
                             : "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91");
        } else if (MODE == 15) { // 2 readlane (lane select in m0) + fmac
            asm volatile("s_mov_b32 m0, %0" :: "s"(P) : "m0");
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_readlane_b32 s92, %[l0], m0\n v_readlane_b32 s93, %[h0], m0\n v_fmac_f64_e32 %[a2], s[94:95], %[m]\n"
                             "v_readlane_b32 s94, %[l1], m0\n v_readlane_b32 s95, %[h1], m0\n v_fmac_f64_e32 %[a3], s[92:93], %[m]\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3) : [l0] "v"(__double2loint(a0)), [h0] "v"(__double2hiint(a0)), [l1] "v"(__double2loint(a1)), [h1] "v"(__double2hiint(a1)), [m] "v"(m) : "s92", "s93", "s94", "s95");
        } else if (MODE == 16) { // 32 readfirstlane + 16 fmac without touching EXEC
#pragma unroll
            for (int r = 0; r < REP / 16; ++r)
                asm volatile("s_nop 0\n"
                             "v_readfirstlane_b32 s60, %[l0]\n v_readfirstlane_b32 s61, %[h0]\n"
                             "v_readfirstlane_b32 s62, %[l1]\n v_readfirstlane_b32 s63, %[h1]\n"
                             "v_readfirstlane_b32 s64, %[l0]\n v_readfirstlane_b32 s65, %[h0]\n"
                             "v_readfirstlane_b32 s66, %[l1]\n v_readfirstlane_b32 s67, %[h1]\n"
                             "v_readfirstlane_b32 s68, %[l0]\n v_readfirstlane_b32 s69, %[h0]\n"
                             "v_readfirstlane_b32 s70, %[l1]\n v_readfirstlane_b32 s71, %[h1]\n"
                             "v_readfirstlane_b32 s72, %[l0]\n v_readfirstlane_b32 s73, %[h0]\n"
                             "v_readfirstlane_b32 s74, %[l1]\n v_readfirstlane_b32 s75, %[h1]\n"
                             "v_readfirstlane_b32 s76, %[l0]\n v_readfirstlane_b32 s77, %[h0]\n"
                             "v_readfirstlane_b32 s78, %[l1]\n v_readfirstlane_b32 s79, %[h1]\n"
                             "v_readfirstlane_b32 s80, %[l0]\n v_readfirstlane_b32 s81, %[h0]\n"
                             "v_readfirstlane_b32 s82, %[l1]\n v_readfirstlane_b32 s83, %[h1]\n"
                             "v_readfirstlane_b32 s84, %[l0]\n v_readfirstlane_b32 s85, %[h0]\n"
                             "v_readfirstlane_b32 s86, %[l1]\n v_readfirstlane_b32 s87, %[h1]\n"
                             "v_readfirstlane_b32 s88, %[l0]\n v_readfirstlane_b32 s89, %[h0]\n"
                             "v_readfirstlane_b32 s90, %[l1]\n v_readfirstlane_b32 s91, %[h1]\n"
                             "s_nop 0\n"
                             "v_fmac_f64_e32 %[a2], s[60:61], %[m]\n"
                             "v_fmac_f64_e32 %[a3], s[62:63], %[m]\n"
                             "v_fmac_f64_e32 %[b0], s[64:65], %[m]\n"
                             "v_fmac_f64_e32 %[b1], s[66:67], %[m]\n"
                             "v_fmac_f64_e32 %[b2], s[68:69], %[m]\n"
                             "v_fmac_f64_e32 %[b3], s[70:71], %[m]\n"
                             "v_fmac_f64_e32 %[b4], s[72:73], %[m]\n"
                             "v_fmac_f64_e32 %[b5], s[74:75], %[m]\n"
                             "v_fmac_f64_e32 %[a2], s[76:77], %[m]\n"
                             "v_fmac_f64_e32 %[a3], s[78:79], %[m]\n"
                             "v_fmac_f64_e32 %[b0], s[80:81], %[m]\n"
                             "v_fmac_f64_e32 %[b1], s[82:83], %[m]\n"
                             "v_fmac_f64_e32 %[b2], s[84:85], %[m]\n"
                             "v_fmac_f64_e32 %[b3], s[86:87], %[m]\n"
                             "v_fmac_f64_e32 %[b4], s[88:89], %[m]\n"
                             "v_fmac_f64_e32 %[b5], s[90:91], %[m]\n"
                             : [a2] "+v"(a2), [a3] "+v"(a3), [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2), [b3] "+v"(b3), [b4] "+v"(b4), [b5] "+v"(b5)
                             : [l0] "v"(__double2loint(a0)), [h0] "v"(__double2hiint(a0)), [l1] "v"(__double2loint(a1)), [h1] "v"(__double2hiint(a1)), [mask] "s"(1ull << P), [m] "v"(m)
                             : "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91");
        } else if (MODE == 7) { // v_max_u32 dpp chain step + s_nop as in the pivot search
#pragma unroll
            for (int r = 0; r < REP / 2; ++r)
                asm volatile("v_max_f64 %[a2], %[a0], %[a2]\n v_max_f64 %[a3], %[a1], %[a3]\n" : [a2] "+v"(a2), [a3] "+v"(a3) : [a0] "v"(a0), [a1] "v"(a1));
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a2 + a3 + b0 + b1 + b2 + b3 + b4 + b5 + x0 + x1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out; long long *cyc;
    const int nb = 2048;
    hipMalloc(&out, sizeof(double) * 64 * nb); hipMemset(out, 0, sizeof(double) * 64 * nb);
    hipMalloc(&cyc, sizeof(long long) * nb);
    const char *names[] = {"2 readlane + fmac(sgpr)", "fmac(vgpr)", "readlane only (x2)", "fmac(sgpr)", "fmac dpp row_newbcast", "v_mov_b64", "1 readlane + fmac(sgpr)", "v_max_f64", "readlane x2, 16 distinct sgprs", "readfirstlane x2", "readlane x2 constant lane", "fmac(vgpr) 8 chains", "fmac dpp 8 chains", "2 v_mov_b32_dpp bcast + nops", "group: exec, 32 rfl, exec, 16 fmac (per column)", "2 readlane(m0) + fmac", "32 rfl + 16 fmac, no exec (per column)"};
    const int per[] = {3, 1, 2, 1, 1, 1, 2, 1, 2, 2, 2, 1, 1, 2, 3, 3, 3};
    for (int waves = 1; waves <= 2; ++waves)
        for (int mode = 0; mode < 17; ++mode) {
            const int iters = 2000, grid = 256 * 4 * waves;   // one (two) wave(s) per SIMD
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 8: hipLaunchKernelGGL(k<8>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 9: hipLaunchKernelGGL(k<9>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 10: hipLaunchKernelGGL(k<10>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 11: hipLaunchKernelGGL(k<11>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 12: hipLaunchKernelGGL(k<12>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 13: hipLaunchKernelGGL(k<13>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 14: hipLaunchKernelGGL(k<14>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 15: hipLaunchKernelGGL(k<15>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                case 16: hipLaunchKernelGGL(k<16>, dim3(grid), dim3(64), 0, 0, out, cyc, 5, iters); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double groups = (double)iters * REP;          // column groups per wave
            // wall time per group per SIMD, in ns; at ~2.4 GHz
            printf("waves/SIMD %d  %-26s  %.2f ns per group per wave-slot (%.1f cycles at 2.4 GHz; %d instr)\n", waves, names[mode],
                   ms * 1e6 / groups, ms * 1e6 / groups * 2.4, per[mode]);
        }
    return 0;
}
