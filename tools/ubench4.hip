// Micro-benchmark 4: what a TESTED update costs per wave and SIMD -- the loop shape of the lane pool / tile pass:
// NV vector instructions (two dependent chains), one vector compare, a branch on VCC, a scalar countdown and the loop
// branch -- with 1..8 waves per SIMD, and with the scalar part varied (none / countdown only / + never-taken branch /
// + extra SALU).  Answers: do SALU and branches add to the VALU time of a loop, and how much do more waves hide?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench4.hip -o build/ubench4
#include <hip/hip_runtime.h>
#include <cstdio>

#define V6 "v_fmac_f32 %0, %2, %1\n v_mul_f32 %1, %0, %3\n v_fmac_f32 %0, %2, %1\n v_mul_f32 %1, %0, %3\n v_fmac_f32 %0, %2, %1\n v_mul_f32 %1, %0, %3\n"

// KIND 0: 6 VALU + cmp, loop control only (s_sub, s_cmp, s_cbranch)            -> 7 VALU + 3 SALU
// KIND 1: + s_cbranch_vccnz (never taken) after the compare                      -> 7 VALU + 4 SALU
// KIND 2: + 2 more SALU (the round-1 pool loop: 6 SALU)                          -> 7 VALU + 6 SALU
// KIND 3: loop unrolled x2 (14 VALU, 2 vcc branches, one countdown)              -> 7 VALU + 2.5 SALU per update
// KIND 4: no compare, no vcc branch: 6 VALU + 3 SALU
// KIND 5: 6 VALU, 16 updates per loop trip (the unchecked block)                 -> 6 VALU + 0.2 SALU
// KIND 6: FOUR tested updates per trip, each compare writes its own SGPR pair, nothing reads a mask until the trip's end:
//         3 s_or + s_cmp + one (never taken) branch + countdown                   -> 7 VALU + 1.75 SALU per update
// KIND 7: the same with EIGHT updates per trip                                    -> 7 VALU + 1.4 SALU per update
// KIND 8: four per trip, the masks accumulated in a VGPR (v_cmp + v_cndmask + v_or), ONE compare + branch per trip
template <int KIND>
__global__ void __launch_bounds__(256) k_loop(float* out, int iters, float a)
{
    float x = a + threadIdx.x * 1e-6f, y = x * 0.5f, c = 0.999f, d = 1.0001f, thr = __builtin_inff();
    int n = iters;
    if (KIND == 0)
        asm volatile("1:\n" V6 "v_cmp_gt_f32 vcc, %0, %4\n s_sub_u32 %5, %5, 1\n s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1b\n"
                     : "+v"(x), "+v"(y) : "v"(c), "v"(d), "v"(thr), "s"(n) : "vcc", "scc");
    else if (KIND == 1)
        asm volatile("1:\n" V6 "v_cmp_gt_f32 vcc, %0, %4\n s_cbranch_vccnz 2f\n s_sub_u32 %5, %5, 1\n s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1b\n2:\n"
                     : "+v"(x), "+v"(y) : "v"(c), "v"(d), "v"(thr), "s"(n) : "vcc", "scc");
    else if (KIND == 2)
        asm volatile("1:\n" V6 "v_cmp_gt_f32 vcc, %0, %4\n s_cbranch_vccnz 2f\n s_sub_u32 %5, %5, 1\n s_add_u32 s20, s20, 1\n s_mov_b32 s21, s20\n s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1b\n2:\n"
                     : "+v"(x), "+v"(y) : "v"(c), "v"(d), "v"(thr), "s"(n) : "vcc", "scc", "s20", "s21");
    else if (KIND == 3)
        asm volatile("1:\n" V6 "v_cmp_gt_f32 vcc, %0, %4\n s_cbranch_vccnz 2f\n" V6 "v_cmp_gt_f32 vcc, %0, %4\n s_cbranch_vccnz 2f\n s_sub_u32 %5, %5, 2\n s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1b\n2:\n"
                     : "+v"(x), "+v"(y) : "v"(c), "v"(d), "v"(thr), "s"(n) : "vcc", "scc");
    else if (KIND == 4)
        asm volatile("1:\n" V6 "s_sub_u32 %5, %5, 1\n s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1b\n"
                     : "+v"(x), "+v"(y) : "v"(c), "v"(d), "v"(thr), "s"(n) : "vcc", "scc");
    else if (KIND == 5)
        asm volatile("1:\n" V6 V6 V6 V6 V6 V6 V6 V6 V6 V6 V6 V6 V6 V6 V6 V6 "s_sub_u32 %5, %5, 16\n s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1b\n"
                     : "+v"(x), "+v"(y) : "v"(c), "v"(d), "v"(thr), "s"(n) : "vcc", "scc");
    else if (KIND == 6)
        asm volatile("1:\n" V6 "v_cmp_gt_f32_e64 s[20:21], %0, %4\n" V6 "v_cmp_gt_f32_e64 s[22:23], %0, %4\n" V6 "v_cmp_gt_f32_e64 s[24:25], %0, %4\n"
                     V6 "v_cmp_gt_f32_e64 s[26:27], %0, %4\n"
                     "s_or_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[24:25], s[24:25], s[26:27]\n s_or_b64 s[20:21], s[20:21], s[24:25]\n"
                     "s_cmp_lg_u64 s[20:21], 0\n s_cbranch_scc1 2f\n s_sub_u32 %5, %5, 4\n s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1b\n2:\n"
                     : "+v"(x), "+v"(y) : "v"(c), "v"(d), "v"(thr), "s"(n) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    else if (KIND == 7)
        asm volatile("1:\n" V6 "v_cmp_gt_f32_e64 s[20:21], %0, %4\n" V6 "v_cmp_gt_f32_e64 s[22:23], %0, %4\n" V6 "v_cmp_gt_f32_e64 s[24:25], %0, %4\n"
                     V6 "v_cmp_gt_f32_e64 s[26:27], %0, %4\n" V6 "v_cmp_gt_f32_e64 s[28:29], %0, %4\n" V6 "v_cmp_gt_f32_e64 s[30:31], %0, %4\n"
                     V6 "v_cmp_gt_f32_e64 s[32:33], %0, %4\n" V6 "v_cmp_gt_f32_e64 s[34:35], %0, %4\n"
                     "s_or_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[24:25], s[24:25], s[26:27]\n s_or_b64 s[28:29], s[28:29], s[30:31]\n s_or_b64 s[32:33], s[32:33], s[34:35]\n"
                     "s_or_b64 s[20:21], s[20:21], s[24:25]\n s_or_b64 s[28:29], s[28:29], s[32:33]\n s_or_b64 s[20:21], s[20:21], s[28:29]\n"
                     "s_cmp_lg_u64 s[20:21], 0\n s_cbranch_scc1 2f\n s_sub_u32 %5, %5, 8\n s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1b\n2:\n"
                     : "+v"(x), "+v"(y) : "v"(c), "v"(d), "v"(thr), "s"(n) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27",
                       "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
    else {
        float acc = 0.0f, one = 1.0f;
        asm volatile("1:\n" V6 "v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %7, 0, %6, vcc\n" V6 "v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                     V6 "v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %7, %7, %6, vcc\n" V6 "v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                     "v_cmp_neq_f32 vcc, 0, %7\n s_cbranch_vccnz 2f\n s_sub_u32 %5, %5, 4\n s_cmp_lg_u32 %5, 0\n s_cbranch_scc1 1b\n2:\n"
                     : "+v"(x), "+v"(y) : "v"(c), "v"(d), "v"(thr), "s"(n), "v"(one), "v"(acc) : "vcc", "scc");
    }
    if (x + y == 123.456f) out[0] = x;
}

template <typename K>
static double time_ms(K kernel, int grid, float* d, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d, iters, 1.5f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d, iters, 1.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* d; (void)hipMalloc(&d, 1024);
    const int iters = 1 << 16;
    printf("cycles per update per SIMD (2.4 GHz nominal), fp32 tested-update loop shapes; waves/SIMD 1 2 4 6 8\n");
    const char* names[9] = {"7 VALU + 3 SALU (countdown)", "7 VALU + 4 SALU (+ vcc branch)", "7 VALU + 6 SALU", "unrolled x2: 7 VALU + 2.5 SALU",
                            "6 VALU + 3 SALU (no compare)", "6 VALU, 16 updates per trip", "4 tested per trip, masks in SGPRs",
                            "8 tested per trip, masks in SGPRs", "4 tested per trip, mask in a VGPR"};
#define ROW(KIND) { printf("%-34s", names[KIND]); for (int w : {1, 2, 4, 6, 8}) { double ms = time_ms(k_loop<KIND>, cus * w, d, iters); \
        printf(" %7.1f", ms * 1e-3 * 2.4e9 / ((double)iters * w)); } printf("\n"); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8)
    return 0;
}
