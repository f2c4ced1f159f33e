// Micro-benchmark 2: issue cost of the NON-arithmetic-core VALU instructions the per-pixel code of the tile pass is made
// of (conversions, frexp, floor, transcendentals, selects, moves, lane reads), each against the plain v_fma_f64 / v_fma_f32
// slot.  One asm statement per instruction, 4 independent register sets per wave, 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o build/ubench2
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(S) S S S S S S S S

#define KERNEL_DD(NAME, INSTR)                                                                            \
    __global__ void __launch_bounds__(256) NAME(double* out, int iters, double a)                         \
    {                                                                                                     \
        double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y0, y1, y2, y3;         \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(INSTR " %0, %1" : "=v"(y0) : "v"(x0)); asm volatile(INSTR " %0, %1" : "=v"(y1) : "v"(x1)); \
                 asm volatile(INSTR " %0, %1" : "=v"(y2) : "v"(x2)); asm volatile(INSTR " %0, %1" : "=v"(y3) : "v"(x3));) \
        }                                                                                                 \
        if (y0 + y1 + y2 + y3 == 123.456) out[0] = y0;                                                    \
    }
#define KERNEL_ID(NAME, INSTR) /* int <- double */                                                        \
    __global__ void __launch_bounds__(256) NAME(double* out, int iters, double a)                         \
    {                                                                                                     \
        double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; int y0, y1, y2, y3;     \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(INSTR " %0, %1" : "=v"(y0) : "v"(x0)); asm volatile(INSTR " %0, %1" : "=v"(y1) : "v"(x1)); \
                 asm volatile(INSTR " %0, %1" : "=v"(y2) : "v"(x2)); asm volatile(INSTR " %0, %1" : "=v"(y3) : "v"(x3));) \
        }                                                                                                 \
        if (y0 + y1 + y2 + y3 == 123456) out[0] = y0;                                                     \
    }
#define KERNEL_DI(NAME, INSTR) /* double <- int/float (32-bit source) */                                  \
    __global__ void __launch_bounds__(256) NAME(double* out, int iters, double a)                         \
    {                                                                                                     \
        int x0 = threadIdx.x + (int)a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; double y0, y1, y2, y3;       \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(INSTR " %0, %1" : "=v"(y0) : "v"(x0)); asm volatile(INSTR " %0, %1" : "=v"(y1) : "v"(x1)); \
                 asm volatile(INSTR " %0, %1" : "=v"(y2) : "v"(x2)); asm volatile(INSTR " %0, %1" : "=v"(y3) : "v"(x3));) \
        }                                                                                                 \
        if (y0 + y1 + y2 + y3 == 123.456) out[0] = y0;                                                    \
    }
#define KERNEL_FF(NAME, INSTR) /* 32 <- 32 */                                                             \
    __global__ void __launch_bounds__(256) NAME(double* out, int iters, double a)                         \
    {                                                                                                     \
        float x0 = (float)a + threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y0, y1, y2, y3;  \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(INSTR " %0, %1" : "=v"(y0) : "v"(x0)); asm volatile(INSTR " %0, %1" : "=v"(y1) : "v"(x1)); \
                 asm volatile(INSTR " %0, %1" : "=v"(y2) : "v"(x2)); asm volatile(INSTR " %0, %1" : "=v"(y3) : "v"(x3));) \
        }                                                                                                 \
        if (y0 + y1 + y2 + y3 == 123.456f) out[0] = y0;                                                   \
    }
#define KERNEL_FFF(NAME, INSTR) /* 32 <- 32, 32 */                                                        \
    __global__ void __launch_bounds__(256) NAME(double* out, int iters, double a)                         \
    {                                                                                                     \
        float x0 = (float)a + threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y0, y1, y2, y3;  \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(INSTR " %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x1)); asm volatile(INSTR " %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(x2)); \
                 asm volatile(INSTR " %0, %1, %2" : "=v"(y2) : "v"(x2), "v"(x3)); asm volatile(INSTR " %0, %1, %2" : "=v"(y3) : "v"(x3), "v"(x0));) \
        }                                                                                                 \
        if (y0 + y1 + y2 + y3 == 123.456f) out[0] = y0;                                                   \
    }
#define KERNEL_DDD(NAME, INSTR) /* 64 <- 64, 64 */                                                        \
    __global__ void __launch_bounds__(256) NAME(double* out, int iters, double a)                         \
    {                                                                                                     \
        double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y0, y1, y2, y3;         \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(INSTR " %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x1)); asm volatile(INSTR " %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(x2)); \
                 asm volatile(INSTR " %0, %1, %2" : "=v"(y2) : "v"(x2), "v"(x3)); asm volatile(INSTR " %0, %1, %2" : "=v"(y3) : "v"(x3), "v"(x0));) \
        }                                                                                                 \
        if (y0 + y1 + y2 + y3 == 123.456) out[0] = y0;                                                    \
    }

KERNEL_DDD(k_mul_f64, "v_mul_f64")
KERNEL_DDD(k_max_f64, "v_max_f64")
KERNEL_DD(k_frexp_mant_f64, "v_frexp_mant_f64")
KERNEL_ID(k_frexp_exp_f64, "v_frexp_exp_i32_f64")
KERNEL_DD(k_floor_f64, "v_floor_f64")
KERNEL_DD(k_fract_f64, "v_fract_f64")
KERNEL_DD(k_rcp_f64, "v_rcp_f64")
KERNEL_DD(k_mov_b64, "v_mov_b64")
KERNEL_DI(k_cvt_f64_i32, "v_cvt_f64_i32")
KERNEL_DI(k_cvt_f64_f32, "v_cvt_f64_f32")
KERNEL_ID(k_cvt_f32_f64, "v_cvt_f32_f64")
KERNEL_FFF(k_mul_f32, "v_mul_f32")
KERNEL_FFF(k_and_b32, "v_and_b32")
KERNEL_FFF(k_lshlrev_b32, "v_lshlrev_b32")
KERNEL_FF(k_mov_b32, "v_mov_b32")
KERNEL_FF(k_log_f32, "v_log_f32")
KERNEL_FF(k_exp_f32, "v_exp_f32")
KERNEL_FF(k_rcp_f32, "v_rcp_f32")
KERNEL_FF(k_floor_f32, "v_floor_f32")
KERNEL_FF(k_cvt_f32_i32, "v_cvt_f32_i32")

/* v_cndmask_b32 (reads VCC), v_cmp_gt_f64 (writes VCC), v_readlane / v_writelane, v_pk_mul_f32 */
__global__ void __launch_bounds__(256) k_cndmask(double* out, int iters, double a)
{
    float x0 = (float)a + threadIdx.x * 1e-3f, x1 = x0 + 1, y0, y1, y2, y3;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y0) : "v"(x0), "v"(x1)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y1) : "v"(x1), "v"(x0));
             asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y2) : "v"(x0), "v"(x1)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y3) : "v"(x1), "v"(x0));)
    }
    if (y0 + y1 + y2 + y3 == 123.456f) out[0] = y0;
}
__global__ void __launch_bounds__(256) k_cmp_f64(double* out, int iters, double a)
{
    double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(x0), "v"(x1) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(x1), "v"(x0) : "vcc");
             asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(x0), "v"(x1) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(x1), "v"(x0) : "vcc");)
    }
    if (x0 == 123.456) out[0] = x0;
}
__global__ void __launch_bounds__(256) k_cmp_f32(double* out, int iters, double a)
{
    float x0 = (float)a + threadIdx.x * 1e-3f, x1 = x0 + 1;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x0), "v"(x1) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x1), "v"(x0) : "vcc");
             asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x0), "v"(x1) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x1), "v"(x0) : "vcc");)
    }
    if (x0 == 123.456f) out[0] = x0;
}
__global__ void __launch_bounds__(256) k_readlane(double* out, int iters, double a)
{
    int x0 = threadIdx.x + (int)a; int s0, s1, s2, s3;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(x0)); asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s1) : "v"(x0));
             asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s2) : "v"(x0)); asm volatile("v_readlane_b32 %0, %1, 9" : "=s"(s3) : "v"(x0));)
    }
    if (s0 + s1 + s2 + s3 == 123456) out[0] = s0;
}
__global__ void __launch_bounds__(256) k_pk_mul_f32(double* out, int iters, double a)
{
    double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1, y0, y1, y2, y3;      /* register pairs */
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x1)); asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(x0));
             asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y2) : "v"(x0), "v"(x1)); asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y3) : "v"(x1), "v"(x0));)
    }
    if (y0 + y1 + y2 + y3 == 123.456) out[0] = y0;
}
/* scalar: dependent s_add chain (one wave issues one SALU per issue slot) and 4 independent ones */
__global__ void __launch_bounds__(256) k_salu(double* out, int iters, double a)
{
    int s0 = iters, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc"); asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) :: "scc");
             asm volatile("s_add_u32 %0, %0, 5" : "+s"(s2) :: "scc"); asm volatile("s_add_u32 %0, %0, 7" : "+s"(s3) :: "scc");)
    }
    if (s0 + s1 + s2 + s3 == 123456) out[0] = s0;
}
/* mixed: 1 VALU (fp64 fma) + 1 SALU alternating within one wave: do they overlap across the waves of a SIMD? */
__global__ void __launch_bounds__(256) k_mix(double* out, int iters, double a)
{
    double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int s0 = iters, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x0)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc");
             asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x1)); asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) :: "scc");
             asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x2)); asm volatile("s_add_u32 %0, %0, 5" : "+s"(s2) :: "scc");
             asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x3)); asm volatile("s_add_u32 %0, %0, 7" : "+s"(s3) :: "scc");)
    }
    if (x0 + x1 + x2 + x3 == 123.456 || s0 + s1 + s2 + s3 == 123456) out[0] = x0;
}
/* mixed 32-bit: 1 v_mul_f32 + 1 SALU */
__global__ void __launch_bounds__(256) k_mix32(double* out, int iters, double a)
{
    float x0 = (float)a + threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int s0 = iters, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x0)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc");
             asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x1)); asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) :: "scc");
             asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x2)); asm volatile("s_add_u32 %0, %0, 5" : "+s"(s2) :: "scc");
             asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x3)); asm volatile("s_add_u32 %0, %0, 7" : "+s"(s3) :: "scc");)
    }
    if (x0 + x1 + x2 + x3 == 123.456f || s0 + s1 + s2 + s3 == 123456) out[0] = x0;
}
/* LDS broadcast-ish reads: ds_read_b128 at a per-lane index into a 2 KB table */
__global__ void __launch_bounds__(256) k_ds128(double* out, int iters, double a)
{
    __shared__ double tab[256];
    tab[threadIdx.x] = a + threadIdx.x;
    __syncthreads();
    unsigned addr = ((threadIdx.x * 37u) & 127u) * 16u;
    double acc = 0;
    for (int i = 0; i < iters; ++i) {
        double2 v0, v1, v2, v3;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(v1) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(v2) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(v3) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v0.x + v1.x + v2.x + v3.x;
    }
    if (acc == 123.456) out[0] = acc;
}


/* selects, more closely: mask in an SGPR pair (e64), a compare in front of every select, and the exec-masked move */
__global__ void __launch_bounds__(256) k_cndmask_e64(double* out, int iters, double a)
{
    float x0 = (float)a + threadIdx.x * 1e-3f, x1 = x0 + 1, y0, y1, y2, y3;
    unsigned long long m = 0x5555555555555555ull * (unsigned)iters;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(y0) : "v"(x0), "v"(x1), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(y1) : "v"(x1), "v"(x0), "s"(m));
             asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(y2) : "v"(x0), "v"(x1), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(y3) : "v"(x1), "v"(x0), "s"(m));)
    }
    if (y0 + y1 + y2 + y3 == 123.456f) out[0] = y0;
}
__global__ void __launch_bounds__(256) k_cmp_cndmask(double* out, int iters, double a)
{
    float x0 = (float)a + threadIdx.x * 1e-3f, x1 = x0 + 1, y0, y1, y2, y3;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_cmp_gt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y0) : "v"(x0), "v"(x1) : "vcc");
             asm volatile("v_cmp_gt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y1) : "v"(x1), "v"(x0) : "vcc");
             asm volatile("v_cmp_gt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y2) : "v"(x0), "v"(x1) : "vcc");
             asm volatile("v_cmp_gt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y3) : "v"(x1), "v"(x0) : "vcc");)
    }
    if (y0 + y1 + y2 + y3 == 123.456f) out[0] = y0;
}
__global__ void __launch_bounds__(256) k_max_f32(double* out, int iters, double a)
{
    float x0 = (float)a + threadIdx.x * 1e-3f, x1 = x0 + 1, y0, y1, y2, y3;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_max_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x1)); asm volatile("v_max_f32 %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(x0));
             asm volatile("v_med3_f32 %0, %1, %2, 1.0" : "=v"(y2) : "v"(x0), "v"(x1)); asm volatile("v_med3_f32 %0, %1, %2, 1.0" : "=v"(y3) : "v"(x1), "v"(x0));)
    }
    if (y0 + y1 + y2 + y3 == 123.456f) out[0] = y0;
}
__global__ void __launch_bounds__(256) k_execmov(double* out, int iters, double a)
{
    float x0 = (float)a + threadIdx.x * 1e-3f, y0 = x0, y1 = x0, y2 = x0, y3 = x0;
    unsigned long long m = 0x5555555555555555ull * (unsigned)iters, full = ~0ull;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("s_mov_b64 exec, %4\n v_mov_b32 %0, %5\n v_mov_b32 %1, %5\n v_mov_b32 %2, %5\n v_mov_b32 %3, %5\n s_mov_b64 exec, %6"
                          : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "s"(m), "v"(x0), "s"(full));)
    }
    if (y0 + y1 + y2 + y3 == 123.456f) out[0] = y0;
}
KERNEL_DDD(k_add_f64, "v_add_f64")
KERNEL_FFF(k_add_u32, "v_add_u32")
KERNEL_FFF(k_add_f32, "v_add_f32")
KERNEL_FFF(k_sub_f32, "v_sub_f32")
KERNEL_FFF(k_min_f32, "v_min_f32")
KERNEL_FFF(k_or_b32, "v_or_b32")
KERNEL_FFF(k_lshrrev_b32, "v_lshrrev_b32")
KERNEL_FFF(k_mul_lo_u32, "v_mul_lo_u32")
KERNEL_FFF(k_mul_u32_u24, "v_mul_u32_u24")
KERNEL_FF(k_cvt_f32_u32, "v_cvt_f32_u32")
KERNEL_FF(k_fract_f32, "v_fract_f32")
KERNEL_FF(k_sqrt_f32, "v_sqrt_f32")
__global__ void __launch_bounds__(256) k_bfe_u32(double* out, int iters, double a)
{
    int x0 = threadIdx.x + (int)a, x1 = x0 + 1, y0, y1, y2, y3;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_bfe_u32 %0, %1, 13, 7" : "=v"(y0) : "v"(x0)); asm volatile("v_bfe_u32 %0, %1, 13, 7" : "=v"(y1) : "v"(x1));
             asm volatile("v_and_or_b32 %0, %1, %2, %2" : "=v"(y2) : "v"(x0), "v"(x1)); asm volatile("v_and_or_b32 %0, %1, %2, %2" : "=v"(y3) : "v"(x1), "v"(x0));)
    }
    if (y0 + y1 + y2 + y3 == 123456) out[0] = y0;
}
__global__ void __launch_bounds__(256) k_fma_f32(double* out, int iters, double a)
{
    float x0 = (float)a + threadIdx.x * 1e-3f, x1 = x0 + 1, y0, y1, y2, y3;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(y0) : "v"(x0), "v"(x1)); asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(y1) : "v"(x1), "v"(x0));
             asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(y2) : "v"(x0), "v"(x1)); asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(y3) : "v"(x1), "v"(x0));)
    }
    if (y0 + y1 + y2 + y3 == 123.456f) out[0] = y0;
}

template <typename K>
static double time_ms(K kernel, int grid, double* d, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d, iters, 1.5);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d, iters, 1.5);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs; 8 waves/SIMD, 4 register sets per wave; cycles at 2.4 GHz nominal (the chip may hold less)\n", prop.name, cus);
    double* d; hipMalloc(&d, 1024);
    const int iters = 4000, grid = cus * 8;
#define RUN(K, NINSTR) { double ms = time_ms(K, grid, d, iters); double winstr = (double)grid * 4 * iters * (NINSTR); \
        printf("%-22s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", #K, ms, (double)cus * 4 * 2.4e9 * ms * 1e-3 / winstr); }
    RUN(k_mul_f64, 32) RUN(k_max_f64, 32) RUN(k_frexp_mant_f64, 32) RUN(k_frexp_exp_f64, 32) RUN(k_floor_f64, 32) RUN(k_fract_f64, 32)
    RUN(k_rcp_f64, 32) RUN(k_mov_b64, 32) RUN(k_cvt_f64_i32, 32) RUN(k_cvt_f64_f32, 32) RUN(k_cvt_f32_f64, 32)
    RUN(k_mul_f32, 32) RUN(k_and_b32, 32) RUN(k_lshlrev_b32, 32) RUN(k_mov_b32, 32) RUN(k_log_f32, 32) RUN(k_exp_f32, 32) RUN(k_rcp_f32, 32)
    RUN(k_floor_f32, 32) RUN(k_cvt_f32_i32, 32) RUN(k_cndmask, 32) RUN(k_cmp_f64, 32) RUN(k_cmp_f32, 32) RUN(k_readlane, 32) RUN(k_pk_mul_f32, 32)
    RUN(k_cndmask_e64, 32) RUN(k_cmp_cndmask, 32) RUN(k_max_f32, 32) RUN(k_execmov, 32) RUN(k_add_f64, 32) RUN(k_add_u32, 32) RUN(k_add_f32, 32) RUN(k_sub_f32, 32)
    RUN(k_min_f32, 32) RUN(k_or_b32, 32) RUN(k_lshrrev_b32, 32) RUN(k_mul_lo_u32, 32) RUN(k_mul_u32_u24, 32) RUN(k_cvt_f32_u32, 32) RUN(k_fract_f32, 32) RUN(k_sqrt_f32, 32) RUN(k_bfe_u32, 32) RUN(k_fma_f32, 32)
    RUN(k_salu, 32) RUN(k_mix, 64) RUN(k_mix32, 64) RUN(k_ds128, 4)
    return 0;
}
