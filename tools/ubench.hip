// Micro-benchmark: measured fp64 / fp32 VALU issue rates on gfx950, the ceilings the escape-time
// kernel is priced against.  Each wave runs CH independent dependent-chains of one instruction kind.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench.hip -o build/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int KIND, int CH>
__global__ void __launch_bounds__(256) k_f64(double* out, int iters, double a, double b)
{
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = a + threadIdx.x * 1e-9 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (KIND == 0) x[c] = __builtin_fma(x[c], a, b);
                else if (KIND == 1) x[c] = x[c] * a;
                else if (KIND == 2) x[c] = x[c] + b;
                else { // the escape-time fast-path step on (X, Yd, x2, y2d) packed in 4 chains
                }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    if (s == 123.456) out[0] = s;
}

// the real inner step: 6 ops, two 3-deep chains
__global__ void __launch_bounds__(256) k_step(double* out, int iters, double cx, double cyd)
{
    double X = 0.001 * threadIdx.x, Yd = 0.002, x2 = X * X, y2d = Yd * Yd;
    cx += 1e-7 * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double p = X * Yd;
            const double t = __builtin_fma(-0.25, y2d, x2);
            X = t + cx;
            Yd = __builtin_fma(2.0, p, cyd);
            x2 = X * X;
            y2d = Yd * Yd;
        }
    }
    if (x2 + y2d == 123.456) out[0] = X;
}

template <int KIND, int CH>
__global__ void __launch_bounds__(256) k_f32(float* out, int iters, float a, float b)
{
    float x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = a + threadIdx.x * 1e-6f + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (KIND == 0) x[c] = __builtin_fmaf(x[c], a, b);
                else if (KIND == 1) x[c] = x[c] * a;
                else x[c] = x[c] + b;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    if (s == 123.456f) out[0] = s;
}

typedef float float2v __attribute__((ext_vector_type(2)));
template <int CH>
__global__ void __launch_bounds__(256) k_pk32(float* out, int iters, float a, float b)
{
    float2v x[CH];
    const float2v av = {a, a}, bv = {b, b};
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = float2v{a + threadIdx.x * 1e-6f + c, a - c};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = x[c] * av + bv;      // contract off: v_pk_mul_f32 + v_pk_add_f32
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c].x + x[c].y;
    if (s == 123.456f) out[0] = s;
}

template <typename F>
static double time_ms(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    double* d; hipMalloc(&d, 1024);
    const int iters = 20000;
    const char* kn[3] = {"v_fma_f64", "v_mul_f64", "v_add_f64"};
    for (int wps = 1; wps <= 8; wps *= 2) {              // waves per SIMD = blocks per CU (256 thr = 4 waves)
        const int grid = cus * wps;
#define RUN64(KIND, CH) { double ms = time_ms([&] { hipLaunchKernelGGL((k_f64<KIND, CH>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9); }); \
        double ops = (double)grid * 256 * iters * 8 * CH; \
        printf("%-10s waves/SIMD %d chains %d: %8.3f ms  %7.2f T lane-op/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", kn[KIND], wps, CH, ms, ops / ms / 1e9, \
               (double)cus * 4 * 2.4e9 * ms * 1e-3 / (ops / 64)); }
        RUN64(0, 1) RUN64(0, 2) RUN64(0, 4) RUN64(1, 2) RUN64(1, 4) RUN64(2, 2) RUN64(2, 4)
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_step, dim3(grid), dim3(256), 0, 0, d, iters, -0.1, 0.2); });
            double its = (double)grid * 256 * iters * 16;
            printf("escape step waves/SIMD %d: %8.3f ms  %7.3f T iter/s  (%.2f cyc/iter/wave/SIMD @2.4GHz; 6 ops)\n", wps, ms, its / ms / 1e9,
                   (double)cus * 4 * 2.4e9 * ms * 1e-3 / (its / 64));
        }
    }
    const char* kn32[3] = {"v_fma_f32", "v_mul_f32", "v_add_f32"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int grid = cus * wps;
#define RUN32(KIND, CH) { double ms = time_ms([&] { hipLaunchKernelGGL((k_f32<KIND, CH>), dim3(grid), dim3(256), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f); }); \
        double ops = (double)grid * 256 * iters * 8 * CH; \
        printf("%-10s waves/SIMD %d chains %d: %8.3f ms  %7.2f T lane-op/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", kn32[KIND], wps, CH, ms, ops / ms / 1e9, \
               (double)cus * 4 * 2.4e9 * ms * 1e-3 / (ops / 64)); }
        RUN32(0, 4) RUN32(1, 4) RUN32(2, 4)
        {
            double ms = time_ms([&] { hipLaunchKernelGGL((k_pk32<4>), dim3(grid), dim3(256), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f); });
            double ops = (double)grid * 256 * iters * 8 * 4 * 2 * 2;     // 2 instrs x 2 values
            printf("v_pk_mul+add_f32 waves/SIMD %d chains 4: %8.3f ms  %7.2f T value-op/s\n", wps, ms, ops / ms / 1e9);
        }
    }
    return 0;
}
