// Micro-benchmark 3: what the chip sustains for the tile pass's OUTPUT alone -- 16 B per pixel of a 4096^2 (8192^2) frame,
// written once: (a) linear, one float4 per thread; (b) in the tile pass's pattern: a wave stores an 8x8-pixel sub-tile =
// 8 segments of 128 B one frame row (64 KB) apart; (c) 16x4 and (d) 64x1 sub-tiles.  Persistent grid of CUs x 5 workgroups
// walking the sub-tiles in row-major order, as the tile pass does.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench3.hip -o build/ubench3
#include <hip/hip_runtime.h>
#include <cstdio>
template <int FPW_LOG2>
__global__ void __launch_bounds__(256) k_tiles(float4* out, int W, int H, int nwaves)
{
    constexpr int FPW = 1 << FPW_LOG2, FPH = 64 / FPW;
    const int lane = threadIdx.x & 63, lx = lane & (FPW - 1), ly = lane >> FPW_LOG2;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nsx = W / FPW, ntiles = nsx * (H / FPH);
    for (int t = wave; t < ntiles; t += nwaves) {
        const int sty = t / nsx, stx = t - sty * nsx;
        const size_t pix = (size_t)(sty * FPH + ly) * W + stx * FPW + lx;
        out[pix] = make_float4((float)t, 1.0f, 2.0f, 1.0f);
    }
}
__global__ void __launch_bounds__(256) k_linear(float4* out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = make_float4((float)i, 1.0f, 2.0f, 1.0f);
}
int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int W : {4096, 8192}) {
        const int H = W; const size_t n = (size_t)W * H;
        float4* d; (void)hipMalloc(&d, n * 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        auto run = [&](const char* name, auto launch) {
            launch(); (void)hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 10; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
            printf("%dx%d %-28s %8.4f ms  %7.1f GB/s\n", W, H, name, best, n * 16 / best / 1e6);
        };
        for (int wg : {5, 8}) {
            const int grid = cus * wg, nw = grid * 4;
            printf("-- %d workgroups per CU\n", wg);
            run("linear float4/thread", [&] { hipLaunchKernelGGL(k_linear, dim3(grid), dim3(256), 0, 0, d, n); });
            run("8x8 sub-tiles", [&] { hipLaunchKernelGGL((k_tiles<3>), dim3(grid), dim3(256), 0, 0, d, W, H, nw); });
            run("16x4 sub-tiles", [&] { hipLaunchKernelGGL((k_tiles<4>), dim3(grid), dim3(256), 0, 0, d, W, H, nw); });
            run("64x1 sub-tiles", [&] { hipLaunchKernelGGL((k_tiles<6>), dim3(grid), dim3(256), 0, 0, d, W, H, nw); });
        }
        run("hipMemsetAsync", [&] { (void)hipMemsetAsync(d, 0, n * 16, 0); });
        (void)hipFree(d);
    }
    return 0;
}
