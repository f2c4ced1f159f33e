/*
 * fr_device.hip -- context, launch logic and the render entry points of the C ABI
 * (include/fractalrenderer_amd.h).  The kernels are in fr_kernels.hip.h.
 *
 * Replaces, for the hot path only:
 *   ComputeEffectManager::dispatch          src/compute_effect_manager.h:435-468
 *   VulkanEngine::render_animation_frame    src/vk_engine.cpp:1181-1418 (render + readback part)
 * There is no CPU fallback in this file: without a HIP device every call fails.
 */
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fr_kernels.hip.h"

using namespace fr;

struct fr_ctx {
    int device;
    int compute_units;
    hipStream_t stream;
    hipEvent_t ev_begin, ev_end;
    bool have_timing;
    uint32_t* d_queue;          /* kShards heads, 128 B apart */
    uint32_t tune_wg_per_cu;    /* 0 = automatic */
    uint32_t tune_run_max;      /* 0 = automatic */
    uint32_t tune_shape;        /* 0 = automatic, else FPW_LOG2 (3, 4, 6) */
    void* scratch;              /* device staging for FR_MEM_HOST outputs */
    size_t scratch_bytes;
};

#define FR_HIP_TRY(expr)                                                               \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fr_set_error(FR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int fr_ctx_create(int device_ordinal, fr_ctx** out)
{
    if (!out) return fr_set_error(FR_ERR_INVALID_ARG, "fr_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fr_set_error(FR_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= ndev)
        return fr_set_error(FR_ERR_NO_DEVICE, "device ordinal %d out of range [0,%d)", device_ordinal, ndev);
    FR_HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    FR_HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));

    fr_ctx* c = (fr_ctx*)calloc(1, sizeof(fr_ctx));
    if (!c) return fr_set_error(FR_ERR_NOMEM, "out of host memory");
    c->device = device_ordinal;
    c->compute_units = prop.multiProcessorCount;
    hipError_t e2;
    if ((e2 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e2 = hipEventCreate(&c->ev_begin)) != hipSuccess ||
        (e2 = hipEventCreate(&c->ev_end)) != hipSuccess ||
        (e2 = hipMalloc((void**)&c->d_queue, kShards * kShardStrideWords * sizeof(uint32_t))) != hipSuccess) {
        free(c);
        return fr_set_error(FR_ERR_HIP, "context setup failed: %s", hipGetErrorString(e2));
    }
    *out = c;
    return FR_OK;
}

extern "C" void fr_ctx_destroy(fr_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->scratch) (void)hipFree(c->scratch);
    (void)hipFree(c->d_queue);
    (void)hipEventDestroy(c->ev_begin);
    (void)hipEventDestroy(c->ev_end);
    (void)hipStreamDestroy(c->stream);
    free(c);
}

extern "C" int fr_ctx_compute_units(fr_ctx* c)
{
    if (!c) return fr_set_error(FR_ERR_INVALID_ARG, "ctx is NULL");
    return c->compute_units;
}

extern "C" int fr_ctx_set_tuning(fr_ctx* c, uint32_t workgroups_per_cu, uint32_t subtiles_per_dequeue)
{
    if (!c) return fr_set_error(FR_ERR_INVALID_ARG, "ctx is NULL");
    if (workgroups_per_cu > 8) return fr_set_error(FR_ERR_INVALID_ARG, "workgroups_per_cu must be <= 8");
    /* the top byte of subtiles_per_dequeue selects the sub-tile shape for experiments:
     * 0 automatic, 3: 8x8, 4: 16x4, 6: 64x1 */
    c->tune_wg_per_cu = workgroups_per_cu;
    c->tune_shape = subtiles_per_dequeue >> 24;
    c->tune_run_max = subtiles_per_dequeue & 0xFFFFFFu;
    if (c->tune_shape != 0 && c->tune_shape != 3 && c->tune_shape != 4 && c->tune_shape != 6)
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown sub-tile shape %u", c->tune_shape);
    return FR_OK;
}

extern "C" float fr_ctx_last_kernel_ms(fr_ctx* c)
{
    if (!c || !c->have_timing) return -1.0f;
    if (hipSetDevice(c->device) != hipSuccess) return -1.0f;
    if (hipEventSynchronize(c->ev_end) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) != hipSuccess) return -1.0f;
    return ms;
}

/* ---- launch ---------------------------------------------------------------------------------- */

template <typename T, int FRACTAL, bool EFFECTS>
static hipError_t launch_shape(int shape, dim3 grid, hipStream_t s, const LaunchArgs& a)
{
    switch (shape) {
    case 6: hipLaunchKernelGGL((escape_kernel<T, FRACTAL, 6, EFFECTS>), grid, dim3(kBlockThreads), 0, s, a); break;
    case 4: hipLaunchKernelGGL((escape_kernel<T, FRACTAL, 4, EFFECTS>), grid, dim3(kBlockThreads), 0, s, a); break;
    default: hipLaunchKernelGGL((escape_kernel<T, FRACTAL, 3, EFFECTS>), grid, dim3(kBlockThreads), 0, s, a); break;
    }
    return hipGetLastError();
}

static int enqueue_render(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_shard* shard,
                          float* rgba, void* nu, int32_t* iter, hipStream_t stream)
{
    fr_shard whole = {0u, 1u, H};
    const fr_shard* sh = shard ? shard : &whole;
    fr_shard norm = *sh;
    if (norm.nparts == 0) norm.nparts = 1;
    if (norm.rows_per_strip == 0) norm.rows_per_strip = (norm.nparts == 1) ? H : 1;
    if (norm.part >= norm.nparts)
        return fr_set_error(FR_ERR_INVALID_ARG, "shard part %u >= nparts %u", norm.part, norm.nparts);
    const uint32_t rows_local = fr_shard_rows(&norm, H);
    if (rows_local == 0) return FR_OK;           /* this part owns no rows */

    const bool julia = p->fractal_type == FR_FRACTAL_JULIA;
    const bool f64 = p->precision == FR_PRECISION_F64;
    const bool effects = !julia && (p->orbit_trap_enabled || p->stripe_enabled || p->interior_style == 2);

    LaunchArgs a;
    memset(&a, 0, sizeof(a));
    a.center_x = p->center_x; a.center_y = p->center_y; a.zoom = p->zoom;
    a.julia_cx = p->julia_c_real; a.julia_cy = p->julia_c_imag;
    a.bailout = p->bailout;
    a.log_bailout = f64 ? log((double)p->bailout) : (double)logf(p->bailout);
    a.max_iter = p->max_iterations;
    a.W = (int32_t)W; a.H = (int32_t)H;
    a.rows_local = (int32_t)rows_local;
    a.part = (int32_t)norm.part; a.nparts = (int32_t)norm.nparts; a.rows_per_strip = (int32_t)norm.rows_per_strip;
    a.aa = p->antialiasing_samples;
    a.palette_mode = p->palette_mode;
    a.color_offset = p->color_offset; a.color_scale = p->color_scale;
    a.interior_style = p->interior_style;
    a.trap_enabled = p->orbit_trap_enabled; a.trap_radius = p->orbit_trap_radius;
    a.stripe_enabled = p->stripe_enabled; a.stripe_density = p->stripe_density;
    a.brightness = p->color_brightness; a.saturation = p->color_saturation; a.contrast = p->color_contrast;
    a.flags = p->flags;
    a.rgba = reinterpret_cast<float4*>(rgba);
    a.nu = nu; a.iter = iter;
    a.queue = c->d_queue;
    fr_palette_table_build(julia ? 1 : 0, p->palette_mode, &a.pal);

    /* escape is absorbing (see escape_run): bailout^2 in [4.5, 1e12], and for Julia |c| <= bailout;
     * Mandelbrot lanes with |c| > bailout retire at i = 0 inside the first, tested block */
    {
        const double B2 = f64 ? (double)p->bailout * (double)p->bailout
                              : (double)(p->bailout * p->bailout);
        const double c2 = a.julia_cx * a.julia_cx + a.julia_cy * a.julia_cy;
        a.fast_ok = (B2 >= 4.5 && B2 <= 1e12 && (!julia || c2 <= B2)) ? 1 : 0;
    }

    /* sub-tile shape and the tile queue */
    const int shape = c->tune_shape ? (int)c->tune_shape : 3;
    const uint32_t fpw = 1u << shape, fph = 64u >> shape;
    a.nsx = (W + fpw - 1) / fpw;
    const uint32_t nsy = (rows_local + fph - 1) / fph;
    a.n_sub = a.nsx * nsy;
    const uint32_t nblk = (a.n_sub + kShardBlock - 1) / kShardBlock;
    for (uint32_t k = 0; k < (uint32_t)kShards; ++k)
        a.shard_len[k] = ((nblk + kShards - 1 - k) / kShards) * kShardBlock;

    uint32_t wg_per_cu = c->tune_wg_per_cu ? c->tune_wg_per_cu : 2u;
    uint32_t grid = (uint32_t)c->compute_units * wg_per_cu;
    const uint32_t waves_needed = a.n_sub;                 /* never more waves than sub-tiles */
    const uint32_t max_grid = (waves_needed + 3) / 4;
    if (grid > max_grid) grid = max_grid < 1 ? 1 : max_grid;
    /* guided run length: remaining / (8 * waves per shard), clamped to [1, run_max] */
    uint32_t waves = grid * 4u, shift = 0;
    while ((1u << shift) < waves) ++shift;
    a.run_shift = shift;
    a.run_max = c->tune_run_max ? c->tune_run_max : 16u;

    FR_HIP_TRY(hipMemsetAsync(c->d_queue, 0, kShards * kShardStrideWords * sizeof(uint32_t), stream));
    FR_HIP_TRY(hipEventRecord(c->ev_begin, stream));
    hipError_t e;
    if (julia) {
        e = f64 ? launch_shape<double, 1, false>(shape, dim3(grid), stream, a)
                : launch_shape<float, 1, false>(shape, dim3(grid), stream, a);
    } else if (effects) {
        e = f64 ? launch_shape<double, 0, true>(shape, dim3(grid), stream, a)
                : launch_shape<float, 0, true>(shape, dim3(grid), stream, a);
    } else {
        e = f64 ? launch_shape<double, 0, false>(shape, dim3(grid), stream, a)
                : launch_shape<float, 0, false>(shape, dim3(grid), stream, a);
    }
    if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    FR_HIP_TRY(hipEventRecord(c->ev_end, stream));
    c->have_timing = true;
    return FR_OK;
}

static int check_common(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_output* out)
{
    if (!c) return fr_set_error(FR_ERR_INVALID_ARG, "ctx is NULL");
    if (!p || !out) return fr_set_error(FR_ERR_INVALID_ARG, "params/out is NULL");
    if (!out->rgba && !out->nu && !out->iter)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_output has no plane to write");
    int st = fr_params_validate(p, W, H);
    if (st != FR_OK) return st;
    return FR_OK;
}

extern "C" int fr_render_shard_async(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H,
                                     const fr_shard* shard, const fr_output* out, void* hip_stream)
{
    int st = check_common(c, p, W, H, out);
    if (st != FR_OK) return st;
    if (out->memory != FR_MEM_DEVICE)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_render_shard_async needs FR_MEM_DEVICE outputs");
    FR_HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return enqueue_render(c, p, W, H, shard, out->rgba, out->nu, out->iter, s);
}

extern "C" int fr_render_shard(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H,
                               const fr_shard* shard, const fr_output* out)
{
    int st = check_common(c, p, W, H, out);
    if (st != FR_OK) return st;
    FR_HIP_TRY(hipSetDevice(c->device));

    if (out->memory == FR_MEM_DEVICE) {
        st = enqueue_render(c, p, W, H, shard, out->rgba, out->nu, out->iter, c->stream);
        if (st != FR_OK) return st;
        FR_HIP_TRY(hipStreamSynchronize(c->stream));
        return FR_OK;
    }
    if (out->memory != FR_MEM_HOST)
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown fr_output.memory %d", out->memory);

    /* host outputs: stage through device scratch owned by the context (PCIe-inclusive path) */
    fr_shard whole = {0u, 1u, H};
    const fr_shard* sh = shard ? shard : &whole;
    fr_shard norm = *sh;
    if (norm.nparts == 0) norm.nparts = 1;
    if (norm.rows_per_strip == 0) norm.rows_per_strip = (norm.nparts == 1) ? H : 1;
    const size_t npx = (size_t)fr_shard_rows(&norm, H) * W;
    if (npx == 0) return FR_OK;
    const size_t nu_bytes = p->precision == FR_PRECISION_F64 ? 8 : 4;
    const size_t off_nu = npx * 16, off_iter = off_nu + npx * 8, need = off_iter + npx * 4;
    if (need > c->scratch_bytes) {
        if (c->scratch) { (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
        FR_HIP_TRY(hipMalloc(&c->scratch, need));
        c->scratch_bytes = need;
    }
    char* base = (char*)c->scratch;
    float* d_rgba = out->rgba ? (float*)base : nullptr;
    void* d_nu = out->nu ? (void*)(base + off_nu) : nullptr;
    int32_t* d_iter = out->iter ? (int32_t*)(base + off_iter) : nullptr;
    st = enqueue_render(c, p, W, H, &norm, d_rgba, d_nu, d_iter, c->stream);
    if (st != FR_OK) return st;
    if (out->rgba) FR_HIP_TRY(hipMemcpyAsync(out->rgba, d_rgba, npx * 16, hipMemcpyDeviceToHost, c->stream));
    if (out->nu) FR_HIP_TRY(hipMemcpyAsync(out->nu, d_nu, npx * nu_bytes, hipMemcpyDeviceToHost, c->stream));
    if (out->iter) FR_HIP_TRY(hipMemcpyAsync(out->iter, d_iter, npx * 4, hipMemcpyDeviceToHost, c->stream));
    FR_HIP_TRY(hipStreamSynchronize(c->stream));
    return FR_OK;
}

extern "C" int fr_render(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_output* out)
{
    return fr_render_shard(c, p, W, H, nullptr, out);
}

/* ---- 8-bit export ------------------------------------------------------------------------------ */
extern "C" int fr_export_rgb8(fr_ctx* c, const float* rgba, uint32_t W, uint32_t H,
                              uint8_t* rgb8, int32_t memory, int32_t through_half)
{
    if (!c || !rgba || !rgb8 || W == 0 || H == 0)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_export_rgb8: bad argument");
    FR_HIP_TRY(hipSetDevice(c->device));
    const size_t npx = (size_t)W * H;
    const float4* d_in = reinterpret_cast<const float4*>(rgba);
    uint8_t* d_out = rgb8;
    if (memory == FR_MEM_HOST) {
        const size_t need = npx * 16 + npx * 3;
        if (need > c->scratch_bytes) {
            if (c->scratch) { (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
            FR_HIP_TRY(hipMalloc(&c->scratch, need));
            c->scratch_bytes = need;
        }
        FR_HIP_TRY(hipMemcpyAsync(c->scratch, rgba, npx * 16, hipMemcpyHostToDevice, c->stream));
        d_in = reinterpret_cast<const float4*>(c->scratch);
        d_out = (uint8_t*)c->scratch + npx * 16;
    } else if (memory != FR_MEM_DEVICE) {
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown memory kind %d", memory);
    }
    size_t blocks = (npx + kBlockThreads - 1) / kBlockThreads;
    const size_t cap = (size_t)c->compute_units * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(export_rgb8_kernel, dim3((uint32_t)blocks), dim3(kBlockThreads), 0, c->stream,
                       d_in, d_out, (int)W, (int)H, (int)through_half);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "export launch failed: %s", hipGetErrorString(e));
    if (memory == FR_MEM_HOST)
        FR_HIP_TRY(hipMemcpyAsync(rgb8, d_out, npx * 3, hipMemcpyDeviceToHost, c->stream));
    FR_HIP_TRY(hipStreamSynchronize(c->stream));
    return FR_OK;
}
