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
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fr_kernels.hip.h"
#include "fr_tuning.h"

using namespace fr;

static constexpr int kMaxStages = 2;            /* tile pass (+ lane-pool pass) */
static constexpr size_t kStageWords = (size_t)2 * kMaxShards * kShardStrideWords;   /* one stage: its queue heads, then its stream counters */
static constexpr size_t kCtrlWords = (size_t)kMaxStages * kStageWords;
static constexpr size_t kReadyWord = kCtrlWords + (size_t)(kFeedbackShards + 1) * kShardStrideWords;   /* LaunchArgs::pro_ready */
static constexpr size_t kFeedbackWord = kCtrlWords;            /* behind the stages: Feedback::dev_flag (kFeedbackShards words, 128 B apart) */
static constexpr uint32_t kProbeEvery = 16;                     /* frames between two looks of a view that closed nothing */

struct fr_ctx {
    int device;
    int compute_units;
    hipStream_t stream;
    hipEvent_t ev_begin, ev_end;
    bool have_timing;           /* ev_begin / ev_end hold the most recent render (only recorded with "timing" on) */
    bool timing;                /* option "timing": record the event pair around every render (fr_ctx_last_kernel_ms).  Off by
                                 * default since 1.1: two timed event records cost a frame ~4.7 us (C2 0.6 %, C3 1.8 %, a 1080p
                                 * frame at max_iter 256 10 %: profiles/r04_timing_events.txt) */
    bool have_render;           /* a render has been enqueued: last_stream is where */
    hipStream_t last_stream;    /* the stream of the most recent render */
    hipEvent_t ev_order;        /* no timing: recorded on last_stream when somebody has to wait for that render */
    uint32_t* d_ctrl;           /* queue heads + stream counters of every stage (kCtrlWords) */
    void* frame_buf;            /* fr_render_frame_png: RGBA f32 frame + RGB8 */
    size_t frame_bytes;
    void* orbit_host;           /* Deep_Zoom: pinned staging (fp64 orbit + its float narrowing) */
    float* orbit_dev;           /* Deep_Zoom: reference orbit as float pairs */
    size_t orbit_cap;           /* capacity in scalars (2 per orbit point) */
    void* stream_buf;           /* survivor stream (tile pass -> lane pool) */
    size_t stream_bytes;
    uint32_t tune_pool_refill;  /* lane pool: idle lanes that trigger a refill (0 = 24) */
    int32_t tune_periodicity;   /* cycle closing: -1 off, 0 automatic (on, first window 128), else the first snapshot window in iterations */
    uint32_t tune_staging;      /* 0 = automatic, 1 = single pass, 3 = tile pass + lane-pool pass whatever max_iter is */
    uint32_t tune_stage_first;  /* iteration budget b0 of the tile pass (0 = automatic) */
    uint32_t tune_stream_run_max, tune_stream_run_min, tune_stream_wg_per_cu;
    size_t diag_stride;         /* words between the diag regions of consecutive stages */
    int last_stages;
    uint32_t tune_wg_per_cu;    /* 0 = automatic */
    uint32_t tune_run_max;      /* 0 = automatic */
    uint32_t tune_run_min;      /* 0 = automatic */
    int tune_shift_bias;        /* added to the guided-run shift */
    uint32_t tune_probes;       /* tile pass: shards a wave probes before exiting (0 = automatic) */
    uint32_t tune_stream_probes;/* same for the stream / lane-pool passes */
    uint32_t tune_stream_rotate;/* 0 automatic, 1 regions by XCD, 2 writers rotate over the regions */
    uint64_t* diag;             /* optional device buffer for per-wave timelines */
    uint32_t last_grid;
    uint32_t tune_shape;        /* 0 = automatic, else FPW_LOG2 (3, 4, 6) */
    void* scratch;              /* device staging for FR_MEM_HOST outputs */
    size_t scratch_bytes;
    uint32_t debug_region_blocks; /* tests only: cap the capacity of a survivor-stream region, to provoke an overflow */
    double2* log2_tab;          /* device copy of the log2 table of the fp64 smooth-count epilogue (log2_tab()) */
    float* export8_thr;         /* device: 256 x {t[b], t[b + 1]}, the byte thresholds of the 8-bit export (fr_export8_thresholds) */
    void* coord_buf;            /* lean tile pass: W + H coordinates of the frame being rendered (prepare_kernel) */
    size_t coord_bytes;
    uint32_t tune_tile_kernel;  /* 0 = automatic (the lean tile kernel where it applies), 1 = the general tile_kernel */
    uint32_t tune_stripes;      /* the Mandelbrot shader's effects (stripes, orbit trap, trap-coloured interior): 0 = automatic (lean tile
                                   pass + lane pool, kernel code FRACTAL = 3), 1 = the effects variant of the general tile kernel */
    uint32_t tune_ssaa_band;    /* staged SSAA: samples per band of a whole frame whose sample grid is larger (0 = automatic: 2^29) */
    uint32_t tune_ssaa;         /* SSAA: 0 = automatic, 1 = the sample loop of the general tile kernel, 2 = staged (sample grid
                                 * through tile pass + lane pool, then ssaa_reduce_kernel) wherever it applies */
    void* ssaa_buf;             /* staged SSAA: the sample planes (colour [+ nu] [+ iter]), grow-only */
    size_t ssaa_bytes;
    uint32_t tune_shards;       /* 0 = automatic, 8 or 64: queue shards / stream regions of a render */
    uint32_t tune_regions;      /* 0 = automatic (= shards), 8 or 64: regions of the survivor streams */
    uint32_t tune_tile_pixels;  /* lean tile kernel: sub-tiles (pixels per lane) per trip, 0 = automatic (2), 1 or 2 */
    uint32_t tune_tile_exit;    /* lean tile pass, staged: occupancy exit -- 0 = automatic, 1 = off, else the per-record cost of the
                                   lane pool in updates that the exit rule assumes (escape_run_lean) */
    uint32_t tune_tile_exit_from; /* ... and the updates a trip runs before it may leave (0 = automatic) */
    uint32_t tune_pool_items_per_wg; /* lane pool grid: at most one workgroup per this many sub-tiles of the frame (0 = 32) */
    uint32_t* overflow_host;    /* pinned, device-mapped word: a survivor stream ran out of blocks (see StreamRef::overflow) */
    uint32_t* overflow_dev;     /* the same word as the kernels address it */
    bool render_on_user_stream; /* the most recent render was enqueued on a caller's stream: ev_end orders the context's
                                 * own stream (exports, colorize) behind it */
    /* automatic cycle closing of the lane pool (pool_wants_cycle_closing) */
    uint32_t render_seq;        /* renders enqueued on this context */
    uint32_t prologue_epoch;    /* lean tile passes launched with the in-kernel prologue (lean_prologue): 28 bits */
    uint32_t tune_prepare;      /* 0 = automatic (the tile pass prepares its own control block and tables, except on a capturing
                                   stream), 1 = prepare_kernel in a launch of its own */
    uint64_t probe_key;         /* what the context renders (fractal, precision, max_iter, geometry, coarse view) */
    int probe_mode;             /* 0 LOOK: every render's pool looks; 1 SKIP: none does, skip_left to go; 2 WAIT: one look is in
                                 * flight (render probe_seq), nobody else looks until its verdict is back */
    uint32_t probe_first;       /* LOOK: first render of the run of looks; WAIT: the one look */
    uint32_t skip_left;
    int last_pool_closing;      /* the most recent render's lane pool looked for cycles (1) / did not (0) / there was none (-1) */
    struct DivCheck { bool valid, julia, f64, ok; uint32_t W, H; };
    DivCheck div_cache[8];      /* exact_division_ok() results */
    uint32_t div_next;
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
    /* log2 table: bin i of [0.5, 1) has midpoint m_i = 0.5 + (i + 0.5) / 256; entry = {y_i = RN(1 / m_i), -log2(y_i)}.
     * The logarithm is taken of the ROUNDED reciprocal (in 64-bit long double), so that m = (1 + r) / y_i holds for
     * the r the kernels compute and the table contributes no error of its own beyond its final rounding. */
    double tab[2 * kLog2Entries];
    for (int i = 0; i < kLog2Entries; ++i) {
        const double m = 0.5 + ((double)i + 0.5) / (2.0 * kLog2Entries);
        const double y = 1.0 / m;
        tab[2 * i] = y;
        tab[2 * i + 1] = (double)(-log2l((long double)y));
    }
    /* byte thresholds of the 8-bit export as the pairs the kernel reads: {t[b], t[b + 1]} */
    float t8[257], pairs[512];
    fr_export8_thresholds(t8);
    for (int b = 0; b < 256; ++b) { pairs[2 * b] = t8[b]; pairs[2 * b + 1] = t8[b + 1]; }
    hipError_t e2;
    if ((e2 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e2 = hipEventCreate(&c->ev_begin)) != hipSuccess ||
        (e2 = hipEventCreate(&c->ev_end)) != hipSuccess ||
        (e2 = hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming)) != hipSuccess ||
        (e2 = hipMalloc((void**)&c->d_ctrl, (kCtrlWords + (size_t)(kFeedbackShards + 2) * kShardStrideWords) * sizeof(uint32_t))) != hipSuccess ||
        (e2 = hipMemset(c->d_ctrl, 0, (kCtrlWords + (size_t)(kFeedbackShards + 2) * kShardStrideWords) * sizeof(uint32_t))) != hipSuccess ||
        (e2 = hipHostMalloc((void**)&c->overflow_host, 64, hipHostMallocMapped)) != hipSuccess ||
        (e2 = hipHostGetDevicePointer((void**)&c->overflow_dev, c->overflow_host, 0)) != hipSuccess ||
        (e2 = hipMalloc((void**)&c->log2_tab, sizeof(tab))) != hipSuccess ||
        (e2 = hipMemcpy(c->log2_tab, tab, sizeof(tab), hipMemcpyHostToDevice)) != hipSuccess ||
        (e2 = hipMalloc((void**)&c->export8_thr, sizeof(pairs))) != hipSuccess ||
        (e2 = hipMemcpy(c->export8_thr, pairs, sizeof(pairs), hipMemcpyHostToDevice)) != hipSuccess) {
        fr_ctx_destroy(c);                       /* releases whatever was created */
        return fr_set_error(FR_ERR_HIP, "context setup failed: %s", hipGetErrorString(e2));
    }
    c->overflow_host[0] = 0u;
    c->overflow_host[1] = 0u;                    /* the feedback word (Feedback::host_word) */
    *out = c;
    return FR_OK;
}

/* Also the unwinding path of a failed fr_ctx_create: every member may still be NULL. */
extern "C" void fr_ctx_destroy(fr_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();                /* renders of this context may have been enqueued on callers' streams */
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->d_ctrl) (void)hipFree(c->d_ctrl);
    if (c->overflow_host) (void)hipHostFree(c->overflow_host);
    if (c->log2_tab) (void)hipFree(c->log2_tab);
    if (c->export8_thr) (void)hipFree(c->export8_thr);
    if (c->coord_buf) (void)hipFree(c->coord_buf);
    if (c->ssaa_buf) (void)hipFree(c->ssaa_buf);
    if (c->stream_buf) (void)hipFree(c->stream_buf);
    if (c->frame_buf) (void)hipFree(c->frame_buf);
    if (c->orbit_host) (void)hipHostFree(c->orbit_host);
    if (c->orbit_dev) (void)hipFree(c->orbit_dev);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    free(c);
}

extern "C" int fr_ctx_compute_units(fr_ctx* c)
{
    if (!c) return fr_set_error(FR_ERR_INVALID_ARG, "ctx is NULL");
    return c->compute_units;
}

/* Options a caller needs, by name; value 0 restores the automatic choice (include/fractalrenderer_amd.h). */
extern "C" int fr_ctx_set_option(fr_ctx* c, const char* name, int64_t value)
{
    if (!c || !name) return fr_set_error(FR_ERR_INVALID_ARG, "ctx/name is NULL");
    if (!strcmp(name, "periodicity")) {
        if (value < -1 || value > (1 << 20)) return fr_set_error(FR_ERR_INVALID_ARG, "periodicity must be -1 (off), 0 (automatic: on), 1 (on) or a first snapshot window in iterations");
        c->tune_periodicity = value <= 0 ? (int32_t)value : (value == 1 ? 128 : (int32_t)((value + 15) / 16 * 16));
    } else if (!strcmp(name, "staging")) {
        /* 2 (block stream passes) and 4 (fused launch) were measured dead ends and left the library in 1.0: accepted,
         * they select the automatic schedule */
        if (value < 0 || value > 4) return fr_set_error(FR_ERR_INVALID_ARG, "staging must be 0 (automatic), 1 (single pass) or 3 (tile pass + lane-pool pass)");
        c->tune_staging = (value == 2 || value == 4) ? 0u : (uint32_t)value;
    } else if (!strcmp(name, "shards")) {
        if (value != 0 && value != 8 && value != 64) return fr_set_error(FR_ERR_INVALID_ARG, "shards must be 0 (automatic), 8 or 64");
        c->tune_shards = (uint32_t)value;
    } else if (!strcmp(name, "tile_kernel")) {
        if (value < 0 || value > 1) return fr_set_error(FR_ERR_INVALID_ARG, "tile_kernel must be 0 (automatic: lean where it applies) or 1 (general)");
        c->tune_tile_kernel = (uint32_t)value;
    } else if (!strcmp(name, "timing")) {
        if (value < 0 || value > 1) return fr_set_error(FR_ERR_INVALID_ARG, "timing must be 0 (off) or 1 (an event pair around every render: fr_ctx_last_kernel_ms)");
        c->timing = value != 0;
        if (!c->timing) c->have_timing = false;
    } else if (!strcmp(name, "diag_buffer")) {
        c->diag = (uint64_t*)(uintptr_t)value;        /* device pointer, 4 x u64 per wave of the grid; 0 = off */
    } else if (!strcmp(name, "diag_stride")) {
        c->diag_stride = (size_t)value;               /* u64 words between the diag regions of consecutive stages */
    } else if (!strcmp(name, "pool") || !strcmp(name, "stage_ratio") || !strcmp(name, "pool_evict_at") ||
               !strcmp(name, "pool_passes") || !strcmp(name, "queue_flags")) {
        /* retired with the schedules they steered (fresh-pixel pool, block stages, eviction passes): accepted, ignored */
    } else {
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown option '%s' (queue / stream tuning names moved to fr_ctx_set_tuning, "
                                                "fractalrenderer_amd/csrc/fr_tuning.h)", name);
    }
    return FR_OK;
}

/* Tuning knobs of the persistent queues and the survivor stream (fr_tuning.h: tests, tools/ and A/B measurements; not part
 * of the public header).  None of them can change a pixel. */
extern "C" int fr_ctx_set_tuning(fr_ctx* c, const char* name, int64_t value)
{
    if (!c || !name) return fr_set_error(FR_ERR_INVALID_ARG, "ctx/name is NULL");
    if (!strcmp(name, "workgroups_per_cu")) {
        if (value < 0 || value > 16) return fr_set_error(FR_ERR_INVALID_ARG, "workgroups_per_cu must be in [0,16]");
        c->tune_wg_per_cu = (uint32_t)value;
    } else if (!strcmp(name, "run_max")) {
        if (value < 0 || value > 1024) return fr_set_error(FR_ERR_INVALID_ARG, "run_max must be in [0,1024]");
        c->tune_run_max = (uint32_t)value;
    } else if (!strcmp(name, "run_min")) {
        if (value < 0 || value > 1024) return fr_set_error(FR_ERR_INVALID_ARG, "run_min must be in [0,1024]");
        c->tune_run_min = (uint32_t)value;
    } else if (!strcmp(name, "shift_bias")) {
        if (value < -16 || value > 16) return fr_set_error(FR_ERR_INVALID_ARG, "shift_bias must be in [-16,16]");
        c->tune_shift_bias = (int)value;
    } else if (!strcmp(name, "subtile_shape")) {
        if (value != 0 && value != 3 && value != 4 && value != 6)
            return fr_set_error(FR_ERR_INVALID_ARG, "subtile_shape must be 0, 3 (8x8), 4 (16x4) or 6 (64x1)");
        c->tune_shape = (uint32_t)value;
    } else if (!strcmp(name, "pool_refill_at")) {
        if (value < 0 || value > 64) return fr_set_error(FR_ERR_INVALID_ARG, "pool_refill_at must be in [0,64]");
        c->tune_pool_refill = (uint32_t)value;
    } else if (!strcmp(name, "stage_first")) {
        if (value < 0 || value > (1 << 24)) return fr_set_error(FR_ERR_INVALID_ARG, "stage_first out of range");
        c->tune_stage_first = (uint32_t)value;
    } else if (!strcmp(name, "stream_run_max")) {
        if (value < 0 || value > 1024) return fr_set_error(FR_ERR_INVALID_ARG, "stream_run_max must be in [0,1024]");
        c->tune_stream_run_max = (uint32_t)value;
    } else if (!strcmp(name, "stream_run_min")) {
        if (value < 0 || value > 1024) return fr_set_error(FR_ERR_INVALID_ARG, "stream_run_min must be in [0,1024]");
        c->tune_stream_run_min = (uint32_t)value;
    } else if (!strcmp(name, "stream_workgroups_per_cu")) {
        if (value < 0 || value > 8) return fr_set_error(FR_ERR_INVALID_ARG, "stream_workgroups_per_cu must be in [0,8]");
        c->tune_stream_wg_per_cu = (uint32_t)value;
    } else if (!strcmp(name, "probes")) {
        c->tune_probes = (uint32_t)value & 0xFu;
    } else if (!strcmp(name, "stream_probes")) {
        c->tune_stream_probes = (uint32_t)value & 0xFu;
    } else if (!strcmp(name, "stream_rotate")) {
        c->tune_stream_rotate = (uint32_t)value;
    } else if (!strcmp(name, "regions")) {
        if (value != 0 && value != 8 && value != 64) return fr_set_error(FR_ERR_INVALID_ARG, "regions must be 0 (automatic), 8 or 64");
        c->tune_regions = (uint32_t)value;
    } else if (!strcmp(name, "tile_pixels")) {
        if (value < 0 || value > 2) return fr_set_error(FR_ERR_INVALID_ARG, "tile_pixels must be 0 (automatic), 1 or 2");
        c->tune_tile_pixels = (uint32_t)value;
    } else if (!strcmp(name, "prepare")) {
        if (value < 0 || value > 1) return fr_set_error(FR_ERR_INVALID_ARG, "prepare must be 0 (automatic: inside the lean tile pass) or 1 (a launch of its own)");
        c->tune_prepare = (uint32_t)value;
    } else if (!strcmp(name, "debug_prologue_epoch")) {
        if (value < 0 || value > 0x0FFFFFFF) return fr_set_error(FR_ERR_INVALID_ARG, "debug_prologue_epoch: 28 bits");
        c->prologue_epoch = (uint32_t)value;          /* tests only: the epoch of the next in-kernel prologue is this + 1 */
    } else if (!strcmp(name, "tile_exit")) {
        if (value < 0 || value > 4096) return fr_set_error(FR_ERR_INVALID_ARG, "tile_exit must be 0 (automatic), 1 (off) or a cost in updates up to 4096");
        c->tune_tile_exit = (uint32_t)value;
    } else if (!strcmp(name, "tile_exit_from")) {
        if (value < 0 || value > (1 << 24)) return fr_set_error(FR_ERR_INVALID_ARG, "tile_exit_from out of range");
        c->tune_tile_exit_from = (uint32_t)value;
    } else if (!strcmp(name, "ssaa")) {
        if (value < 0 || value > 2) return fr_set_error(FR_ERR_INVALID_ARG, "ssaa must be 0 (automatic), 1 (sample loop of the general tile kernel) or 2 (staged)");
        c->tune_ssaa = (uint32_t)value;
    } else if (!strcmp(name, "stripes")) {
        if (value < 0 || value > 1) return fr_set_error(FR_ERR_INVALID_ARG, "stripes must be 0 (automatic) or 1 (Mandelbrot effects by the effects variant of the general tile kernel)");
        c->tune_stripes = (uint32_t)value;
    } else if (!strcmp(name, "ssaa_band_samples")) {
        if (value < 0 || value > (1ll << 30)) return fr_set_error(FR_ERR_INVALID_ARG, "ssaa_band_samples must be 0 (automatic: 2^29) or up to 2^30");
        c->tune_ssaa_band = (uint32_t)value;
    } else if (!strcmp(name, "pool_items_per_wg")) {
        if (value < 0 || value > 4096) return fr_set_error(FR_ERR_INVALID_ARG, "pool_items_per_wg must be in [0,4096]");
        c->tune_pool_items_per_wg = (uint32_t)value;
    } else if (!strcmp(name, "debug_region_blocks")) {
        c->debug_region_blocks = (uint32_t)value;     /* tests only (overflow reporting); 0 = the real capacity */
    } else {
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown tuning name '%s'", name);
    }
    return FR_OK;
}

/* workgroups of 256 threads per launch of the most recent render; bits 16.. = number of stages */
extern "C" int fr_ctx_last_grid(fr_ctx* c)
{
    if (!c) return fr_set_error(FR_ERR_INVALID_ARG, "ctx is NULL");
    return (int)(c->last_grid | ((uint32_t)c->last_stages << 16));
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

/* The kernels map pixel -> plane coordinate with  q' = fma(a - b*RN(a*y) , y, RN(a*y)),  y = RN(1/b)
 * instead of the as-written IEEE divide a / b (Markstein's correction step).  That is the
 * correctly rounded quotient in all but exotic cases; rather than rely on the theorem's side
 * conditions, evaluate the identical expression here for EVERY numerator the frame uses (one per
 * column, one per row) and allow the divide-free path only if all of them equal a / b.
 * Result cached per (W, H, fractal, precision). */
template <typename T>
static bool quotients_exact(uint32_t W, uint32_t H, bool julia)
{
    const T resx = (T)W, resy = (T)H;
    const T inv_w = (T)1 / resx, inv_h = (T)1 / resy;
    auto same = [](T a, T b, T rb) {
        const T q = a * rb;
        const T r = std::fma(-q, b, a);
        return std::fma(r, rb, q) == a / b;
    };
    if (julia) {                                   /* shaders/julia.comp:325  uv = pix / size */
        for (uint32_t x = 0; x < W; ++x) if (!same((T)x, resx, inv_w)) return false;
        for (uint32_t y = 0; y < H; ++y) if (!same((T)y, resy, inv_h)) return false;
    } else {                                       /* shaders/mandelbrot.comp:150  (pix - 0.5 res) / res.y */
        for (uint32_t x = 0; x < W; ++x) if (!same((T)x - (T)0.5 * resx, resy, inv_h)) return false;
        for (uint32_t y = 0; y < H; ++y) if (!same((T)y - (T)0.5 * resy, resy, inv_h)) return false;
    }
    return true;
}

static bool exact_division_ok(fr_ctx* c, uint32_t W, uint32_t H, bool julia, bool f64)
{
    for (int k = 0; k < 8; ++k) {
        const fr_ctx::DivCheck& d = c->div_cache[k];
        if (d.valid && d.W == W && d.H == H && d.julia == julia && d.f64 == f64) return d.ok;
    }
    const bool ok = f64 ? quotients_exact<double>(W, H, julia) : quotients_exact<float>(W, H, julia);
    fr_ctx::DivCheck& slot = c->div_cache[c->div_next++ & 7];
    slot.valid = true; slot.W = W; slot.H = H; slot.julia = julia; slot.f64 = f64; slot.ok = ok;
    return ok;
}

template <typename T, int FRACTAL, bool EFFECTS, bool SSAA>
static hipError_t launch_tile_aa(int shape, dim3 grid, hipStream_t s, const LaunchArgs& a)
{
    switch (shape) {
    case 6: hipLaunchKernelGGL((tile_kernel<T, FRACTAL, 6, EFFECTS, SSAA>), grid, dim3(kBlockThreads), 0, s, a); break;
    case 4: hipLaunchKernelGGL((tile_kernel<T, FRACTAL, 4, EFFECTS, SSAA>), grid, dim3(kBlockThreads), 0, s, a); break;
    default: hipLaunchKernelGGL((tile_kernel<T, FRACTAL, 3, EFFECTS, SSAA>), grid, dim3(kBlockThreads), 0, s, a); break;
    }
    return hipGetLastError();
}

template <typename T, int FRACTAL, bool EFFECTS>
static hipError_t launch_tile(int shape, dim3 grid, hipStream_t s, const LaunchArgs& a)
{
    if constexpr (!EFFECTS) {
        /* one-sample pass run to max_iter with cycle closing on: its own variant (8x8 sub-tiles only), so that the
         * default kernel does not carry the snapshot registers */
        if (a.aa <= 1 && a.period_window && shape == 3) {
            hipLaunchKernelGGL((tile_kernel<T, FRACTAL, 3, false, false, true>), grid, dim3(kBlockThreads), 0, s, a);
            return hipGetLastError();
        }
    }
    return a.aa > 1 ? launch_tile_aa<T, FRACTAL, EFFECTS, true>(shape, grid, s, a)
                    : launch_tile_aa<T, FRACTAL, EFFECTS, false>(shape, grid, s, a);
}

template <typename T, int FRACTAL>
static hipError_t launch_tile_lean(int np, dim3 grid, hipStream_t s, const LaunchArgs& a)
{
    if (np == 2) {
        if (a.period_window)
            hipLaunchKernelGGL((tile_lean_kernel<T, FRACTAL, true, 2>), grid, dim3(kBlockThreads), 0, s, a);
        else
            hipLaunchKernelGGL((tile_lean_kernel<T, FRACTAL, false, 2>), grid, dim3(kBlockThreads), 0, s, a);
    } else {
        if (a.period_window)
            hipLaunchKernelGGL((tile_lean_kernel<T, FRACTAL, true, 1>), grid, dim3(kBlockThreads), 0, s, a);
        else
            hipLaunchKernelGGL((tile_lean_kernel<T, FRACTAL, false, 1>), grid, dim3(kBlockThreads), 0, s, a);
    }
    return hipGetLastError();
}

/* the Mandelbrot shader's effects through the lean kernels (kernel code FRACTAL = 3, shade_stripes): two sub-tiles per trip, no cycle closing */
template <typename T>
static hipError_t launch_tile_lean_stripes(dim3 grid, hipStream_t s, const LaunchArgs& a)
{
    hipLaunchKernelGGL((tile_lean_kernel<T, 3, false, 2>), grid, dim3(kBlockThreads), 0, s, a);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_pool_stripes(dim3 grid, hipStream_t s, const LaunchArgs& a)
{
    hipLaunchKernelGGL((pool_kernel<T, 3, false>), grid, dim3(kBlockThreads), 0, s, a);
    return hipGetLastError();
}

/* control block + coordinate tables of a lean render (prepare_kernel) */
template <typename T, int FRACTAL>
static hipError_t launch_prepare(hipStream_t s, const LaunchArgs& a, uint32_t* ctrl, uint32_t n_ctrl, const Feedback& fb)
{
    const uint32_t n = (uint32_t)(a.W + a.H) > n_ctrl ? (uint32_t)(a.W + a.H) : n_ctrl;
    const dim3 grid((n + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL((prepare_kernel<T, FRACTAL == 0 ? 0 : 1>), grid, dim3(kBlockThreads), 0, s, a, ctrl, n_ctrl, fb);
    return hipGetLastError();
}

template <typename T, int FRACTAL>
static hipError_t launch_stream_pool(dim3 grid, hipStream_t s, const LaunchArgs& a)
{
    if (a.period_window)
        hipLaunchKernelGGL((pool_kernel<T, FRACTAL, true>), grid, dim3(kBlockThreads), 0, s, a);
    else
        hipLaunchKernelGGL((pool_kernel<T, FRACTAL, false>), grid, dim3(kBlockThreads), 0, s, a);
    return hipGetLastError();
}

/* run fn(T{}, integral_constant<int, FRACTAL>{}) for the runtime (fractal, precision) */
template <class Fn>
static hipError_t by_variant(int fractal, bool f64, Fn&& fn)
{
    using std::integral_constant;
    switch (fractal) {
    case FR_FRACTAL_JULIA:        return f64 ? fn(double{}, integral_constant<int, 1>{}) : fn(float{}, integral_constant<int, 1>{});
    case FR_FRACTAL_BURNING_SHIP: return f64 ? fn(double{}, integral_constant<int, 2>{}) : fn(float{}, integral_constant<int, 2>{});
    default:                      return f64 ? fn(double{}, integral_constant<int, 0>{}) : fn(float{}, integral_constant<int, 0>{});
    }
}

static uint32_t ceil_log2(uint32_t v)
{
    uint32_t b = 0;
    while ((1u << b) < v && b < 31) ++b;
    return b;
}

/* control block in device memory, zeroed by ONE memset per render:
 *   words [s * kStageWords, + kMaxShards * 32): the (8 or 64) queue heads of stage s, 128 B apart
 *   the next kMaxShards * 32 words: the (8 or 64) region counters of the survivor stream written by stage s */
static uint32_t* stage_heads(fr_ctx* c, int s) { return c->d_ctrl + (size_t)s * kStageWords; }
static uint32_t* stage_counter(fr_ctx* c, int s) { return c->d_ctrl + (size_t)s * kStageWords + (size_t)kMaxShards * kShardStrideWords; }

/* A survivor stream that ran out of blocks (StreamRef::overflow) loses pixels: report it as a failed render at the
 * next point where the host knows the kernels are done.  Sticky until reported. */
static int check_overflow(fr_ctx* c)
{
    if (__atomic_load_n(c->overflow_host, __ATOMIC_RELAXED) == 0u) return FR_OK;
    __atomic_store_n(c->overflow_host, 0u, __ATOMIC_RELAXED);
    return fr_set_error(FR_ERR_INTERNAL, "a kernel reported an internal error (a survivor stream overflowed, or a lane-pool wave "
                                         "gave up on a stretch that would not end): the frame of the last render on this context "
                                         "is incomplete, please report the parameters");
}

/* Cycle closing ("periodicity"): on unless switched off.  Where it takes effect: the lane-pool pass (PERIOD
 * instantiation), the tile kernel when it runs samples to max_iter with 8x8 sub-tiles -- a one-pass frame (PERIOD
 * instantiation) and SSAA (always compiled in).  Where it does not: the effects variants, one-pass frames with 16x4 /
 * 64x1 sub-tiles and Deep_Zoom -- those iterate every sample to max_iter, as the reference does. */
static uint32_t period_window(const fr_ctx* c)
{
    return c->tune_periodicity < 0 ? 0u : (c->tune_periodicity == 0 ? 128u : (uint32_t)c->tune_periodicity);
}

/* what the first launch of a render forwards from the previous one (see Feedback) */
static Feedback feedback_of(fr_ctx* c)
{
    Feedback fb;
    fb.dev_flag = c->d_ctrl + kFeedbackWord;
    fb.host_word = c->overflow_dev + 1;
    fb.prev_seq = c->render_seq;                 /* the caller increments render_seq after its launches */
    return fb;
}

/* Automatic cycle closing ("periodicity" = 0) of the lane pool.  Its PERIOD instantiation costs a frame in which nothing
 * ever closes -- a Julia dust, the C5 view -- 7 % / 3.5 % (profiles/r03_periodicity_cost.txt: the comparisons were made
 * all but free in round 3, the rest would not yield), and frames come in sequences of similar views.  So a context that
 * has looked and closed NOTHING renders its next kProbeEvery - 1 frames of the same kind (fractal, precision, max_iter,
 * geometry) with the plain instantiation, then looks again.  What the pool found travels back without a synchronisation
 * (Feedback), so the verdict on frame n is known when frame n + 2 is planned -- or later, if the host runs ahead of the
 * device: until then the pool keeps looking.  A view whose pools close cycles always looks.  Nothing a pixel depends on. */
static bool pool_wants_cycle_closing(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t rows)
{
    if (c->tune_periodicity != 0) return c->tune_periodicity > 0;          /* explicit: on (any window) or off */
    uint64_t key = 1469598103934665603ull;
    /* ... and WHERE it looks, coarsely: the octave of the zoom, the centre in units of that octave's view height, the Julia
     * constant to 1/64 -- a sequence that leaves a dust for an interior-heavy view (or pans by a view, or zooms by 2x) starts
     * looking again at once instead of finishing its 14 frames of not looking */
    int zexp = 0;
    (void)frexp(fabs(p->zoom), &zexp);
    const double cell = ldexp(1.0, zexp);
    const uint64_t parts[10] = {(uint64_t)p->fractal_type, (uint64_t)p->precision, (uint64_t)p->max_iterations, W, rows,
                                (uint64_t)(int64_t)zexp, (uint64_t)(int64_t)floor(p->center_x / cell), (uint64_t)(int64_t)floor(p->center_y / cell),
                                (uint64_t)(int64_t)floor(p->julia_c_real * 64.0), (uint64_t)(int64_t)floor(p->julia_c_imag * 64.0)};
    for (uint64_t v : parts) key = (key ^ v) * 1099511628211ull;
    if (key != c->probe_key) { c->probe_key = key; c->probe_mode = 0; c->probe_first = 0; }
    /* the verdict the device forwarded last: (render number << 1) | "closing cycles paid" (an eighth of the pool's records
     * and more were retired by a closed cycle), of a render whose pool looked */
    const uint32_t word = __atomic_load_n(c->overflow_host + 1, __ATOMIC_RELAXED);
    const uint32_t seq = word >> 1;
    const bool nothing_closed = (word & 1u) == 0u;
    const uint32_t mine = c->render_seq + 1;                                /* this render's number */
    switch (c->probe_mode) {
    case 0:      /* LOOK: only renders whose pools looked are forwarded, and from probe_first on those are renders of this key:
                  * any of their verdicts counts (a host that runs ahead of the device may never see the one of a particular
                  * render) */
        if (c->probe_first != 0 && seq >= c->probe_first && seq <= c->render_seq && nothing_closed) {
            c->probe_mode = 1; c->skip_left = kProbeEvery - 2;
            return false;
        }
        if (c->probe_first == 0) c->probe_first = mine;
        return true;
    case 1:      /* SKIP */
        if (c->skip_left > 0) { --c->skip_left; return false; }
        c->probe_mode = 2; c->probe_first = mine;                           /* the one look */
        return true;
    default:     /* WAIT: the host may be many frames ahead of the device; one look in flight is enough */
        if (seq == c->probe_first) {
            if (nothing_closed) { c->probe_mode = 1; c->skip_left = kProbeEvery - 2; return false; }
            c->probe_mode = 0; c->probe_first = mine;                       /* the view closes cycles now: look again */
            return true;
        }
        return false;
    }
}

/* zero the queue heads and stream counters of the next render (a kernel, not a memset node: see clear_words_kernel) */
static hipError_t clear_control_block(fr_ctx* c, hipStream_t stream, int nstages)
{
    const uint32_t n = (uint32_t)((size_t)nstages * kStageWords);
    hipLaunchKernelGGL(clear_words_kernel, dim3((n + 4 * kBlockThreads - 1) / (4 * kBlockThreads)), dim3(kBlockThreads), 0, stream, c->d_ctrl, n,
                       feedback_of(c));
    return hipGetLastError();
}

/* Work on the context's own stream (exports, colorize without a stream argument) must see the planes of a render that
 * was enqueued on a CALLER's stream: ev_end was recorded there behind the last launch. */
static hipError_t order_after_last_render(fr_ctx* c, hipStream_t s)
{
    if (!c->render_on_user_stream || !c->have_render || c->last_stream == s) return hipSuccess;
    /* recorded NOW, on the stream that render went to: behind it (and behind whatever the caller has enqueued there since) --
     * the renders themselves record nothing for this */
    const hipError_t e = hipEventRecord(c->ev_order, c->last_stream);
    if (e != hipSuccess) return e;
    return hipStreamWaitEvent(s, c->ev_order, 0);
}

/* Deep_Zoom: what VulkanEngine::prepare_deep_zoom_rendering + dispatch do per frame
 * (src/vk_engine.cpp:215-251, src/compute_effect_manager.h:236-324): recompute the fp64 reference orbit
 * at the view centre on the host (single point, sequential), narrow it to float pairs
 * (src/deep_zoom_system.cpp:102-110), upload, launch the perturbation kernel. */
/* Deep_Zoom orbit buffers: pinned staging (fp64 orbit + its float narrowing, `cap` scalars each) + the device copy */
static int reserve_orbit(fr_ctx* c, size_t need)
{
    if (need <= c->orbit_cap) return FR_OK;
    if (c->orbit_host) { (void)hipHostFree(c->orbit_host); c->orbit_host = nullptr; }
    if (c->orbit_dev) { (void)hipFree(c->orbit_dev); c->orbit_dev = nullptr; }
    c->orbit_cap = 0;
    FR_HIP_TRY(hipHostMalloc((void**)&c->orbit_host, need * sizeof(double) + need * sizeof(float)));
    FR_HIP_TRY(hipMalloc((void**)&c->orbit_dev, need * sizeof(float)));
    c->orbit_cap = need;
    return FR_OK;
}

static int enqueue_deep_zoom(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_shard* norm,
                             uint32_t rows_local, float* rgba, void* nu, int32_t* iter, hipStream_t stream, bool reserve_only,
                             bool out_frame)
{
    const int32_t max_iter = p->max_iterations;
    int32_t ref_iter = 0;
    if (reserve_only) return p->use_perturbation ? reserve_orbit(c, (size_t)max_iter * 2) : FR_OK;
    if (p->use_perturbation) {
        const size_t need = (size_t)max_iter * 2;
        /* the pinned staging buffer may still feed an earlier asynchronous upload */
        FR_HIP_TRY(hipStreamSynchronize(stream));
        int rs = reserve_orbit(c, need);
        if (rs != FR_OK) return rs;
        double* xy = (double*)c->orbit_host;
        float* xyf = (float*)(xy + c->orbit_cap);
        int st = fr_reference_orbit(p->center_x, p->center_y, max_iter, xy, &ref_iter);
        if (st != FR_OK) return st;
        for (int32_t i = 0; i < 2 * ref_iter; ++i) xyf[i] = (float)xy[i];
        FR_HIP_TRY(hipMemcpyAsync(c->orbit_dev, xyf, (size_t)ref_iter * 2 * sizeof(float), hipMemcpyHostToDevice, stream));
    }

    DeepZoomArgs a;
    memset(&a, 0, sizeof(a));
    a.cx_hi = (float)p->center_x; a.cx_lo = (float)(p->center_x - (double)a.cx_hi);      /* split_double, :252-257 */
    a.cy_hi = (float)p->center_y; a.cy_lo = (float)(p->center_y - (double)a.cy_hi);
    a.zoom_hi = (float)p->zoom;   a.zoom_lo = (float)(p->zoom - (double)a.zoom_hi);
    a.bailout = p->bailout; a.color_offset = p->color_offset; a.color_scale = p->color_scale;
    a.palette_mode = p->palette_mode; a.max_iter = max_iter; a.ref_iter = ref_iter;
    a.W = (int32_t)W; a.H = (int32_t)H; a.rows_local = (int32_t)rows_local;
    a.part = (int32_t)norm->part; a.nparts = (int32_t)norm->nparts; a.rows_per_strip = (int32_t)norm->rows_per_strip;
    a.out_frame = out_frame ? 1 : 0;
    a.orbit = reinterpret_cast<const float2*>(c->orbit_dev);
    a.rgba = reinterpret_cast<float4*>(rgba); a.nu = (float*)nu; a.iter = iter;

    QueueArgs& q = a.q;
    q.heads = stage_heads(c, 0);
    q.nsx = (W + 7) / 8;
    q.nsx_shift = -1;
    q.n_items = q.nsx * ((rows_local + 7) / 8);
    q.n_blk = (q.n_items + kShardBlock - 1) / kShardBlock;
    uint32_t grid = (uint32_t)c->compute_units * 8u;
    const uint32_t max_grid = (q.n_items + 7) / 8;           /* a wave takes at least 2 sub-tiles per dequeue */
    if (grid > max_grid) grid = max_grid < 1 ? 1 : max_grid;
    q.run_shift = ceil_log2(16u * ((grid * 4u + kShards - 1) / kShards));
    q.run_min = 2; q.run_max = 8; q.flags = 0; q.ns_log2 = 3;
    c->last_grid = grid;
    c->last_stages = 1;

    FR_HIP_TRY(clear_control_block(c, stream, 1));
    if (c->timing) FR_HIP_TRY(hipEventRecord(c->ev_begin, stream));
    hipLaunchKernelGGL((deep_zoom_kernel<3>), dim3(grid), dim3(kBlockThreads), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "deep-zoom kernel launch failed: %s", hipGetErrorString(e));
    if (c->timing) FR_HIP_TRY(hipEventRecord(c->ev_end, stream));
    c->have_timing = c->timing;
    c->have_render = true;
    c->last_stream = stream;
    ++c->render_seq;                             /* its first launch forwarded the previous render's verdict (Feedback) */
    c->last_pool_closing = -1;
    return FR_OK;
}

/* the parameter part of the kernel argument block (everything that does not depend on the frame geometry) */
static void fill_params(LaunchArgs& a, const fr_params* p)
{
    const bool f64 = p->precision == FR_PRECISION_F64;
    memset(&a, 0, sizeof(a));
    a.center_x = p->center_x; a.center_y = p->center_y; a.zoom = p->zoom;
    a.julia_cx = p->julia_c_real; a.julia_cy = p->julia_c_imag;
    a.bailout = p->bailout;
    a.log_bailout = f64 ? log((double)p->bailout) : (double)logf(p->bailout);
    a.max_iter = p->max_iterations;
    a.aa = p->antialiasing_samples;
    a.palette_mode = p->palette_mode;
    a.color_offset = p->color_offset; a.color_scale = p->color_scale;
    a.interior_style = p->interior_style;
    a.trap_enabled = p->orbit_trap_enabled; a.trap_radius = p->orbit_trap_radius;
    a.stripe_enabled = p->stripe_enabled; a.stripe_density = p->stripe_density;
    a.brightness = p->color_brightness; a.saturation = p->color_saturation; a.contrast = p->color_contrast;
    a.flags = p->flags;
    /* shaders/mandelbrot.comp numbering for Mandelbrot; burning_ship.comp:14-182 == julia.comp:20-181 */
    fr_palette_table_build(p->fractal_type != FR_FRACTAL_MANDELBROT ? 1 : 0, p->palette_mode, &a.pal);
    a.inv_max_iter = 1.0 / (double)p->max_iterations;
    a.inv_log2_bailout = 1.0 / log2((double)p->bailout);
    a.inv_max_iter_f = (float)a.inv_max_iter; a.inv_log2_bailout_f = (float)a.inv_log2_bailout;
    a.color_scale_d = (double)p->color_scale; a.color_offset_d = (double)p->color_offset;
    a.lib_log = !(p->bailout > 1.0f);        /* log2_pos() needs positive arguments: |z|^2 > 1 */
}

/* colourings that need more of the orbit than (escape index, |z|^2): the as-written effects loops */
static bool needs_effects(const fr_params* p)
{
    switch (p->fractal_type) {
    case FR_FRACTAL_MANDELBROT:   return p->orbit_trap_enabled || p->stripe_enabled || p->interior_style == 2;
    case FR_FRACTAL_BURNING_SHIP: return p->orbit_trap_enabled || (p->stripe_enabled && p->interior_style == 2) ||
                                         p->interior_style == 3;
    default: return false;
    }
}

/* ---- geometry of the tile pass ----------------------------------------------------------------------
 * Sub-tiles of 64 pixels (2^shape wide) in blocks of 16 dealt round by round to the 8 or 64 shards, a shard's place
 * rotating with the round (WaveQueue::block_of; blocks >= n_blk are skipped by the kernels); a persistent grid of
 * exactly the resident set; run lengths and probe limit of the queue. */
static QueueArgs plan_tile_queue(const fr_ctx* c, uint32_t W, uint32_t rows_local, int shape, bool bounded, bool moderate,
                                 bool six_fit, uint32_t* grid_out, uint32_t* waves_per_shard_out)
{
    const uint32_t fpw = 1u << shape, fph = 64u >> shape;
    QueueArgs tq;
    memset(&tq, 0, sizeof(tq));
    tq.nsx = (W + fpw - 1) / fpw;
    tq.nsx_shift = -1;
    for (int b = 0; b < 31; ++b)
        if (tq.nsx == (1u << b)) tq.nsx_shift = b;
    const uint32_t nsy = (rows_local + fph - 1) / fph;
    tq.n_items = tq.nsx * nsy;
    tq.n_blk = (tq.n_items + kShardBlock - 1) / kShardBlock;

    /* The fp64 tile kernel holds 5 workgroups of 256 threads per CU (the per-wave timeline of the diag buffer
     * shows workgroups beyond the resident set only start when resident ones exit, and find the queue dry):
     * launch exactly the resident set.  Measured 5 vs 4: C2 +1.9 %, C3 +5.6 %, C5 +1.7 %; 6-8 (the one-sample
     * kernel fits 7 at 69 VGPRs) within 1 %. */
    /* the staged lean tile kernel in fp32 (52 VGPRs, 8.5 KB of LDS) holds 6: C3 -1.4 % */
    const uint32_t wg_per_cu = c->tune_wg_per_cu ? c->tune_wg_per_cu : (six_fit ? 6u : 5u);
    uint32_t grid = (uint32_t)c->compute_units * wg_per_cu;
    /* never more waves than the shortest runs can feed: a wave takes at least run_min sub-tiles per dequeue (4 when
     * bounded, 2 otherwise), and waves that find nothing still cost their launch and their exit probes -- at 512^2
     * a grid of one wave per sub-tile left 3 of 4 waves without work: 0.083 ms per frame against 0.048 ms */
    const uint32_t per_wave = c->tune_run_min ? c->tune_run_min : (bounded ? 4u : 2u);
    const uint32_t max_grid = (tq.n_items + 4u * per_wave - 1u) / (4u * per_wave);
    if (grid > max_grid) grid = max_grid < 1 ? 1 : max_grid;
    /* Shards: 64 (8 per XCD) where waves stop at their home shard(s) and the frame has work for them -- 64 queue heads
     * (and 64 block counters of the survivor stream) instead of 8 take the same claims at 8x the rate (kMaxShards);
     * launches with unlimited stealing keep 8: a wave probes every shard before it exits. */
    const bool limited = (bounded || moderate) && grid >= 64u;
    /* (from 4 blocks per shard and 256 workgroups: a 512^2 frame -- 256 blocks -- measured -6 % with 64 shards, end of round 4;
     * the rule had asked for 8 blocks per shard and 512 workgroups) */
    uint32_t ns = (limited && grid >= 256u && tq.n_blk >= 4u * (uint32_t)kMaxShards) ? (uint32_t)kMaxShards : (uint32_t)kShards;
    if (c->tune_shards) ns = c->tune_shards;
    tq.ns_log2 = ns == (uint32_t)kMaxShards ? 6u : 3u;
    const uint32_t waves_per_shard = (grid * 4u + ns - 1) / ns;
    auto clamp_shift = [&](int v) { v += c->tune_shift_bias; return (uint32_t)(v < 0 ? 0 : (v > 31 ? 31 : v)); };
    /* Run length of a dequeue = clamp(remaining >> run_shift, run_min, run_max).
     *  - unbounded items (single pass, measured on C2, profiles/r01_sweep_c2.txt): sub-tile cost varies 100x,
     *    so long runs leave a tail of waves holding several max_iter sub-tiles while single sub-tile claims
     *    saturate the queue words (~88 dequeues/us each: a 0.44 ms floor): short runs of 2..8;
     *  - bounded items (staged tile pass: at most b0 iterations each): long runs are safe and hide the
     *    dequeue latency that dominates cheap sub-tiles. */
    if (bounded) {
        /* 64 shards: runs a quarter as long again (remaining / (8 waves' worth)) -- with waves that stop at their home
         * shards the last runs of a shard are its tail, and at 80 waves per shard a run of 16 sub-tiles inside the set is
         * 20 us on a chip that is otherwise done: C2 tile pass 114 -> 105 us, C3 125 -> 112 us, C5 1639 -> 1593 us
         * (a view where every sub-tile costs the same pays for the extra claims: 76 -> 85 us) */
        tq.run_shift = clamp_shift((int)ceil_log2(2u * waves_per_shard) + (ns == (uint32_t)kMaxShards ? 2 : 0));
        tq.run_max = c->tune_run_max ? c->tune_run_max : 32u;
        tq.run_min = c->tune_run_min ? c->tune_run_min : 4u;
    } else {
        tq.run_shift = clamp_shift((int)ceil_log2(16u * waves_per_shard));
        tq.run_max = c->tune_run_max ? c->tune_run_max : 8u;
        tq.run_min = c->tune_run_min ? c->tune_run_min : 2u;
    }
    if (tq.run_min > tq.run_max) tq.run_min = tq.run_max;
    tq.flags = 0u;
    /* Bounded items are dealt evenly to the shards, so a wave whose home shard is dry exits instead of
     * probing the other 7 (measured: the exit storm of 4096 waves x 8 serialized atomics costs 31 us of the
     * 260 us tile pass of C2 and 36 of the 74 us of a 1/8 shard, profiles/r01_probe_limit.txt).  Unbounded
     * passes keep full stealing; so do grids with fewer workgroups than shards. */
    /* Passes whose items are long (SSAA: aa^2 samples to max_iter per pixel; effects; a forced single pass) keep full
     * stealing: with home + one neighbour the C5 view at 2x2 samples takes 9.96 ms instead of 8.12 ms. */
    /* 64 shards: home + the next one of the same XCD (80 waves per shard: a second look evens out the ends) */
    uint32_t probes = c->tune_probes ? c->tune_probes : (limited ? (ns == (uint32_t)kMaxShards ? 2u : 1u) : 0u);
    if (grid < ns) probes = 0;
    tq.flags |= probes << kQueueProbeShift;
    *grid_out = grid;
    *waves_per_shard_out = waves_per_shard;
    return tq;
}

/* ---- stage schedule ----------------------------------------------------------------------------------
 * Two passes -- the tile pass runs [0, b0), the lane pool [b0, max_iter) -- or one.  Not staged: SSAA (samples of a pixel
 * must meet again to be averaged), the effects variants (accumulators along the whole orbit), short max_iter.  Returns
 * the number of passes; bounds[k] = upper iteration bound of pass k.  (Block stream passes with x4 budgets and a fused
 * one-launch schedule were built and measured slower everywhere: DESIGN.md section 7.) */
static int staging_threshold(const fr_params* p, size_t npx)
{
    const bool big = npx > ((size_t)1 << 23);
    if (p->fractal_type == FR_FRACTAL_JULIA) return npx <= ((size_t)1 << 20) ? 512 : 256;   /* (small frames: 256^2 ... 1024x768 at 256, the
                                                                                              * dust 55 -> 36 us in one pass, a filled set 50 -> 23) */
    /* Small frames (end of round 4, profiles/r04_small_frame_staging.txt): the second launch and the lane pool's ramp and
     * run-out are ~45-60 us whatever the frame, which a frame of half a megapixel does not win back before max_iter 1024-2048
     * (256^2 at 512, fp64: 74 -> 49 us in one pass; 512^2 at 1024, fp32: 92 -> 72; but the Seahorse view at 2048: 129 against
     * 168-190 in one pass): up to 2^19 pixels fp32 stages from 1536, fp64 from 1024 -- from 1536 up to 2^18 pixels; a Julia set from 512 up to 2^20 pixels. */
    if (npx <= ((size_t)1 << 19)) return p->precision == FR_PRECISION_F64 ? (npx <= ((size_t)1 << 18) ? 1536 : 1024) : 1536;
    return p->precision == FR_PRECISION_F64 ? (big ? 384 : 512) : (big ? 512 : 768);
}

static int plan_stages(const fr_ctx* c, const fr_params* p, bool effects, size_t npx, bool pool_runs_everything, int bounds[kMaxStages])
{
    const int max_iter = p->max_iterations;
    int nstage = 0;
    const bool allow = !effects && p->antialiasing_samples <= 1 && c->tune_staging != 1u;
    /* tile-pass budget: ~max_iter/28 rounded to the unchecked block, within [32, 192] (measured best:
     * 32 at max_iter 1024, 64 at 2048, 128-192 at 4096, flat at 16384) */
    int auto_first = ((max_iter / 28 + kFastBlock / 2) / kFastBlock) * kFastBlock;
    auto_first = auto_first < 32 ? 32 : (auto_first > 192 ? 192 : auto_first);
    /* fp64 frames of 2^24 pixels and more whose lane pool runs every survivor to max_iter (cycle closing off, or skipped
     * because it closed nothing lately): ~max_iter/11 within [96, 192].  Round 3's sweeps with the lean tile kernel
     * (profiles/r03_b0_and_pool_tuning.txt): C2 (4096^2, 1024) 96 against 32: -1.8 %, C5 (8192^2, 4096) 192 against 144:
     * -0.5 %; a 1080p frame at 1024 keeps 32 (96: +8 %), the fp32 Julia dust its 80 (96-128 within noise, 256: +8 %).
     * A pool that closes cycles makes a survivor cheap, and the long tile pass only costs: C2 with cycle closing
     * 0.517 ms at 32, 0.583 ms at 96. */
    if (pool_runs_everything && p->precision == FR_PRECISION_F64 && npx >= ((size_t)1 << 24)) {
        int big = ((max_iter / 11 + kFastBlock / 2) / kFastBlock) * kFastBlock;
        big = big < 96 ? 96 : (big > 192 ? 192 : big);
        if (big > auto_first) auto_first = big;
    }
    const int first = c->tune_stage_first ? (int)c->tune_stage_first : auto_first;
    /* The second pass pays off where orbits are long.  Rounds 1-3: below max_iter ~768 (~384 on frames above 4K) ONE pass
     * whose waves stop at their home shard was faster -- 1080p at max_iter 256: 0.061 ms against 0.109 ms -- and above it
     * the two passes won by up to 35 % (profiles/r01_staging_crossover.txt).  Round 4's lane pool (deferred escapes) moved
     * the crossover down where escapes are spread out (profiles/r04_staging_crossover.txt, 168 cells): a Julia set wins
     * with two passes from max_iter 256 (-5 to -12 %; 384: -15 to -30 %; 512: -23 to -36 %), an fp64 Mandelbrot view
     * from 512 at every size up to 4K (-6 to -13 %; 384 on frames above 2^23 pixels as before: the default view loses 5-9 %
     * there, the Seahorse view wins 15-28 %), an fp32 one stays at 768 (512 above 2^23 pixels, where 384 lost 7-11 %).
     * An explicit "staging" or "stage_first" stages whenever there is room for two budgets. */
    const int auto_min = staging_threshold(p, npx);
    const bool forced = c->tune_staging != 0 || c->tune_stage_first != 0;
    if (allow && (forced ? max_iter >= 2 * first : max_iter >= auto_min)) {
        int b = first - first % kFastBlock;                      /* the budget is a multiple of the unchecked block */
        if (b < kFastBlock) b = kFastBlock;
        if (b < max_iter) bounds[nstage++] = b;
    }
    bounds[nstage++] = max_iter;
    return nstage;
}

/* Survivor stream: blocks of 64 records {pixel u32, iterations done u32, nfields x T}.  Worst case: every
 * sample survives (npx/64 full blocks) + one partial block per writer wave; the regions of the stream hold
 * 1.5x that, so a region that fills up can spill into its neighbours.  Grow-only (happens on the first
 * render of a larger geometry, not capturable). */
static int reserve_stream(fr_ctx* c, size_t npx, size_t nfields, bool f64, uint32_t writer_workgroups, uint32_t nregions,
                          uint32_t* region_blocks)
{
    const size_t block_bytes = 2 * 64 * 4 + nfields * 64 * (f64 ? 8 : 4);
    const uint32_t worst_blocks = (uint32_t)((npx + 63) / 64) + writer_workgroups * 4u + 16u;
    *region_blocks = (worst_blocks * 3u / 2u + nregions - 1) / nregions + 1u;
    const size_t need = (size_t)*region_blocks * nregions * block_bytes;
    if (need <= c->stream_bytes) return FR_OK;
    if (c->stream_buf) { (void)hipFree(c->stream_buf); c->stream_buf = nullptr; }
    c->stream_bytes = 0;
    FR_HIP_TRY(hipMalloc(&c->stream_buf, need));
    c->stream_bytes = need;
    return FR_OK;
}

/* reserve_only: do everything a render of this geometry would do BEFORE its first launch -- grow the survivor streams,
 * the Deep_Zoom orbit buffers, fill the exact-division cache -- and stop (fr_ctx_reserve). */
static int enqueue_ssaa_staged(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_shard* norm, uint32_t rows_local,
                               float* rgba, void* nu, int32_t* iter, hipStream_t stream, bool reserve_only, bool out_frame);
static int staging_threshold(const fr_params* p, size_t npx);

/* occupancy exit of the lean tile pass (escape_run_lean): what a record costs the lane pool, in updates, and the updates a
 * trip runs before it may leave */
constexpr uint32_t kTileExitCost = 48u;
constexpr uint32_t kTileExitFrom = 32u;

/* ssaa_of > 1: this IS the sample grid of a supersampled res_w x res_h frame (enqueue_ssaa_staged): lean kernels only, the
 * coordinate tables hold the samples' coordinates */
static int enqueue_render(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_shard* shard,
                          float* rgba, void* nu, int32_t* iter, hipStream_t stream, bool reserve_only = false,
                          bool out_frame = false, int ssaa_of = 0, uint32_t res_w = 0, uint32_t res_h = 0)
{
    if (!reserve_only) {
        const int ov = check_overflow(c);          /* of an earlier asynchronous render nobody has asked about */
        if (ov != FR_OK) return ov;
    }
    fr_shard whole = {0u, 1u, H};
    const fr_shard* sh = shard ? shard : &whole;
    fr_shard norm = *sh;
    if (norm.nparts == 0) norm.nparts = 1;
    if (norm.rows_per_strip == 0) norm.rows_per_strip = (norm.nparts == 1) ? H : 1;
    if (norm.part >= norm.nparts)
        return fr_set_error(FR_ERR_INVALID_ARG, "shard part %u >= nparts %u", norm.part, norm.nparts);
    const uint32_t rows_local = fr_shard_rows(&norm, H);
    if (rows_local == 0) return FR_OK;           /* this part owns no rows */
    if (p->fractal_type == FR_FRACTAL_DEEP_ZOOM)
        return enqueue_deep_zoom(c, p, W, H, &norm, rows_local, rgba, nu, iter, stream, reserve_only, out_frame);

    const int fractal = p->fractal_type;                      /* 0 Mandelbrot, 1 Julia, 2 Burning Ship */
    const bool julia = fractal == FR_FRACTAL_JULIA;
    const bool uv_map = fractal != FR_FRACTAL_MANDELBROT;     /* julia.comp:325 / burning_ship.comp:393 viewport map */
    const bool f64 = p->precision == FR_PRECISION_F64;
    /* The Mandelbrot shader's effects need nothing along the orbit: stripe shading reads the z of the sample's last update, and
     * the orbit trap's minimum is the constant 0 (see shade_stripes: the first update makes z = c, and distToC is part of
     * the minimum).  Such frames take the lean tile pass and the lane pool in their code-3 instantiations instead of the
     * effects variant's lockstep run to max_iter with four running minima -- 8x8 sub-tiles, two per trip.  (Burning Ship's
     * trap and stripe sums are real accumulators: the effects variant keeps them.) */
    const bool m_effects = fractal == FR_FRACTAL_MANDELBROT && needs_effects(p) &&
                              c->tune_stripes != 1u && c->tune_tile_kernel != 1u && (c->tune_shape == 0u || c->tune_shape == 3u) &&
                              c->tune_tile_pixels != 1u;
    const bool m_effects_lean = m_effects && p->antialiasing_samples <= 1 && (norm.nparts == 1 || norm.rows_per_strip % 8u == 0u);
    const bool effects = needs_effects(p) && !m_effects_lean;
    const int max_iter = p->max_iterations;

    /* SSAA.  The sample loop of the general tile kernel runs a pixel's aa x aa samples one after the other, each to max_iter
     * in lockstep with the 63 other pixels of its sub-tile: no compaction, no lane pool -- on escape-dense views a sample
     * costs 1.6-1.8x what a pixel of the same view costs without SSAA (profiles/r04_ssaa_staged.txt).  Staged: the sample
     * grid is a frame of W aa x H aa "pixels" whose coordinates are the samples' (prepare_kernel writes them into the lean
     * kernels' tables), rendered through tile pass + lane pool into scratch planes, and ssaa_reduce_kernel averages.  Same
     * arithmetic per sample, same summation order: bit-identical planes.  Applies where the lean kernels do (no effects,
     * 8x8 sub-tiles, strips of whole sub-tile rows in sample space) and the sample grid is a legal frame (< 2^31 samples). */
    if (ssaa_of <= 1 && p->antialiasing_samples > 1 && (!needs_effects(p) || m_effects) && c->tune_ssaa != 1u) {   /* (striped sample
                                                                  grids too: their samples take the stripe instantiations) */
        const uint32_t aa = (uint32_t)p->antialiasing_samples;
        const uint64_t nsamples = (uint64_t)W * aa * (uint64_t)H * aa;
        const bool lean_ok = c->tune_tile_kernel != 1u && (c->tune_shape == 0u || c->tune_shape == 3u) &&
                             (norm.nparts == 1 || (norm.rows_per_strip * aa) % 8u == 0u);
        /* (measured: -32 to -61 % on every view and size but the C2 frame with cycle closing off, +-2 %; also where the sample
         * grid takes ONE pass -- 1080p at max_iter 256: -38 % -- because the lean kernel beats the general one.  The sample
         * planes and the survivor stream of the sample grid are context scratch: 16 B + up to 60 B per sample; above 2^29
         * samples -- 8192^2 at aa 3 -- the sample loop stays.) */
        const uint64_t band_cap = c->tune_ssaa_band ? c->tune_ssaa_band : (1ull << 29);
        const bool fits = nsamples <= band_cap || (c->tune_ssaa == 2u && !c->tune_ssaa_band);
        if (lean_ok && fits && nsamples < (1ull << 31) && (uint64_t)norm.rows_per_strip * aa <= 0xFFFFFFFFull)
            return enqueue_ssaa_staged(c, p, W, H, &norm, rows_local, rgba, nu, iter, stream, reserve_only, out_frame);
        /* A WHOLE frame whose sample grid is larger (a print export: 8192^2 at aa 3 is 6e8 samples, 46 GB of sample planes and
         * survivor stream) goes through the same scratch band by band: contiguous bands of whole sub-tile rows, each rendered as
         * the one strip of "part b of B" straight into the caller's planes (FR_LAYOUT_FRAME addressing), one after the other on
         * the stream.  Same samples, same sums: bit-identical (test_staged_ssaa_is_bit_identical_to_the_sample_loop).  Row-strip shards keep the sample
         * loop above the cap: a band of a shard is not a shard. */
        if (lean_ok && !fits && norm.nparts == 1 && !out_frame && (uint64_t)W * aa * 8u * aa <= band_cap) {
            const uint64_t per_row = (uint64_t)W * aa * aa;                     /* samples per pixel row */
            uint32_t band_rows = (uint32_t)(band_cap / per_row) & ~7u;           /* whole sub-tile rows, >= 8 by the test above */
            if (band_rows > H) band_rows = (H + 7u) & ~7u;
            const uint32_t nbands = (H + band_rows - 1u) / band_rows;
            const bool timed = c->timing;
            if (timed && !reserve_only) FR_HIP_TRY(hipEventRecord(c->ev_begin, stream));
            c->timing = false;                                                   /* one event pair around all bands */
            int st = FR_OK;
            for (uint32_t b = 0; b < nbands && st == FR_OK; ++b) {
                const fr_shard band = {b, nbands, band_rows};
                st = enqueue_ssaa_staged(c, p, W, H, &band, fr_shard_rows(&band, H), rgba, nu, iter, stream, reserve_only, true);
                if (reserve_only) break;                                         /* the first band is the largest */
            }
            c->timing = timed;
            if (st == FR_OK && timed && !reserve_only) {
                FR_HIP_TRY(hipEventRecord(c->ev_end, stream));
                c->have_timing = true;
            }
            return st;
        }
    }

    LaunchArgs a;
    fill_params(a, p);
    a.ssaa = ssaa_of > 1 ? ssaa_of : 0;
    a.res_w = (int32_t)res_w; a.res_h = (int32_t)res_h;
    a.W = (int32_t)W; a.H = (int32_t)H;
    a.rows_local = (int32_t)rows_local;
    a.part = (int32_t)norm.part; a.nparts = (int32_t)norm.nparts; a.rows_per_strip = (int32_t)norm.rows_per_strip;
    a.out_frame = out_frame ? 1 : 0;
    a.rgba = reinterpret_cast<float4*>(rgba);
    a.nu = nu; a.iter = iter;
    a.log2_tab = c->log2_tab;

    /* escape is absorbing (see escape_run): bailout^2 in [4.5, 1e12], and for Julia |c| <= bailout;
     * Mandelbrot lanes with |c| > bailout retire at i = 0 inside the first, tested block */
    {
        const double B2 = f64 ? (double)p->bailout * (double)p->bailout
                              : (double)(p->bailout * p->bailout);
        const double c2 = a.julia_cx * a.julia_cx + a.julia_cy * a.julia_cy;
        a.fast_ok = (B2 >= 4.5 && B2 <= 1e12 && (!julia || c2 <= B2)) ? 1 : 0;
        /* 4 bailout^2 as the kernels form it: B * B in the kernel's precision, times 4 (exact) */
        const float b2f = p->bailout * p->bailout;
        a.b2x4_d = 4.0 * ((double)p->bailout * (double)p->bailout);
        a.b2x4_f = 4.0f * b2f;
    }

    /* host-prepared reciprocals; the divide-free viewport map is enabled only when verified exact */
    a.inv_w_d = 1.0 / (double)W;  a.inv_h_d = 1.0 / (double)H;
    a.inv_w_f = 1.0f / (float)W;  a.inv_h_f = 1.0f / (float)H;
    a.aspect_d = (double)W / (double)H;
    a.aspect_f = (float)W / (float)H;
    a.exact_div_ok = exact_division_ok(c, W, H, uv_map, f64) ? 1 : 0;

    int bounds[kMaxStages];
    int nstage = plan_stages(c, p, effects, (size_t)rows_local * W, false, bounds);
    const bool staged = nstage > 1;
    c->last_pool_closing = -1;
    /* does this render's lane pool look for cycles?  (fr_ctx_reserve sizes for the pool that looks: the shorter tile pass
     * leaves more survivors.) */
    const bool pool_looks = staged && !m_effects_lean &&           /* (a closed cycle has no z after max_iter updates) */
                            (reserve_only ? c->tune_periodicity >= 0 : pool_wants_cycle_closing(c, p, W, rows_local));
    if (staged && !pool_looks) nstage = plan_stages(c, p, effects, (size_t)rows_local * W, true, bounds);   /* (still two passes) */
    /* survivor-stream writers move to the next region after every block: the regions come out equally
     * long with the same mix of blocks, so the reading pass is balanced with little stealing (measured,
     * profiles/r01_region_rotation.txt: C2 0.883 -> 0.831 ms, C3 0.598 -> 0.539 ms; regions by XCD = 1) */
    const uint32_t rotate_regions = c->tune_stream_rotate == 1u ? 0u : 1u;
    const int shape = c->tune_shape ? (int)c->tune_shape : 3;

    /* bounded, cheap items: the staged tile pass, and an unstaged pass whose samples run at most 128 updates
     * (measured at max_iter <= 32: 0.31 ms with short runs -- the queue words saturate -- 0.17 ms with long) */
    const int aa1 = p->antialiasing_samples > 1 ? p->antialiasing_samples : 1;
    const bool bounded = staged || (!effects && (long long)max_iter * aa1 * aa1 <= 128);
    /* items of moderate cost (an unstaged pass below the staging threshold): short runs as for unbounded items, but
     * the waves stop at their home shard -- the blocks of 16 sub-tiles dealt round-robin keep the shards level */
    const bool moderate = !staged && !effects && (long long)max_iter * aa1 * aa1 < 768;
    uint32_t grid = 0, waves_per_shard = 0;
    /* (fp32 only: the fp64 instantiation's 82 VGPRs leave room for 5 waves per SIMD) */
    const bool lean_staged = !f64 && staged && !effects && p->antialiasing_samples <= 1 && shape == 3 && c->tune_tile_kernel != 1u &&
                             (norm.nparts == 1 || norm.rows_per_strip % 8u == 0u);
    const QueueArgs tq = plan_tile_queue(c, W, rows_local, shape, bounded, moderate, lean_staged, &grid, &waves_per_shard);
    auto clamp_shift = [&](int v) { v += c->tune_shift_bias; return (uint32_t)(v < 0 ? 0 : (v > 31 ? 31 : v)); };
    c->last_grid = grid;

    /* ---- survivor streams in context scratch ------------------------------------------------------------ */
    /* the pool / stream kernels hold 6 workgroups per CU; their blocks are latency bound (dequeue -> record
     * loads -> iterate -> scattered stores), so run all of them */
    uint32_t sgrid = (uint32_t)c->compute_units * (c->tune_stream_wg_per_cu ? c->tune_stream_wg_per_cu : 6u);
    {   /* small frames: at most one wave per 8 sub-tiles of the frame (every survivor block holds 64 records, and
         * a frame rarely leaves more than a quarter of its pixels alive after the tile pass: ~2 blocks per wave;
         * 1080p at max_iter 1024: 0.144 ms with 6 workgroups per CU, 0.128 ms with the 4 this cap gives) */
        const uint32_t per_wg = c->tune_pool_items_per_wg ? c->tune_pool_items_per_wg : 32u;
        const uint32_t cap = (tq.n_items + per_wg - 1u) / per_wg;
        if (sgrid > cap) sgrid = cap < 1u ? 1u : cap;
    }
    /* the survivor streams have as many regions as the tile queue has shards ("regions" overrides) */
    const uint32_t nregions = c->tune_regions ? c->tune_regions : (1u << tq.ns_log2);
    const uint32_t nregions_log2 = nregions == (uint32_t)kMaxShards ? 6u : 3u;
    uint32_t region_blocks = 0;
    if (staged) {
        const int st = reserve_stream(c, (size_t)rows_local * W, julia ? 2 : 4, f64, grid > sgrid ? grid : sgrid, nregions,
                                      &region_blocks);
        if (st != FR_OK) return st;
        if (c->debug_region_blocks && c->debug_region_blocks < region_blocks) region_blocks = c->debug_region_blocks;
    }
    /* the lean tile kernel: every one-sample render without effects on 8x8 sub-tiles whose row strips (if sharded) are
     * whole sub-tile rows; "tile_kernel" = 1 keeps the general kernel (tests compare the two bitwise) */
    const bool lean = !effects && p->antialiasing_samples <= 1 && shape == 3 && c->tune_tile_kernel != 1u &&
                      (norm.nparts == 1 || norm.rows_per_strip % 8u == 0u);
    if (ssaa_of > 1 && !lean) return fr_set_error(FR_ERR_INTERNAL, "staged SSAA reached a render the lean tile kernel does not serve");
    if (lean) {
        const size_t need = ((size_t)W + H) * sizeof(double);
        if (need > c->coord_bytes) {
            if (c->coord_buf) { (void)hipFree(c->coord_buf); c->coord_buf = nullptr; }
            c->coord_bytes = 0;
            FR_HIP_TRY(hipMalloc(&c->coord_buf, need));
            c->coord_bytes = need;
        }
        a.xs = c->coord_buf;
        a.yds = (uint8_t*)c->coord_buf + (size_t)W * (f64 ? sizeof(double) : sizeof(float));
    }
    if (reserve_only) return FR_OK;

    /* the frame's device time (fr_ctx_last_kernel_ms) includes the small launch that prepares it */
    if (c->timing) FR_HIP_TRY(hipEventRecord(c->ev_begin, stream));
    /* the lean tile pass prepares its own control block and coordinate tables (lean_prologue: its first workgroups, the others
     * wait on a word keyed by a per-context epoch) -- except on a capturing stream: a replayed launch would carry a stale epoch */
    bool in_kernel_prologue = false;
    if (lean && c->tune_prepare != 1u) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
        in_kernel_prologue = cap == hipStreamCaptureStatusNone;
    }
    if (in_kernel_prologue) {
        if (c->prologue_epoch >= 0x0FFFFFF0u) {                 /* 28 bits: start over behind a cleared word */
            FR_HIP_TRY(hipMemsetAsync(c->d_ctrl + kReadyWord, 0, sizeof(uint32_t), stream));
            c->prologue_epoch = 0;
        }
        const Feedback fb = feedback_of(c);
        a.pro_ready = c->d_ctrl + kReadyWord;
        a.pro_epoch = ++c->prologue_epoch << 4;
        a.pro_ctrl = c->d_ctrl;
        a.pro_ctrl_words = (uint32_t)((size_t)nstage * kStageWords);
        a.pro_fb_flag = fb.dev_flag; a.pro_fb_host = fb.host_word; a.pro_prev_seq = fb.prev_seq;
        uint32_t n = 8;
        while (n > grid) n >>= 1;
        a.pro_n = n ? n : 1u;
    } else if (lean) {
        hipError_t ep = by_variant(fractal, f64, [&](auto t, auto f) {
            return launch_prepare<decltype(t), decltype(f)::value>(stream, a, c->d_ctrl, (uint32_t)((size_t)nstage * kStageWords), feedback_of(c)); });
        if (ep != hipSuccess) return fr_set_error(FR_ERR_HIP, "prepare kernel launch failed: %s", hipGetErrorString(ep));
    } else {
        FR_HIP_TRY(clear_control_block(c, stream, nstage));
    }

    /* ---- tile pass ---------------------------------------------------------------------------------- */
    a.q = tq;
    a.q.heads = stage_heads(c, 0);
    a.i0 = 0;
    a.i1 = bounds[0];
    a.exit_from = 0;
    a.exit_cost = 0u;
    if (staged && lean && c->tune_tile_exit != 1u) {
        a.exit_cost = c->tune_tile_exit ? c->tune_tile_exit : kTileExitCost;
        /* not in the first half of the budget (C5, b0 192: 4.22 -> 4.09 ms leaving from 64, 4.06 from 96; C2, b0 96: +-0.3 %
         * whatever the rule -- profiles/r04_tile_occupancy_exit.txt) */
        const uint32_t half = ((uint32_t)bounds[0] / 2u + 15u) / 16u * 16u;
        a.exit_from = (int32_t)(c->tune_tile_exit_from ? c->tune_tile_exit_from : (half > kTileExitFrom ? half : kTileExitFrom));
    }
    a.out.overflow = c->overflow_dev;                           /* (the prologue's timeout reports through it too) */
    if (staged) {
        a.out.base = (uint8_t*)c->stream_buf;
        a.out.n_blocks = stage_counter(c, 0);
        a.out.region_blocks = region_blocks;
        a.out.rotate = rotate_regions;
        a.out.nregions = nregions;
        a.out.overflow = c->overflow_dev;
    }
    a.diag = c->diag;
    hipError_t e;
    if (effects) {
        e = by_variant(fractal, f64, [&](auto t, auto f) {
            constexpr int F = decltype(f)::value == 1 ? 0 : decltype(f)::value;      /* Julia has no effects variant */
            return launch_tile<decltype(t), F, true>(shape, dim3(grid), stream, a); });
    } else {
        /* a pass that runs its samples to max_iter closes cycles in escape_run: SSAA (any shape; always compiled in) and
         * the one-sample kernel with 8x8 sub-tiles (its PERIOD instantiation, launch_tile) */
        a.period_window = !staged && !m_effects_lean ? period_window(c) : 0u;   /* a staged tile pass hands its survivors on */
        if (lean && m_effects_lean)
            e = f64 ? launch_tile_lean_stripes<double>(dim3(grid), stream, a) : launch_tile_lean_stripes<float>(dim3(grid), stream, a);
        else if (lean)
            e = by_variant(fractal, f64, [&](auto t, auto f) {
                return launch_tile_lean<decltype(t), decltype(f)::value>(c->tune_tile_pixels == 1u ? 1 : 2, dim3(grid), stream, a); });
        else
            e = by_variant(fractal, f64, [&](auto t, auto f) {
                return launch_tile<decltype(t), decltype(f)::value, false>(shape, dim3(grid), stream, a); });
    }
    if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "tile kernel launch failed: %s", hipGetErrorString(e));

    /* ---- lane-pool pass: the survivors, to max_iter ------------------------------------------------------ */
    if (staged) {
        a.i0 = bounds[0];
        a.i1 = max_iter;
        a.pro_ready = nullptr;
        a.in = a.out;                                           /* what the tile pass wrote */
        a.in.n_blocks = stage_counter(c, 0);
        memset(&a.out, 0, sizeof(a.out));                       /* the pool pass runs everything out ... */
        a.out.overflow = c->overflow_dev;                       /* ... and reports a stretch loop that will not end */
        memset(&a.q, 0, sizeof(a.q));
        a.q.heads = stage_heads(c, 1);
        a.q.ns_log2 = nregions_log2;                            /* region r of the input stream is shard r of this queue */
        const uint32_t swps = (sgrid * 4u + nregions - 1) / nregions;
        a.q.run_shift = clamp_shift((int)ceil_log2(2u * swps));
        /* a lane-pool wave holds its claimed blocks as a private reserve and only stalls for a dequeue
         * once per reserve, so claim little and never ahead: what a wave has reserved when the queue
         * runs dry is exactly the tail of the pass (measured: 1-3 block runs + one run prefetched left
         * a 315 us drain on C2; a block of 64 interior records is ~60 us of work at 5 waves/SIMD).
         * ONE block per claim since round 3 (runs of 1-2 before): C2 -3.1 %, 1080p/1024 -0.9 %, C3 / C5 / C4 within
         * +-1 % (profiles/r03_b0_and_pool_tuning.txt) */
        a.q.run_min = c->tune_stream_run_min ? c->tune_stream_run_min : 1u;
        a.q.run_max = c->tune_stream_run_max ? c->tune_stream_run_max : 1u;
        if (a.q.run_min > a.q.run_max) a.q.run_min = a.q.run_max;
        {
            uint32_t probes = c->tune_stream_probes ? c->tune_stream_probes : (rotate_regions ? 4u : 0u);
            if (sgrid < 64u || sgrid < nregions) probes = 0;
            a.q.flags = probes << kQueueProbeShift;
        }
        a.diag = c->diag ? c->diag + c->diag_stride : nullptr;
        /* finished lanes wait until this many are idle: 24 in fp32, 16 in fp64 (where a retire + refill round is cheaper
         * relative to an update: C5 -1.5 %, 1080p/1024 -1.8 %, C2 / C4 unchanged; the fp32 dust +1.5 % with 16) */
        a.pool_refill_at = c->tune_pool_refill ? c->tune_pool_refill : (f64 ? 16u : 24u);
        if (a.pool_refill_at > 64u) a.pool_refill_at = 64u;
        a.period_window = pool_looks ? period_window(c) : 0u;
        c->last_pool_closing = a.period_window != 0u;
        a.closed_flag = c->d_ctrl + kFeedbackWord;
        if (m_effects_lean)
            e = f64 ? launch_pool_stripes<double>(dim3(sgrid), stream, a) : launch_pool_stripes<float>(dim3(sgrid), stream, a);
        else
            e = by_variant(fractal, f64, [&](auto t, auto f) {
                return launch_stream_pool<decltype(t), decltype(f)::value>(dim3(sgrid), stream, a); });
        if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "lane-pool kernel launch failed: %s", hipGetErrorString(e));
    }
    if (c->timing) FR_HIP_TRY(hipEventRecord(c->ev_end, stream));
    c->have_timing = c->timing;
    c->have_render = true;
    c->last_stream = stream;
    c->last_stages = nstage;
    ++c->render_seq;
    return FR_OK;
}

static int enqueue_ssaa_staged(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_shard* norm, uint32_t rows_local,
                               float* rgba, void* nu, int32_t* iter, hipStream_t stream, bool reserve_only, bool out_frame)
{
    const uint32_t aa = (uint32_t)p->antialiasing_samples;
    const uint32_t Ws = W * aa, Hs = H * aa;
    const size_t nsamp = (size_t)rows_local * aa * Ws;
    const bool f64 = p->precision == FR_PRECISION_F64;
    const size_t nu_elt = f64 ? 8 : 4;
    /* sample planes: colour, then nu, then iter -- only those the caller's planes need */
    const size_t off_nu = rgba ? nsamp * 16 : 0, off_iter = off_nu + (nu ? nsamp * nu_elt : 0), need = off_iter + (iter ? nsamp * 4 : 0);
    if (need > c->ssaa_bytes) {
        if (c->ssaa_buf) { (void)hipFree(c->ssaa_buf); c->ssaa_buf = nullptr; c->ssaa_bytes = 0; }
        FR_HIP_TRY(hipMalloc(&c->ssaa_buf, need));
        c->ssaa_bytes = need;
    }
    char* base = (char*)c->ssaa_buf;
    float* s_rgba = rgba ? (float*)base : nullptr;
    void* s_nu = nu ? (void*)(base + off_nu) : nullptr;
    int32_t* s_iter = iter ? (int32_t*)(base + off_iter) : nullptr;
    fr_params q = *p;
    q.antialiasing_samples = 1;
    q.flags &= ~FR_FLAG_POST_CHAIN;                            /* the post chain follows the average */
    const fr_shard sh = {norm->part, norm->nparts, norm->rows_per_strip * aa};
    int st = enqueue_render(c, &q, Ws, Hs, &sh, s_rgba, s_nu, s_iter, stream, reserve_only, false, (int)aa, W, H);
    if (st != FR_OK || reserve_only) return st;
    SsaaArgs r;
    r.s_rgba = reinterpret_cast<const float4*>(s_rgba); r.s_nu = s_nu; r.s_iter = s_iter;
    r.rgba = reinterpret_cast<float4*>(rgba); r.nu = nu; r.iter = iter;
    r.W = (int32_t)W; r.rows_local = (int32_t)rows_local; r.aa = (int32_t)aa;
    r.sx_outer = p->fractal_type == FR_FRACTAL_MANDELBROT ? 0 : 1;
    r.part = (int32_t)norm->part; r.nparts = (int32_t)norm->nparts; r.rows_per_strip = (int32_t)norm->rows_per_strip; r.out_frame = out_frame ? 1 : 0;
    r.flags = p->flags; r.brightness = p->color_brightness; r.saturation = p->color_saturation; r.contrast = p->color_contrast;
    r.julia_floors = p->fractal_type != FR_FRACTAL_MANDELBROT ? 1 : 0;
    size_t blocks = ((size_t)rows_local * W + kBlockThreads - 1) / kBlockThreads;
    const size_t cap = (size_t)c->compute_units * 16;
    if (blocks > cap) blocks = cap;
    if (f64) hipLaunchKernelGGL(ssaa_reduce_kernel<double>, dim3((uint32_t)blocks), dim3(kBlockThreads), 0, stream, r);
    else hipLaunchKernelGGL(ssaa_reduce_kernel<float>, dim3((uint32_t)blocks), dim3(kBlockThreads), 0, stream, r);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "SSAA reduce launch failed: %s", hipGetErrorString(e));
    if (c->timing) FR_HIP_TRY(hipEventRecord(c->ev_end, stream));      /* the frame's device time includes the average */
    return FR_OK;
}

static int check_common(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_output* out)
{
    if (!c) return fr_set_error(FR_ERR_INVALID_ARG, "ctx is NULL");
    if (!p || !out) return fr_set_error(FR_ERR_INVALID_ARG, "params/out is NULL");
    int st = fr_params_validate(p, W, H);
    if (st != FR_OK) return st;
    if (out->layout != FR_LAYOUT_PACKED && out->layout != FR_LAYOUT_FRAME)
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown fr_output.layout %d", out->layout);
    if (out->layout == FR_LAYOUT_FRAME && out->memory != FR_MEM_DEVICE)
        return fr_set_error(FR_ERR_INVALID_ARG, "FR_LAYOUT_FRAME needs FR_MEM_DEVICE planes");
    return FR_OK;
}

/* 1: this part owns rows and has a plane to write; 0: it owns no rows (nothing to do); < 0: error */
static int check_planes(const fr_shard* shard, uint32_t H, const fr_output* out)
{
    if (shard && shard->nparts && shard->part >= shard->nparts)
        return fr_set_error(FR_ERR_INVALID_ARG, "shard part %u >= nparts %u", shard->part, shard->nparts);
    if (fr_shard_rows(shard, H) == 0) return 0;
    if (!out->rgba && !out->nu && !out->iter)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_output has no plane to write");
    return 1;
}

extern "C" int fr_render_shard_async(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H,
                                     const fr_shard* shard, const fr_output* out, void* hip_stream)
{
    int st = check_common(c, p, W, H, out);
    if (st != FR_OK) return st;
    if (out->memory != FR_MEM_DEVICE)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_render_shard_async needs FR_MEM_DEVICE outputs");
    st = check_planes(shard, H, out);
    if (st <= 0) return st;
    FR_HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    st = enqueue_render(c, p, W, H, shard, out->rgba, out->nu, out->iter, s, false, out->layout == FR_LAYOUT_FRAME);
    if (st == FR_OK) c->render_on_user_stream = s != c->stream;
    return st;
}

extern "C" int fr_ctx_reserve(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_shard* shard)
{
    if (!c || !p) return fr_set_error(FR_ERR_INVALID_ARG, "fr_ctx_reserve: ctx/params is NULL");
    int st = fr_params_validate(p, W, H);
    if (st != FR_OK) return st;
    if (shard && shard->nparts && shard->part >= shard->nparts)
        return fr_set_error(FR_ERR_INVALID_ARG, "shard part %u >= nparts %u", shard->part, shard->nparts);
    FR_HIP_TRY(hipSetDevice(c->device));
    /* growing a buffer frees the old one, which a render still in flight may be reading: wait for the context's own
     * stream and, when the most recent render went to a caller's stream, for the event recorded behind it there */
    FR_HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->render_on_user_stream && c->have_render) FR_HIP_TRY(hipStreamSynchronize(c->last_stream));
    return enqueue_render(c, p, W, H, shard, nullptr, nullptr, nullptr, c->stream, true);
}

extern "C" int fr_ctx_check(fr_ctx* c)
{
    if (!c) return fr_set_error(FR_ERR_INVALID_ARG, "ctx is NULL");
    return check_overflow(c);
}

extern "C" int fr_ctx_synchronize(fr_ctx* c)
{
    if (!c) return fr_set_error(FR_ERR_INVALID_ARG, "ctx is NULL");
    FR_HIP_TRY(hipSetDevice(c->device));
    FR_HIP_TRY(hipStreamSynchronize(c->stream));
    return check_overflow(c);
}

/* fr_tuning.h: 1 / 0 = the lane pool of the most recent render looked / did not look for cycles, -1 = it had no lane pool */
extern "C" int fr_ctx_last_pool_closing(const fr_ctx* c) { return c ? c->last_pool_closing : -1; }

extern "C" void* fr_ctx_stream_handle(fr_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int fr_ctx_device(const fr_ctx* c) { return c ? c->device : -1; }

extern "C" int fr_render_shard(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H,
                               const fr_shard* shard, const fr_output* out)
{
    int st = check_common(c, p, W, H, out);
    if (st != FR_OK) return st;
    st = check_planes(shard, H, out);
    if (st <= 0) return st;
    FR_HIP_TRY(hipSetDevice(c->device));

    if (out->memory == FR_MEM_DEVICE) {
        st = enqueue_render(c, p, W, H, shard, out->rgba, out->nu, out->iter, c->stream, false, out->layout == FR_LAYOUT_FRAME);
        if (st != FR_OK) return st;
        c->render_on_user_stream = false;
        FR_HIP_TRY(hipStreamSynchronize(c->stream));
        return check_overflow(c);
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
    const size_t nu_bytes = (p->precision == FR_PRECISION_F64 && p->fractal_type != FR_FRACTAL_DEEP_ZOOM) ? 8 : 4;
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
    c->render_on_user_stream = false;
    if (out->rgba) FR_HIP_TRY(hipMemcpyAsync(out->rgba, d_rgba, npx * 16, hipMemcpyDeviceToHost, c->stream));
    if (out->nu) FR_HIP_TRY(hipMemcpyAsync(out->nu, d_nu, npx * nu_bytes, hipMemcpyDeviceToHost, c->stream));
    if (out->iter) FR_HIP_TRY(hipMemcpyAsync(out->iter, d_iter, npx * 4, hipMemcpyDeviceToHost, c->stream));
    FR_HIP_TRY(hipStreamSynchronize(c->stream));
    return check_overflow(c);
}

extern "C" int fr_render(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const fr_output* out)
{
    return fr_render_shard(c, p, W, H, nullptr, out);
}

/* ---- 8-bit export ------------------------------------------------------------------------------ */
/* 1 when a frame's colour plane is a function of its smooth-count plane alone (fr_colorize_async) */
extern "C" int fr_colorize_supported(const fr_params* p)
{
    if (!p) return 0;
    if (p->fractal_type != FR_FRACTAL_MANDELBROT && p->fractal_type != FR_FRACTAL_JULIA &&
        p->fractal_type != FR_FRACTAL_BURNING_SHIP) return 0;
    if (needs_effects(p) || p->antialiasing_samples > 1) return 0;
    /* every escaped sample must have nu < max_iter, so that nu == max_iter identifies the interior:
     * Mandelbrot nu = i + 1 - log2(log2|z|) needs |z| > 2 with margin; the Julia form subtracts
     * log2(log|z|^2 / log B) > 1 for any B > 1 */
    if (p->fractal_type == FR_FRACTAL_MANDELBROT ? !(p->bailout >= 2.5f) : !(p->bailout >= 1.25f)) return 0;
    /* ... and must REPRESENT it: in fp32 a sample escaping at i = max_iter - 1 has nu = RN(max_iter - mu), mu > 0.3,
     * which rounds up to max_iter itself once the float spacing at max_iter reaches 0.5 (max_iter >= 2^23; mu can be
     * as small as ~0.3 at the smallest bailouts allowed above, so stop a binade earlier): it would be recoloured as
     * interior.  fp64 has no such limit at any max_iter the library accepts (<= 2^24). */
    if (p->precision == FR_PRECISION_F32 && p->max_iterations > (1 << 22)) return 0;
    return 1;
}

extern "C" int fr_colorize_async(fr_ctx* c, const fr_params* p, uint64_t n_pixels, const void* nu, float* rgba,
                                 void* hip_stream)
{
    if (!c || !p || !nu || !rgba) return fr_set_error(FR_ERR_INVALID_ARG, "fr_colorize_async: NULL argument");
    int st = fr_params_validate(p, 1, 1);
    if (st != FR_OK) return st;
    if (!fr_colorize_supported(p))
        return fr_set_error(FR_ERR_UNSUPPORTED, "colour is not a function of the smooth count for these parameters "
                                                "(effects, SSAA, Deep_Zoom or a small bailout)");
    if (n_pixels == 0) return FR_OK;
    FR_HIP_TRY(hipSetDevice(c->device));
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (!hip_stream) FR_HIP_TRY(order_after_last_render(c, stream));
    LaunchArgs a;
    fill_params(a, p);
    size_t blocks = ((size_t)n_pixels + kBlockThreads - 1) / kBlockThreads;
    const size_t cap = (size_t)c->compute_units * 8;
    if (blocks > cap) blocks = cap;
    const size_t n = (size_t)n_pixels;
    float4* out = reinterpret_cast<float4*>(rgba);
    hipError_t e = by_variant(p->fractal_type, p->precision == FR_PRECISION_F64, [&](auto t, auto f) {
        using T = decltype(t);
        hipLaunchKernelGGL((colorize_kernel<T, decltype(f)::value>), dim3((uint32_t)blocks), dim3(kBlockThreads), 0,
                           stream, a, reinterpret_cast<const T*>(nu), out, n);
        return hipGetLastError(); });
    if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "colorize launch failed: %s", hipGetErrorString(e));
    return FR_OK;
}

static size_t export_blocks(const fr_ctx* c, size_t npx)
{
    size_t blocks = ((npx + 3) / 4 + kBlockThreads - 1) / kBlockThreads;     /* four pixels per thread */
    const size_t cap = (size_t)c->compute_units * 8;
    return blocks > cap ? cap : (blocks < 1 ? 1 : blocks);
}

/* four pixels per thread (dword / dwordx2 stores) need a width that is a multiple of four AND an output pointer aligned
 * for those stores -- a caller may hand in a sub-buffer at any byte offset; otherwise one pixel per thread */
static hipError_t launch_export(const fr_ctx* c, const float4* in, uint8_t* out, uint32_t W, uint32_t H, int through_half,
                                hipStream_t s)
{
    const int quads_ok = (W & 3u) == 0u && ((uintptr_t)out & 3u) == 0u;
    const dim3 grid((uint32_t)export_blocks(c, (size_t)W * H));
    if (through_half)
        hipLaunchKernelGGL(export_rgb8_kernel<true>, grid, dim3(kBlockThreads), 0, s, in, out, (int)W, (int)H, quads_ok, (const float2*)c->export8_thr);
    else
        hipLaunchKernelGGL(export_rgb8_kernel<false>, grid, dim3(kBlockThreads), 0, s, in, out, (int)W, (int)H, quads_ok, (const float2*)c->export8_thr);
    return hipGetLastError();
}
static hipError_t launch_export(const fr_ctx* c, const float4* in, uint16_t* out, uint32_t W, uint32_t H, int through_half,
                                hipStream_t s)
{
    const int quads_ok = (W & 3u) == 0u && ((uintptr_t)out & 7u) == 0u;
    const dim3 grid((uint32_t)export_blocks(c, (size_t)W * H));
    if (through_half)
        hipLaunchKernelGGL(export_rgb16_kernel<true>, grid, dim3(kBlockThreads), 0, s, in, out, (int)W, (int)H, quads_ok);
    else
        hipLaunchKernelGGL(export_rgb16_kernel<false>, grid, dim3(kBlockThreads), 0, s, in, out, (int)W, (int)H, quads_ok);
    return hipGetLastError();
}

/* both export entry points: OUT = uint8_t (8-bit animation frames) or uint16_t (16-bit print export) */
template <typename OUT>
static int export_sync(fr_ctx* c, const float* rgba, uint32_t W, uint32_t H, OUT* out, int32_t memory, int32_t through_half,
                       const char* what)
{
    if (!c || !rgba || !out || W == 0 || H == 0 || (uint64_t)W * H >= (1ull << 31))
        return fr_set_error(FR_ERR_INVALID_ARG, "%s: bad argument (NULL pointer, empty frame or 2^31 pixels and more)", what);
    FR_HIP_TRY(hipSetDevice(c->device));
    const size_t npx = (size_t)W * H;
    const float4* d_in = reinterpret_cast<const float4*>(rgba);
    OUT* d_out = out;
    if (memory == FR_MEM_HOST) {
        const size_t need = npx * 16 + npx * 3 * sizeof(OUT);
        if (need > c->scratch_bytes) {
            if (c->scratch) { (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
            FR_HIP_TRY(hipMalloc(&c->scratch, need));
            c->scratch_bytes = need;
        }
        FR_HIP_TRY(hipMemcpyAsync(c->scratch, rgba, npx * 16, hipMemcpyHostToDevice, c->stream));
        d_in = reinterpret_cast<const float4*>(c->scratch);
        d_out = reinterpret_cast<OUT*>((uint8_t*)c->scratch + npx * 16);
    } else if (memory == FR_MEM_DEVICE) {
        /* the plane may come from a render this context enqueued on a caller's stream */
        FR_HIP_TRY(order_after_last_render(c, c->stream));
    } else {
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown memory kind %d", memory);
    }
    hipError_t e = launch_export(c, d_in, d_out, W, H, (int)through_half, c->stream);
    if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "export launch failed: %s", hipGetErrorString(e));
    if (memory == FR_MEM_HOST)
        FR_HIP_TRY(hipMemcpyAsync(out, d_out, npx * 3 * sizeof(OUT), hipMemcpyDeviceToHost, c->stream));
    FR_HIP_TRY(hipStreamSynchronize(c->stream));
    return check_overflow(c);
}

template <typename OUT>
static int export_async(fr_ctx* c, const float* rgba, uint32_t W, uint32_t H, OUT* out, int32_t through_half, void* hip_stream,
                        const char* what)
{
    if (!c || !rgba || !out || W == 0 || H == 0 || (uint64_t)W * H >= (1ull << 31))
        return fr_set_error(FR_ERR_INVALID_ARG, "%s: bad argument (NULL pointer, empty frame or 2^31 pixels and more)", what);
    FR_HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (!hip_stream) FR_HIP_TRY(order_after_last_render(c, s));
    hipError_t e = launch_export(c, reinterpret_cast<const float4*>(rgba), out, W, H, (int)through_half, s);
    if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "export launch failed: %s", hipGetErrorString(e));
    return FR_OK;
}

extern "C" int fr_export_rgb8(fr_ctx* c, const float* rgba, uint32_t W, uint32_t H,
                              uint8_t* rgb8, int32_t memory, int32_t through_half)
{
    return export_sync<uint8_t>(c, rgba, W, H, rgb8, memory, through_half, "fr_export_rgb8");
}

extern "C" int fr_export_rgb16(fr_ctx* c, const float* rgba, uint32_t W, uint32_t H,
                               uint16_t* rgb16, int32_t memory, int32_t through_half)
{
    return export_sync<uint16_t>(c, rgba, W, H, rgb16, memory, through_half, "fr_export_rgb16");
}

extern "C" int fr_export_rgb8_async(fr_ctx* c, const float* rgba, uint32_t W, uint32_t H, uint8_t* rgb8,
                                    int32_t through_half, void* hip_stream)
{
    return export_async<uint8_t>(c, rgba, W, H, rgb8, through_half, hip_stream, "fr_export_rgb8_async");
}

extern "C" int fr_export_rgb16_async(fr_ctx* c, const float* rgba, uint32_t W, uint32_t H, uint16_t* rgb16,
                                     int32_t through_half, void* hip_stream)
{
    return export_async<uint16_t>(c, rgba, W, H, rgb16, through_half, hip_stream, "fr_export_rgb16_async");
}

/* RenderFrameCallback body: src/vk_engine.cpp:1181-1418 (render -> readback -> CPU tonemap/flip -> PNG),
 * with everything up to the 3 B/pixel readback on the GPU. */
extern "C" int fr_render_frame_png(fr_ctx* c, const fr_params* p, uint32_t W, uint32_t H, const char* path)
{
    if (!c || !p || !path) return fr_set_error(FR_ERR_INVALID_ARG, "fr_render_frame_png: NULL argument");
    int st = fr_params_validate(p, W, H);
    if (st != FR_OK) return st;
    FR_HIP_TRY(hipSetDevice(c->device));
    const size_t npx = (size_t)W * H;
    const size_t need = npx * 16 + npx * 3;
    if (need > c->frame_bytes) {
        if (c->frame_buf) { (void)hipFree(c->frame_buf); c->frame_buf = nullptr; c->frame_bytes = 0; }
        FR_HIP_TRY(hipMalloc(&c->frame_buf, need));
        c->frame_bytes = need;
    }
    fr_params q = *p;
    if (q.fractal_type != FR_FRACTAL_DEEP_ZOOM) q.flags |= FR_FLAG_POST_CHAIN;   /* the storage image holds the post-chained colour;
                                                                                     the deep-zoom shader has no post chain */
    float* d_rgba = (float*)c->frame_buf;
    uint8_t* d_rgb8 = (uint8_t*)c->frame_buf + npx * 16;
    st = enqueue_render(c, &q, W, H, nullptr, d_rgba, nullptr, nullptr, c->stream);
    if (st != FR_OK) return st;
    c->render_on_user_stream = false;
    hipError_t e = launch_export(c, reinterpret_cast<const float4*>(d_rgba), d_rgb8, W, H, 1, c->stream);
    if (e != hipSuccess) return fr_set_error(FR_ERR_HIP, "export launch failed: %s", hipGetErrorString(e));
    uint8_t* host = (uint8_t*)malloc(npx * 3);
    if (!host) return fr_set_error(FR_ERR_NOMEM, "out of host memory");
    hipError_t ce = hipMemcpyAsync(host, d_rgb8, npx * 3, hipMemcpyDeviceToHost, c->stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(c->stream);
    if (ce != hipSuccess) { free(host); return fr_set_error(FR_ERR_HIP, "readback failed: %s", hipGetErrorString(ce)); }
    st = check_overflow(c);
    if (st == FR_OK) st = fr_write_png(path, W, H, 8, host, nullptr, 0, 0);
    free(host);
    return st;
}
