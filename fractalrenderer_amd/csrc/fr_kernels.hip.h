/*
 * fr_kernels.hip.h -- hand-written CDNA4 (gfx950) escape-time kernels.
 *
 * What is computed is the per-pixel loop of shaders/mandelbrot.comp:147-208, shaders/julia.comp:216-249
 * and shaders/burning_ship.comp:217-309 (z <- z^2 + c, update-then-test, smooth iteration count,
 * palette), in fp32 (what the reference does) or fp64, plus the reference's perturbation shader
 * (shaders/test_deep_zoom.comp).  How it is computed is MI355X-first and shares nothing with the
 * reference's 16x16-workgroup GLSL dispatch:
 *
 *   - PREPARE (prepare_kernel): a small launch in front of a render: control block zeroed, the frame's W + H
 *     coordinates (the viewport map is separable) written as two tables.
 *   - TILE PASS (tile_lean_kernel; tile_kernel is its general form: SSAA, effects, other sub-tile shapes): a wave owns
 *     "sub-tiles" of 64 pixels (8x8; the general kernel also 16x4 or 64x1 -- every lane row is a whole number of
 *     128-byte lines of the row-major RGBA-f32 frame, so stores are full-line coalesced), the lean kernel two of them
 *     per trip (two pixels per lane).  A PERSISTENT grid pulls runs of sub-tiles from a queue of 8 or 64 shards (heads
 *     128 B apart, home shard by XCD); it runs only the first b0 iterations of every pixel.  Pixels that escaped are
 *     shaded and stored; pixels still alive are COMPACTED -- through a per-wave LDS ring into dense blocks of 64
 *     survivor records {pixel, z, c} in HBM, the writers rotating over 8 or 64 regions.
 *   - LANE POOL (pool_kernel): the survivors, to max_iter.  Persistent LANES: a lane that finishes is
 *     refilled with the next record, so waves stay full whatever the spread of escape times and the
 *     frame balances at pixel granularity.  Sub-tile cost varies 100x (a few iterations outside the
 *     set, max_iter inside) and a 64-pixel wave runs as long as its slowest lane (25 % lane occupancy
 *     on a Julia dust): after the tile pass every work item has bounded cost, and the expensive
 *     remainder runs dense.  PERIOD variant ("periodicity" option, the library's default): a lane whose
 *     state returns to its own snapshot is on a cycle and is retired as interior at once -- exact, the planes stay
 *     byte-identical, but fewer iterations run than the reference executes (bench.py's headline switches it off).
 *   - the iteration index is wave-uniform and lives in SGPRs; a lane that escapes records
 *     (i, |z|^2) and is parked at the fixed point z = 0, c = 0 (lean tile kernel: its escape threshold becomes NaN
 *     instead), so no per-lane "active" predicate
 *     exists in the loop; a tile wave leaves the loop as soon as the ballot of finished lanes is full
 *     (wave-uniform early-out); where escape is absorbing, waves run UNCHECKED blocks of 16 updates
 *     (6 VALU ops each); a dirty block is rolled back and replayed tested (tile pass), or -- lane pool -- only its escaped
 *     lanes are, later, 64 of them at a time from a ring in LDS (DeferRing), while the wave runs on unchecked;
 *   - scalar instructions share one issue port per CU: the tested loops carry a countdown and one
 *     vector-compare branch, nothing else ("Scalar issue is a roofline too");
 *   - the arithmetic is the reference's, one rounding per operation, NO contraction of the
 *     as-written a*b+c (file is built with -ffp-contract=off).  Where an fma is written explicitly
 *     it multiplies by an exact power of two, which rounds exactly like the as-written
 *     two-operation form (see "scaled-imaginary form" below);
 *   - the palette knot table and the viewport constants are staged once per workgroup into LDS (per-lane lookups);
 *     what steers control flow is read from the kernel arguments (wave-uniform), and arguments needed rarely are
 *     re-read where they are used (kargs()) instead of being held -- and spilled -- across the main loops.
 *
 * There is no dense contraction in this path: no MFMA.  The bound is fp64 (fp32) VALU issue; HBM
 * traffic is the write-once 16 B/pixel output plus 24-40 B per survivor record, written and read once.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "fr_internal.h"

#pragma clang fp contract(off)

namespace fr {

constexpr int kWave = 64;
constexpr int kBlockThreads = 256;
constexpr int kWavesPerBlock = kBlockThreads / kWave;
constexpr int kShards = 8;               /* XCDs: the smallest number of queue shards / stream regions             */
constexpr int kMaxShards = 64;           /* the largest: 8 per XCD.  A queue head (a stream's block counter) is ONE word that
                                          * every wave of its shard updates with a returning atomic, and same-address atomics
                                          * complete ~15 ns apart: 8 heads serve 0.5 G claims/s.  The staged tile pass makes 37 k
                                          * queue claims + 47 k block claims per C2 frame, the pool pass of C3 29 k in 0.2 ms with
                                          * 768 waves asking one word at once when it starts (first claim: 6 us); with 64 heads
                                          * the same traffic meets 8x the service rate.  Launches with unlimited stealing keep 8
                                          * (a wave probes every shard before it exits).                                      */
constexpr int kShardStrideWords = 32;    /* 128 B between heads                            */
constexpr int kShardBlock = 16;          /* sub-tiles are dealt to shards in blocks of 16  */
constexpr int kFastBlock = 16;           /* iterations per unchecked block                 */
constexpr uint32_t kQueueProbeShift = 4; /* bits 4-7: shards a wave may probe before giving up (0 = all of them) */
constexpr uint32_t kInvalidPixel = 0xFFFFFFFFu;
constexpr int kRingSlots = 128;          /* survivor ring capacity per wave (records)      */

/* One survivor stream in device scratch: blocks of 64 records, each block laid out
 * [pixel u32 x64][iterations done u32 x64][X x64][Yd x64]([cx x64][cyd x64] for Mandelbrot), so
 * every field is one coalesced access per wave. */
struct StreamRef {
    uint8_t* base;               /* kShards regions of region_blocks blocks each */
    uint32_t* n_blocks;          /* kShards device counters, kShardStrideWords apart: blocks appended per region */
    uint32_t region_blocks;      /* capacity of one region; 8 regions hold 1.5x the worst-case total */
    uint32_t rotate;             /* writers move to the next region after every block: equal-length regions
                                  * with the same mix of light and heavy blocks (the reader then needs no stealing) */
    uint32_t nregions;           /* kShards or kMaxShards (a power of two) */
    uint32_t* overflow;          /* host-mapped word of the context: set when all regions were found full (the host
                                  * sizes them for 1.5x the worst case, so this means a sizing bug) -- the records of that
                                  * block are lost and the render is reported as failed, never silently incomplete */
};

/* Queue geometry of one launch. */
struct QueueArgs {
    uint32_t* heads;             /* kShards words, kShardStrideWords apart, zeroed per render */
    uint32_t n_items;            /* sub-tiles (tile pass); stream pass reads its count from the stream */
    uint32_t nsx;                /* sub-tiles per sub-tile row                    */
    int32_t  nsx_shift;          /* log2(nsx) when nsx is a power of two, else -1 */
    uint32_t n_blk;              /* blocks of kShardBlock sub-tiles               */
    uint32_t run_shift;          /* run length = clamp(remaining >> run_shift, run_min, run_max) */
    uint32_t run_max, run_min;
    uint32_t flags;              /* bits 4-7: probe limit (kQueueProbeShift) */
    uint32_t ns_log2;            /* log2 of the number of shards: 3 (kShards) or 6 (kMaxShards) */
};

/* Kernel argument block (passed by value; lands in SGPRs / the scalar cache). */
struct LaunchArgs {
    /* viewport -- FractalState fields, src/fractal_state.h:18-21,29-30,36 */
    double center_x, center_y, zoom;
    double julia_cx, julia_cy;
    double log_bailout;          /* log(bailout) in the kernel's precision, computed on the host */
    float  bailout;
    int32_t max_iter;
    int32_t W, H;                /* whole frame                                  */
    int32_t rows_local;          /* rows this part renders                       */
    int32_t part, nparts, rows_per_strip;
    int32_t out_frame;           /* 1: the output planes are WHOLE-FRAME planes and the part's rows are stored in place
                                  * (frame row py), 0: the part's rows are stored packed (local row) -- FR_LAYOUT_FRAME */
    int32_t aa;
    int32_t ssaa;                /* > 1: this render IS the sample grid of an aa x aa supersampled frame of res_w x res_h pixels
                                  * (W = res_w * ssaa, H = res_h * ssaa): prepare_kernel writes the samples' coordinates */
    int32_t res_w, res_h;
    /* colouring */
    int32_t palette_mode;
    float color_offset, color_scale;
    int32_t interior_style;
    int32_t trap_enabled; float trap_radius;
    int32_t stripe_enabled; float stripe_density;
    float brightness, saturation, contrast;
    uint32_t flags;
    int32_t fast_ok;             /* escape is absorbing for every lane (see escape_run) */
    /* host-prepared reciprocals (each the correctly rounded 1/x in the kernel's precision) */
    int32_t exact_div_ok;        /* div_by() verified == IEEE divide for every column/row of this frame */
    double inv_w_d, inv_h_d;     /* RN(1/W), RN(1/H) in double */
    float  inv_w_f, inv_h_f;     /* RN(1/W), RN(1/H) in float  */
    double aspect_d;             /* (double)W / (double)H */
    float  aspect_f;             /* (float)W / (float)H   */
    double inv_max_iter;         /* 1 / max_iter (colour stage only) */
    double inv_log2_bailout;     /* 1 / log2(bailout) (Julia smooth count) */
    /* the same, and the colour knobs, in the precision the kernel reads them in: wave-uniform SGPR operands (a float
     * widened on the device is a VALU conversion per use, or two more live VGPRs) */
    float  inv_max_iter_f, inv_log2_bailout_f;
    double color_scale_d, color_offset_d;
    int32_t lib_log;             /* bailout <= 1: smooth count through the library log(), as written */
    /* stage: this launch runs iterations [i0, i1); i1 < max_iter -> unfinished pixels go to `out` */
    int32_t i0, i1;
    /* lean tile pass, staged: a trip whose live samples are few leaves before i1 (escape_run_lean); 0 = never */
    int32_t exit_from;
    uint32_t exit_cost;
    StreamRef in, out;
    uint32_t pool_refill_at;     /* lane pool: finished lanes wait until this many are idle */
    uint32_t period_window;      /* lane pool, PERIOD variant: iterations between the snapshots cycles are looked for against */
    /* outputs */
    float4* rgba;
    void* nu;
    int32_t* iter;
    QueueArgs q;
    const double2* log2_tab;     /* kLog2Entries x {1/m_i, log2 m_i}: see log2_tab() (built on the host, staged into LDS) */
    /* lean tile pass (tile_lean_kernel): per-column / per-row coordinates of the frame, written by prepare_kernel in
     * front of every render: xs[px] = Re of the pixel's point, yds[py] = 2 Im of it (T = the render's precision) */
    void* xs; void* yds;
    double b2x4_d; float b2x4_f; /* 4 bailout^2 in the kernel's precision */
    uint64_t* diag;              /* optional: 4 words per wave (t_start, t_end, items, dequeues) */
    uint32_t* closed_flag;       /* lane pool, PERIOD: set by a wave that closed a cycle (Feedback::dev_flag) */
    /* PROLOGUE of the lean tile pass (tile_lean_kernel): what prepare_kernel does in a launch of its own -- control words,
     * coordinate tables, the previous render's cycle-closing verdict -- done by the first `pro_n` workgroups of the tile pass
     * itself, everybody else waiting on one word: see lean_prologue_produce().  pro_ready == nullptr: prepare_kernel ran. */
    uint32_t* pro_ready;         /* (epoch << 4) | preparing workgroups that have finished, of the most recent render */
    uint32_t pro_epoch;          /* this render's epoch << 4 (per context, increasing) */
    uint32_t pro_n;              /* preparing workgroups: 1, 2, 4 or 8, <= the grid */
    uint32_t* pro_ctrl;          /* the control block: its live words (every kShardStrideWords-th of pro_ctrl_words) are zeroed */
    uint32_t pro_ctrl_words;
    uint32_t pro_prev_seq;       /* Feedback of the context (dev_flag, host_word, prev_seq) */
    uint32_t* pro_fb_flag;
    uint32_t* pro_fb_host;
    fr_palette_table pal;
};

/* The kernel arguments, re-read where they are used.  Everything a persistent kernel reads from its argument block is
 * loop-invariant, so the optimiser loads it all in front of the main loop -- and then spills what does not fit in the
 * ~100 SGPRs to VGPR lanes: v_writelane / v_readlane in the hot path, each a VALU issue slot (the general tile kernel
 * carries 29-48 spilled SGPRs).  A pointer that went through an empty asm statement is a new value every time: loads
 * through it stay where they are written (s_load from the scalar cache + one wait), and values needed once per run of
 * sub-tiles or once per 64 survivors stop occupying registers in between.  Valid in kernels whose ONLY parameter is the
 * LaunchArgs block (it then sits at offset 0 of the kernarg segment). */
typedef __attribute__((address_space(4))) const LaunchArgs* KArgs;
__device__ __forceinline__ KArgs kargs()
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (KArgs)p;
}

/* What one workgroup keeps in LDS: the palette knot table and the viewport constants. */
struct LdsBlock {
    fr_palette_table pal;
    double center_x, center_y, zoom, julia_cx, julia_cy, log_bailout;
    float bailout, color_offset, color_scale, trap_radius, stripe_density;
    float brightness, saturation, contrast;
    int32_t max_iter, W, H, aa;
    float interior_rgb[3];       /* colour of a sample that never escaped (stage_interior): the same for every such sample */
};

__device__ __forceinline__ void stage_constants(LdsBlock& S, const LaunchArgs& A)
{
    if (threadIdx.x == 0) {
        S.pal = A.pal;
        S.center_x = A.center_x; S.center_y = A.center_y; S.zoom = A.zoom;
        S.julia_cx = A.julia_cx; S.julia_cy = A.julia_cy; S.log_bailout = A.log_bailout;
        S.bailout = A.bailout; S.color_offset = A.color_offset; S.color_scale = A.color_scale;
        S.trap_radius = A.trap_radius; S.stripe_density = A.stripe_density;
        S.brightness = A.brightness; S.saturation = A.saturation; S.contrast = A.contrast;
        S.max_iter = A.max_iter; S.W = A.W; S.H = A.H; S.aa = A.aa;
    }
    __syncthreads();
}

template <typename T> struct Real;
template <> struct Real<double> {
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double floor(double x) { return ::floor(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double fabs(double x) { return ::fabs(x); }
    static __device__ __forceinline__ double fmin(double a, double b) { return ::fmin(a, b); }
    /* NOT inlined: the fp64 atan2 and sin of the stripe epilogue (once per pixel) held the whole effects variant of the tile
     * kernel at 156 VGPRs = 3 waves per SIMD; called, it takes 95 = 5, what the launcher's 5 workgroups per CU assume
     * (trap -2 %, stripes -4 %; same library routines on the same values: bit-identical) */
    static __device__ __attribute__((noinline)) double atan2(double y, double x) { return ::atan2(y, x); }
    static __device__ __attribute__((noinline)) double sin(double x) { return ::sin(x); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static constexpr double ln2() { return 0.693147180559945309417232121458; }
};
template <> struct Real<float> {
    static __device__ __forceinline__ float log(float x) { return ::logf(x); }
    static __device__ __forceinline__ float floor(float x) { return ::floorf(x); }
    static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
    static __device__ __forceinline__ float fabs(float x) { return ::fabsf(x); }
    static __device__ __forceinline__ float fmin(float a, float b) { return ::fminf(a, b); }
    static __device__ __forceinline__ float atan2(float y, float x) { return ::atan2f(y, x); }
    static __device__ __forceinline__ float sin(float x) { return ::sinf(x); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static constexpr float ln2() { return 0.693147180559945309417232121458f; }
};

/* ---- smooth-iteration helpers ---------------------------------------------------------------
 * The reference writes mu = log(log|z| / log 2) / log 2 (shaders/mandelbrot.comp:174-176), i.e.
 * log2(log2|z|).  Library log() costs 98 VALU instructions in fp64 on gfx950; the arguments here
 * are always positive, finite and normal (|z|^2 > bailout^2, and a log2 of it that is > 0), so a
 * range-reduced atanh series is enough.  Relative error < 4e-16, so nu agrees with the libm
 * evaluation to ~1e-14 (the parity tests assert 1e-9; the north-star bar is 1e-6). */
__device__ __forceinline__ double log2_pos(double x)
{
    int e = __builtin_amdgcn_frexp_exp(x);              /* x = m * 2^e, m in [0.5, 1) */
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;                                /* m in [sqrt(1/2), sqrt(2)) */
    e = low ? e - 1 : e;
    const double s = (m - 1.0) / (m + 1.0);             /* |s| <= 0.1716 */
    const double z = s * s;
    /* ln m = 2 s (1 + z/3 + z^2/5 + ... + z^9/19), truncation < 2e-17 */
    double p = 1.0 / 19.0;
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    p = __builtin_fma(p, z, 1.0);
    return __builtin_fma(s * p, 2.8853900817779268147 /* 2/ln 2 */, (double)e);
}
/* fp32: the hardware log2 (v_log_f32, ~1 ulp), which is also what a GLSL log() lowers to */
__device__ __forceinline__ float log2_pos(float x) { return __builtin_amdgcn_logf(x); }

/* The same function from a table in LDS: x = 2^e m, m in [0.5, 1); the top 7 mantissa bits select a bin whose midpoint
 * m_i has y_i = RN(1/m_i) and L_i = -log2(y_i) (to 64 bits on the host) in the table; r = m y_i - 1 (one fma, |r| < 2^-8)
 * and log2 x = (e + L_i) + log2(1 + r), the last term a degree-5 polynomial (truncation r^6 / (6 ln 2) < 9e-16).
 * 10 fp64-rate instructions + one ds_read_b128 against 33 for the series above: the two logs of the smooth count are
 * 24 of every 26 fp64 instructions the epilogue of a pixel costs.  Error < 2e-15 absolute: nu moves by ~1e-15. */
constexpr int kLog2Entries = 128;
__device__ __forceinline__ double log2_tab(const double2* __restrict__ tab, double x)
{
    const int e = __builtin_amdgcn_frexp_exp(x);
    const double m = __builtin_amdgcn_frexp_mant(x);                        /* [0.5, 1) */
    const uint32_t hi = (uint32_t)((uint64_t)__double_as_longlong(m) >> 32);
    const double2 en = tab[(hi >> 13) & (uint32_t)(kLog2Entries - 1)];      /* top 7 of the 52 mantissa bits */
    const double r = __builtin_fma(m, en.x, -1.0);
    double p = 0.28853900817779268;                                         /* +1/(5 ln 2) */
    p = __builtin_fma(p, r, -0.36067376022224085);                          /* -1/(4 ln 2) */
    p = __builtin_fma(p, r, 0.48089834696298783);                           /* +1/(3 ln 2) */
    p = __builtin_fma(p, r, -0.72134752044448170);                          /* -1/(2 ln 2) */
    p = __builtin_fma(p, r, 1.4426950408889634);                            /* +1/ln 2     */
    return __builtin_fma(p, r, (double)e + en.y);
}
/* what a kernel hands to shade(): the staged table (fp64) or nothing (fp32: hardware log2) */
template <typename T> struct LogTab;
template <> struct LogTab<double> {
    const double2* tab;
    __device__ __forceinline__ double log2(double x) const { return log2_tab(tab, x); }
};
template <> struct LogTab<float> {
    __device__ __forceinline__ float log2(float x) const { return log2_pos(x); }
};
/* stage the table into the workgroup's LDS (fp64 kernels): 128 x 16 B, one half entry per thread */
template <typename T>
__device__ __forceinline__ LogTab<T> stage_log2(double2* lds, const LaunchArgs& A)
{
    if constexpr (sizeof(T) == 8) {
        reinterpret_cast<double*>(lds)[threadIdx.x] = reinterpret_cast<const double*>(A.log2_tab)[threadIdx.x];
        __syncthreads();
        return LogTab<double>{lds};
    } else {
        (void)lds; (void)A;
        return LogTab<float>{};
    }
}

/* Marks a rarely taken branch: the optimiser otherwise evaluates BOTH sides of a cheap-looking if/else and
 * selects (found in the ISA: the 12-instruction IEEE fp64 divide ran next to its 3-op replacement for every
 * pixel).  An empty volatile asm cannot be speculated, so the branch stays a branch. */
__device__ __forceinline__ void cold_path() { asm volatile("" ::: "memory"); }

/* Correctly rounded a/b from y = RN(1/b) without a divide (Markstein): q = RN(a*y),
 * r = a - b*q exactly (fma), q' = RN(q + r*y).  The host enables this only after checking, for
 * every column and row coordinate of the frame, that q' equals the IEEE quotient. */
template <typename T>
__device__ __forceinline__ T div_by(T a, T b, T rb)
{
    const T q = a * rb;
    const T r = Real<T>::fma(-q, b, a);
    return Real<T>::fma(r, rb, q);
}

/* ---- colour stage (float, as the shaders) ---------------------------------------------- */

__device__ __forceinline__ float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

/* pow(u, e) for u in [0, 1], e > 0 as exp2(e * log2 u) on the hardware transcendentals -- the
 * lowering GPU drivers give GLSL pow().  u = 0 -> log2 = -inf -> exp2 = 0. */
__device__ __forceinline__ float pow01(float u, float e)
{
    return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(u));
}

/* get_palette_color (shaders/mandelbrot.comp:129-141, shaders/julia.comp:162-181), evaluated from the knot table;
 * like the shader's, its first step is t = fract(t).
 * U = the table in the kernel arguments (wave-uniform: SGPRs), L = its copy in LDS.  Everything that steers control
 * flow -- warp kind, segment count, break points -- is read from U: read from LDS they are VGPR values, every `if` on
 * them is compiled as a divergent branch (compare, exec save / restore, skip branch) and the break-point compares
 * need the loads first (found in the ISA: ~45 VALU + 20 SALU per call).  Only what is indexed by the lane's segment
 * (its lower break point, factor and two knots) comes from L.
 * Fire-style ramps return knot[last] on their last segment: mix(a, b, 0) == a exactly (knots are >= 0), so the
 * factor is forced to 0 there instead of branching. */
template <class PAL>
__device__ __forceinline__ void palette_eval(PAL& U, const fr_palette_table& L, float t, float rgb[3])
{
    const float u = t - floorf(t);
    const int warp = U.warp;
    if (warp == FR_WARP_GRAY) { rgb[0] = rgb[1] = rgb[2] = u; return; }
    float w = u;
    if (warp == FR_WARP_POW) {
        w = pow01(u, U.warp_exp);
    } else if (warp == FR_WARP_SMOOTHSTEP) {
        float s = clamp01((u - 0.0f) / (1.0f - 0.0f));
        w = s * s * (3.0f - 2.0f * s);
    }
    /* cascade "if (w < b1) .. else if (w < b2) .." == count of break points <= w */
    const int nseg = U.nseg;
    int seg = 0;
#pragma unroll
    for (int k = 1; k < 5; ++k) {
        const float lo = k < nseg ? U.seg_lo[k] : __builtin_inff();       /* wave-uniform select */
        seg += !(w < lo) ? 1 : 0;
    }
    const float d = w - L.seg_lo[seg];
    float k = d * L.seg_k[seg];
    if (U.any_div) {                        /* only julia.comp's cosmic and lava ramps divide (:87-101, :149-163) */
        cold_path();
        if (L.seg_div[seg]) k = d / L.seg_k[seg];
    }
    const int const_seg = U.last_const ? nseg - 1 : 99;                   /* wave-uniform */
    k = seg == const_seg ? 0.0f : k;
    const float* a = L.knot[seg];
    const float* b = L.knot[seg + 1];
    for (int c = 0; c < 3; ++c) rgb[c] = a[c] * (1.0f - k) + b[c] * k;   /* GLSL mix */
}

/* Narrowing of the palette argument: fp32 hands t to the palette untouched (the shader);
 * fp64 reduces with fract() in double first so the narrowing keeps the fractional part. */
__device__ __forceinline__ float pal_arg(float t) { return t; }
__device__ __forceinline__ float pal_arg(double t) { return (float)(t - ::floor(t)); }

__device__ __forceinline__ float aces(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;   /* shaders/mandelbrot.comp:38-45 */
    return clamp01((x * (a * x + b)) / (x * (c * x + d) + e));
}

/* enhance_color -> aces_tonemap -> gamma, shaders/mandelbrot.comp:48-54,233-235;
 * Julia's floors shaders/julia.comp:319-322 */
__device__ __forceinline__ void post_chain(float rgb[3], float brightness, float saturation,
                                           float contrast, bool julia)
{
    if (julia) {
        brightness = fmaxf(brightness, 0.1f);
        saturation = fmaxf(saturation, 0.0f);
        contrast = fmaxf(contrast, 0.1f);
    }
    float c[3];
    for (int k = 0; k < 3; ++k) c[k] = rgb[k] * brightness;
    for (int k = 0; k < 3; ++k) c[k] = (c[k] - 0.5f) * contrast + 0.5f;
    const float gray = c[0] * 0.299f + c[1] * 0.587f + c[2] * 0.114f;
    for (int k = 0; k < 3; ++k) c[k] = clamp01(gray * (1.0f - saturation) + c[k] * saturation);
    for (int k = 0; k < 3; ++k) rgb[k] = pow01(aces(c[k]), 1.0f / 2.2f);
}

/* Colour of a finished sample from its smooth count alone (no trap/stripe effects): the part of the
 * shaders after nu is known.  Mandelbrot: shaders/mandelbrot.comp:179-190; Julia: shaders/julia.comp:243-248;
 * Burning Ship: shaders/burning_ship.comp:259-299 (interior style 0 / no accumulators).
 * Shared by the render kernels and by colorize_kernel, so a frame recoloured from its nu plane is
 * bit-identical to the frame the render kernels write. */
/* wave-uniform colour-stage constants in the kernel's precision; ARGS is LaunchArgs by reference (general kernels) or
 * the kernel-argument segment itself (kargs(): re-read at the point of use) */
template <typename T> struct ArgsOf;
template <> struct ArgsOf<double> {
    template <class ARGS> static __device__ __forceinline__ double inv_max_iter(ARGS& a) { return a.inv_max_iter; }
    template <class ARGS> static __device__ __forceinline__ double inv_log2_bailout(ARGS& a) { return a.inv_log2_bailout; }
    template <class ARGS> static __device__ __forceinline__ double color_scale(ARGS& a) { return a.color_scale_d; }
    template <class ARGS> static __device__ __forceinline__ double color_offset(ARGS& a) { return a.color_offset_d; }
};
template <> struct ArgsOf<float> {
    template <class ARGS> static __device__ __forceinline__ float inv_max_iter(ARGS& a) { return a.inv_max_iter_f; }
    template <class ARGS> static __device__ __forceinline__ float inv_log2_bailout(ARGS& a) { return a.inv_log2_bailout_f; }
    template <class ARGS> static __device__ __forceinline__ float color_scale(ARGS& a) { return a.color_scale; }
    template <class ARGS> static __device__ __forceinline__ float color_offset(ARGS& a) { return a.color_offset; }
};

template <typename T, int FRACTAL, class ARGS>
__device__ __forceinline__ void colour_of(ARGS& A, const LdsBlock& S, const T nu, const bool interior,
                                          float rgb[3])
{
    const T inv_max_iter = ArgsOf<T>::inv_max_iter(A);
    const T color_scale = ArgsOf<T>::color_scale(A), color_offset = ArgsOf<T>::color_offset(A);
    if constexpr (FRACTAL == 0) {
        T t = nu * inv_max_iter * color_scale;                        /* :179 */
        t = t < T(0) ? T(0) : (t > T(1) ? T(1) : t);
        palette_eval(A.pal, S.pal, pal_arg(t + color_offset), rgb);
        if (A.interior_style == 1) {                                  /* :182-183, :190 (wave-uniform test) */
            rgb[0] = interior ? 0.0f : rgb[0]; rgb[1] = interior ? 0.0f : rgb[1]; rgb[2] = interior ? 0.0f : rgb[2];
        }
    } else {
        T t = nu * inv_max_iter;
        t = color_offset + t * color_scale;
        palette_eval(A.pal, S.pal, pal_arg(t), rgb);
        rgb[0] = interior ? 0.0f : rgb[0]; rgb[1] = interior ? 0.0f : rgb[1]; rgb[2] = interior ? 0.0f : rgb[2];   /* :243-244 interior: black */
    }
}

/* Smooth count + colour of one finished sample (no trap/stripe effects).  `it` is the escape
 * index, it >= max_iter for a sample that never escaped.
 * Mandelbrot: shaders/mandelbrot.comp:172-190; Julia: shaders/julia.comp:237-248. */
template <typename T, int FRACTAL, class ARGS>
__device__ __forceinline__ void shade(ARGS& A, const LdsBlock& S, const LogTab<T>& lg, const int it, const T r2,
                                      const bool want_nu, const bool want_rgb, T& nu, float rgb[3])
{
    const int max_iter = A.max_iter;                                  /* wave-uniform */
    nu = T(0);
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    if (!want_nu) return;
    nu = (T)max_iter;                                                 /* mandelbrot.comp:172, julia.comp:243 */
    if (it < max_iter) {
        const T i1 = (T)(it + 1);                                     /* == (T)it + 1: integers below 2^24 */
        if constexpr (FRACTAL == 0) {                                 /* :173-177: mu = log2(log2|z|) */
            if (!A.lib_log) {
                nu = i1 - lg.log2(T(0.5) * lg.log2(r2));
            } else {
                cold_path();
                const T log_zn = Real<T>::log(r2) / T(2);
                nu = (T)it + T(1) - Real<T>::log(log_zn / Real<T>::ln2()) / Real<T>::ln2();
            }
        } else {                                                      /* julia.comp:237-248 */
            /* log(log(r2)/log(B))/log 2 == log2(log2(r2) / log2(B)) */
            if (!A.lib_log) {
                nu = i1 - lg.log2(lg.log2(r2) * ArgsOf<T>::inv_log2_bailout(A));
            } else {
                cold_path();
                nu = (T)it + T(1) - Real<T>::log(Real<T>::log(r2) / (T)S.log_bailout) / Real<T>::ln2();
            }
        }
    }
    if (want_rgb) {
        /* Samples that never escaped all get one colour (nu = max_iter): computed once per workgroup (stage_interior).
         * A wave that retires only such samples -- C2's lane pool: 92 % of its records, mostly retiring together when a
         * deadline or a closed cycle takes a whole refill group -- skips the colour stage altogether. */
        const bool escaped = it < max_iter;
        if (__builtin_amdgcn_ballot_w64(escaped) != 0ull) colour_of<T, FRACTAL>(A, S, nu, false, rgb);
        if (!escaped) { rgb[0] = S.interior_rgb[0]; rgb[1] = S.interior_rgb[1]; rgb[2] = S.interior_rgb[2]; }
    }
}

/* the colour of interior samples, by the same colour_of() every sample used to run: thread 0, once per workgroup */
template <typename T, int FRACTAL>
__device__ __forceinline__ void stage_interior(LdsBlock& S, const LaunchArgs& A)
{
    if (threadIdx.x == 0) {
        float rgb[3] = {0.0f, 0.0f, 0.0f};
        colour_of<T, FRACTAL>(A, S, (T)A.max_iter, true, rgb);
        S.interior_rgb[0] = rgb[0]; S.interior_rgb[1] = rgb[1]; S.interior_rgb[2] = rgb[2];
    }
    __syncthreads();
}

/* ---- recolour from the smooth-count plane ---------------------------------------------------------
 * rgba[i] = the colour the render kernels write for a sample whose smooth count is nu[i].  Only valid
 * when the colour is a function of nu alone and "interior" can be read off nu (nu == max_iter): no
 * effects variant, no SSAA, bailout > 2 so that every escaped sample has nu < max_iter (|z| > 2 gives
 * log2(log2|z|) > 0; the host enforces this).  HBM-bound: 8 (4) B read + 16 B written per pixel.  Used by
 * the multi-GPU exchange, which ships the 8-byte nu plane over xGMI instead of the 16-byte colour. */
template <typename T, int FRACTAL>
__global__ void __launch_bounds__(kBlockThreads)
colorize_kernel(const LaunchArgs A, const T* __restrict__ nu_in, float4* __restrict__ rgba, const size_t n)
{
    __shared__ LdsBlock S;
    stage_constants(S, A);
    const T interior_nu = (T)S.max_iter;
    const size_t stride = (size_t)gridDim.x * kBlockThreads;
    for (size_t i = (size_t)blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += stride) {
        const T nu = nu_in[i];
        float rgb[3] = {0.0f, 0.0f, 0.0f};
        colour_of<T, FRACTAL>(A, S, nu, nu == interior_nu, rgb);
        if (A.flags & FR_FLAG_POST_CHAIN)
            post_chain(rgb, S.brightness, S.saturation, S.contrast, FRACTAL != 0);
        rgba[i] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
    }
}

/* ---- control block ---------------------------------------------------------------------------------------------
 * Queue heads and stream counters of a render are zeroed by this kernel, not by hipMemsetAsync: a memset node captured
 * into a HIP graph did its work on the first replay only (ROCm 7.2: the second replay found the heads where the first
 * had left them and rendered nothing), and fr_render_shard_async is meant to be capturable. */
/* What the lane pool of a render tells the host about itself -- "did any wave close a cycle?" -- travels with the NEXT
 * render's first launch: the pool's waves set a word in device memory, thread 0 of this launch forwards it (with the
 * sequence number of the render it belongs to) to a host-mapped word and clears it.  No synchronisation, no extra launch;
 * the host reads the word when it plans a later render (fr_device.hip, pool_wants_cycle_closing).  A hint only: nothing a
 * pixel depends on. */
constexpr int kFeedbackShards = 64;  /* the flag is 64 words, 128 B apart, a wave sets the one of its workgroup index: every wave
                                      * of a C2 pool pass closes cycles, and 6 144 stores to ONE word at wave exit -- same-address
                                      * stores are served ~15 ns apart, like the atomics -- held the end of the pass up by 80-120 us
                                      * (1080p / 1024: 83 -> 164 us; profiles/r03_periodicity_cost.txt) */
struct Feedback {
    uint32_t* dev_flag;          /* kFeedbackShards pairs of words {samples retired by a closed cycle, records taken in},
                                  * kShardStrideWords apart, added to by the waves of pool_kernel<.., PERIOD = true> when they
                                  * leave; one more word behind them: "the render's lane pool LOOKED for cycles" (set by one
                                  * wave of that kernel) -- only such a render has a verdict to forward */
    uint32_t* host_word;         /* host-mapped: (sequence number of the render << 1) | "closing cycles paid" (an eighth of the
                                  * pool's records and more were retired by a closed cycle), of the most recent render whose
                                  * lane pool looked */
    uint32_t prev_seq;           /* sequence number of the context's previous render */
};
__device__ __forceinline__ void forward_feedback(const Feedback& fb)
{
    if (fb.dev_flag && blockIdx.x == 0 && threadIdx.x < (uint32_t)kFeedbackShards) {          /* wave 0 of workgroup 0 */
        uint32_t* w = fb.dev_flag + threadIdx.x * kShardStrideWords;
        uint32_t closed = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t records = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(w + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            closed += (uint32_t)__shfl_xor((int)closed, off, 64);
            records += (uint32_t)__shfl_xor((int)records, off, 64);
        }
        /* looking costs a frame in which nothing closes 4-7 %; a closed cycle saves most of its record's updates */
        const bool any = closed != 0u && (uint64_t)closed * 8u >= (uint64_t)records;
        if (threadIdx.x == 0) {
            uint32_t* lw = fb.dev_flag + kFeedbackShards * kShardStrideWords;
            const uint32_t looked = __hip_atomic_load(lw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (looked != 0u) {
                __hip_atomic_store(lw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(fb.host_word, (fb.prev_seq << 1) | (any ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void __launch_bounds__(kBlockThreads)
clear_words_kernel(uint32_t* __restrict__ p, const uint32_t n, const Feedback fb)
{
    forward_feedback(fb);
    for (uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += gridDim.x * kBlockThreads) p[i] = 0u;
}

/* ---- diagnostic build (-DFR_STAMP): where a wave's time goes ---------------------------------------------------
 * Per wave, shader-clock cycles spent inside the marked regions, written next to the timeline words of the diag buffer
 * (8 words per wave in this build: t0, t1, items, claims, then cycles in: dequeue, block claim, refill wait, retire/shade).
 * Never compiled into the product library. */
#ifdef FR_STAMP
#define FR_STAMP_DECL uint64_t st_acc[4] = {0, 0, 0, 0}; uint64_t st_t = 0; (void)st_t;
#define FR_STAMP_BEGIN() do { st_t = __builtin_amdgcn_s_memtime(); } while (0)
#ifdef FR_STAMP_TESTED   /* -DFR_STAMP -DFR_STAMP_TESTED: the slots count updates instead of cycles -- 0: updates a lane-pool wave ran
                          * TESTED, 2: updates of its dirty unchecked stretches (FR_CLOCK_GHZ=0.001 tools/stamps.py prints counts) */
#define FR_STAMP_END(k) do {} while (0)
#else
#define FR_STAMP_END(k) do { st_acc[k] += __builtin_amdgcn_s_memtime() - st_t; } while (0)
#endif
#define FR_STAMP_WRITE(A, lane) do { if ((A).diag && (lane) == 0) { uint64_t* d_ = (A).diag + (size_t)(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * 8 + 4; \
        d_[0] = st_acc[0]; d_[1] = st_acc[1]; d_[2] = st_acc[2]; d_[3] = st_acc[3]; } } while (0)
constexpr int kDiagWords = 8;
#else
#define FR_STAMP_DECL
#define FR_STAMP_BEGIN() do {} while (0)
#define FR_STAMP_END(k) do {} while (0)
#define FR_STAMP_WRITE(A, lane) do {} while (0)
constexpr int kDiagWords = 4;
#endif

/* ---- XCD id ------------------------------------------------------------------------------ */
__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;      /* speed hint only: picks the home shard, never correctness */
}

/* ---- XCD-sharded work queue ------------------------------------------------------------------
 * 8 heads, one per XCD.  A wave claims a run of items from its home shard with one atomicAdd,
 * run length clamp(remaining >> run_shift, run_min, run_max); when a shard is dry it moves to
 * the next one and returns false after `max_tries` shards (all of them by default) were found dry -- every
 * wave reaches that.  (Claiming the next run ahead of need was measured equal or slower on every
 * workload -- it commits the wave to one more run -- and is gone: the state it carried cost SGPRs.) */
struct WaveQueue {
    uint32_t* heads;
    uint32_t n_groups, group;     /* groups dealt round-robin: shard k owns groups k, k+8, ... of `group` items */
    const uint32_t* len_words;    /* or: per-shard lengths in memory (kShardStrideWords apart), capped at len_cap */
    uint32_t len_cap;
    uint32_t run_shift, run_min, run_max, run_even;
    uint32_t ns, ns_log2;         /* shards (8 or 64) */
    uint32_t lane;
    uint32_t shard, tried, seen, max_tries;
    uint32_t cur_n, cur_raw;

    __device__ __forceinline__ uint32_t shard_len(uint32_t sh) const
    {
        if (len_words) {
            const uint32_t v = len_words[sh * kShardStrideWords];      /* uniform address: scalar load */
            return v < len_cap ? v : len_cap;
        }
        (void)sh;                  /* every shard owns one block per round of ns blocks (see block_of); the last round may be short */
        return ((n_groups + ns - 1u) >> ns_log2) * group;
    }
    __device__ __forceinline__ uint32_t run_len(uint32_t sh, uint32_t seen_head) const
    {
        const uint32_t l = shard_len(sh);
        const uint32_t rem = seen_head < l ? l - seen_head : 0u;
        uint32_t n = rem >> run_shift;
        n = n < run_min ? run_min : n;
        n = n > run_max ? run_max : n;
        return (n + run_even) & ~run_even;          /* run_even = 1: whole pairs of sub-tiles (lean tile pass, NP = 2) */
    }
    __device__ __forceinline__ uint32_t claim(uint32_t sh, uint32_t n) const
    {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&heads[sh * kShardStrideWords], n);
        return v;
    }
    /* Shard-local block index jb of shard sh -> block of the frame.  Round jb of the deal covers blocks [jb ns, (jb+1) ns);
     * the shard's place in it ROTATES with the round: with a fixed place shard s of 64 would own the same 128-pixel column
     * stripe of every other sub-tile row (32 blocks per row at 4096 pixels) -- a stripe through the set costs several times
     * one at the frame's edge, and waves that stop at their home shard cannot level that out (measured: C3 tile pass
     * 134 -> 157 us with 64 fixed-place shards).  Rotating, every shard visits every column. */
    static __device__ __forceinline__ uint32_t block_of(uint32_t jb, uint32_t sh, uint32_t nslog2)
    {
        return (jb << nslog2) + ((sh + jb) & ((1u << nslog2) - 1u));
    }
    /* home shard: the wave's XCD, and -- with 64 shards -- one of that XCD's eight by workgroup */
    static __device__ __forceinline__ uint32_t home_of(uint32_t nshards)
    {
        return (xcc_id() | ((blockIdx.x >> 3) << 3)) & (nshards - 1u);
    }
    __device__ __forceinline__ void init(uint32_t* h, uint32_t groups, uint32_t group_items, uint32_t shift,
                                         uint32_t rmin, uint32_t rmax, uint32_t ln, uint32_t nslog2)
    {
        heads = h; ns_log2 = nslog2; ns = 1u << nslog2;
        n_groups = groups; group = group_items; len_words = nullptr; len_cap = 0;
        run_shift = shift; run_min = rmin; run_max = rmax; run_even = 0u; lane = ln;
        shard = home_of(ns); tried = 0; seen = 0; max_tries = ns;
        cur_n = cur_raw = 0;
    }
    __device__ __forceinline__ void init_lengths(uint32_t* h, const uint32_t* lengths, uint32_t cap, uint32_t shift,
                                                 uint32_t rmin, uint32_t rmax, uint32_t ln, uint32_t nslog2)
    {
        init(h, 0, 1, shift, rmin, rmax, ln, nslog2);
        len_words = lengths; len_cap = cap;
    }
    /* Probing every other shard before exiting costs up to 8 dependent atomics per wave, and all waves of
     * a launch do it at the same moment (waves x 8 serialized RMWs on 8 words); a launch whose items are
     * dealt evenly and cost about the same can stop after its home shard (+ a neighbour). */
    __device__ __forceinline__ void set_probes(uint32_t flags)
    {
        const uint32_t p = (flags >> kQueueProbeShift) & 0xFu;
        if (p == 0u || p >= ns) return;
        max_tries = p;
        /* with limited probing every shard must be SOMEBODY's home whatever the hardware's workgroup ->
         * XCD placement is: take the home from the workgroup index (identical to the XCD id under the
         * usual round-robin dispatch).  The host only limits probing on grids of >= 64 workgroups. */
        shard = blockIdx.x & (ns - 1u);
    }
    /* the next shard to try: with 64 shards the other seven of the same XCD first (s + 8, s + 16, ...) */
    __device__ __forceinline__ void advance()
    {
        if (ns > (uint32_t)kShards) {
            shard = (shard + (uint32_t)kShards) & (ns - 1u);
            if ((tried & 7u) == 0u) shard = (shard + 1u) & (ns - 1u);
        } else {
            shard = (shard + 1u) & (ns - 1u);
        }
    }
    /* hands out the next run [begin, begin+count) of shard `sh`; false = no work left anywhere */
    __device__ __forceinline__ bool next(uint32_t& begin, uint32_t& count, uint32_t& sh)
    {
        const bool got = next_(begin, count, sh);
        /* Every claim above has been waited for and read -- but not as far as the compiler's s_waitcnt pass can tell on
         * every path: it carried "a returning atomic may still be writing this VGPR" into the callers' loops and put an
         * s_waitcnt vmcnt(0) in front of the first instruction that reuses the register -- EVERY sub-tile of the tile pass
         * then waited for the previous sub-tile's stores to be acknowledged (found in the ISA; SQ_WAIT_ANY was 47 % of
         * the tile pass's wave-cycles).  One wait the pass can see, here, where nothing is in flight anyway, clears
         * its scoreboard.  (vmcnt(0), expcnt / lgkmcnt untouched: simm16 0x0F70 on gfx9.) */
        __builtin_amdgcn_s_waitcnt(0x0F70);
        return got;
    }
    __device__ __forceinline__ bool next_(uint32_t& begin, uint32_t& count, uint32_t& sh)
    {
        /* shards known to be empty cost no atomic (a follow-up pass may have nothing to do) */
        while (shard_len(shard) == 0u) {
            if (++tried >= max_tries) return false;
            advance();
            seen = 0;
        }
        cur_n = run_len(shard, seen); cur_raw = claim(shard, cur_n);
        for (;;) {
            const uint32_t b = __builtin_amdgcn_readfirstlane(cur_raw);
            const uint32_t l = shard_len(shard);
            if (b >= l) {
                do {
                    if (++tried >= max_tries) return false;
                    advance();
                } while (shard_len(shard) == 0u);
                seen = 0;
                cur_n = run_len(shard, seen);
                cur_raw = claim(shard, cur_n);
                continue;
            }
            uint32_t c = cur_n;
            if (b + c > l) c = l - b;
            seen = b + c;
            begin = b; count = c; sh = shard;
            return true;
        }
    }
};

/* ---- survivor ring ---------------------------------------------------------------------------
 * A wave's unfinished samples are appended to a ring in LDS (wave-private: DS operations of one
 * wave execute in order, no barrier needed); whenever 64 are queued, lane l takes record head+l
 * and the wave writes one dense, fully coalesced block to the output stream.  Block indices come
 * from one atomicAdd per 64 records, claimed one block ahead so the latency never shows. */
template <typename T, int NF>
struct WaveRing {
    uint32_t pix[kRingSlots];
    uint32_t it[kRingSlots];
    T f[NF][kRingSlots];
};

template <typename T, int NF>
struct RingWriter {
    WaveRing<T, NF>* ring;
    StreamRef out;
    uint32_t lane;
    uint32_t head, tail;          /* wave-uniform record counters */
    uint32_t home;                /* region this wave appends to (its XCD) */
#ifdef FR_STAMP
    uint64_t st_block = 0;        /* cycles inside take_block */
#endif

    static constexpr size_t kHeaderBytes = 2 * 64 * 4;           /* pixel + iterations-done planes */
    static constexpr size_t kBlockBytes = kHeaderBytes + (size_t)NF * 64 * sizeof(T);

    __device__ __forceinline__ void init(WaveRing<T, NF>* r, const StreamRef& o, uint32_t ln)
    {
        ring = r; out = o; lane = ln; head = tail = 0; home = WaveQueue::home_of(o.nregions);
    }
    /* one block of the home region (one atomicAdd on the region's counter: 8 counters share the
     * load); a full region spills to the next one -- the regions together hold 1.5x the worst case */
    __device__ __forceinline__ bool take_block(uint32_t& region, uint32_t& blk)
    {
#ifdef FR_STAMP
        const uint64_t st0 = __builtin_amdgcn_s_memtime();
        struct Acc { uint64_t& a; uint64_t t0; __device__ ~Acc() { a += __builtin_amdgcn_s_memtime() - t0; } } acc_{st_block, st0};
#endif
        for (uint32_t t = 0; t < out.nregions; ++t) {
            region = (home + t) & (out.nregions - 1u);
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(&out.n_blocks[region * kShardStrideWords], 1u);
            blk = __builtin_amdgcn_readfirstlane(v);
            if (blk < out.region_blocks) {
                if (out.rotate) home = (region + 1u) & (out.nregions - 1u);
                return true;
            }
        }
        /* cannot happen (capacity is worst-case x 1.5); if it does, say so: the host fails the render */
        if (lane == 0 && out.overflow) __hip_atomic_store(out.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
    }
    __device__ __forceinline__ void write_block(uint32_t nvalid)
    {
        uint32_t region, blk;
        if (!take_block(region, blk)) { head += nvalid; return; }
        const uint32_t slot = (head + lane) & (kRingSlots - 1);
        const bool valid = lane < nvalid;
        uint8_t* b = out.base + ((size_t)region * out.region_blocks + blk) * kBlockBytes;
        reinterpret_cast<uint32_t*>(b)[lane] = valid ? ring->pix[slot] : kInvalidPixel;
        reinterpret_cast<uint32_t*>(b)[64 + lane] = valid ? ring->it[slot] : 0u;
        T* fields = reinterpret_cast<T*>(b + kHeaderBytes);
#pragma unroll
        for (int k = 0; k < NF; ++k) fields[k * 64 + lane] = valid ? ring->f[k][slot] : T(0);
        head += nvalid;
    }
    /* append the lanes with `keep` set; `done` = iterations the sample has run, v[] the record fields */
    __device__ __forceinline__ void append(bool keep, uint32_t pixel, uint32_t done, const T (&v)[NF])
    {
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        if (m == 0ull) return;
        const uint32_t n = (uint32_t)__builtin_popcountll(m);
        if (keep) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t slot = (tail + rank) & (kRingSlots - 1);
            ring->pix[slot] = pixel;
            ring->it[slot] = done;
#pragma unroll
            for (int k = 0; k < NF; ++k) ring->f[k][slot] = v[k];
        }
        tail += n;
        __builtin_amdgcn_wave_barrier();
        if (tail - head >= 64u) write_block(64u);
        __builtin_amdgcn_wave_barrier();
    }
    /* end of kernel: flush the partial block (the only not-full block a wave ever writes) */
    __device__ __forceinline__ void finish()
    {
        const uint32_t left = tail - head;
        if (left > 0u) write_block(left);
    }
};

/* ---- one escape-time run over a wave's 64 samples -------------------------------------------
 *
 * Scaled-imaginary form.  State per lane: X = Re z, Yd = 2*Im z, cx, cyd = 2*Im c,
 * x2 = X*X, y2d = Yd*Yd (= 4*(Im z)^2).  One update:
 *     p   = X * Yd                 == round(2*zx*zy)            (as-written  2.0*z.x*z.y)
 *     t   = fma(-0.25, y2d, x2)    == round(zx*zx - zy*zy)      (y2d/4 is exact)
 *     X'  = t + cx                 == round(t + cx)
 *     Yd' = fma(2, p, cyd)         == 2*round(p + cy)           (scaling by 2 commutes with rounding)
 *     x2' = X'*X' ; y2d' = Yd'*Yd'
 *     r2  = fma(0.25, y2d', x2')   == round(zx'^2 + zy'^2)      (dot(z,z))
 * i.e. 7 VALU ops instead of the as-written 9 (2*zx, zx*zy, zx*zx, zy*zy, -, +cx, +cy, and
 * the dot's add), each value bit-identical to the as-written one barring denormal products
 * (|Im z| < 1e-154), which cannot influence an escape.
 *
 * Unchecked blocks.  When escape is absorbing (fast_ok: bailout^2 >= 4.5 and |c| <= bailout
 * for every live lane, so |z| > bailout implies |z'| >= |z|(|z|-1) > 1.12|z|) the wave runs
 * kFastBlock updates without computing r2 or testing, then tests once: a lane that escaped
 * inside the block is still escaped (or inf/NaN) at its end.  If any did, the block is
 * rolled back to its snapshot and replayed with per-iteration tests, which reproduces the
 * exact escape index and |z|^2.  The wave stays in tested mode until a whole block passes
 * with no escape.
 */
template <typename T>
struct Orbit {
    T X, Yd, cx, cyd, x2, y2d;
};

/* ABS (Burning Ship, shaders/burning_ship.comp:241-245): z = abs(z) before the square only changes
 * the sign of the cross term, 2|zx||zy| = |X|*|Yd|; the abs are VOP3 input modifiers, no extra ops. */
template <typename T, bool ABS = false>
__device__ __forceinline__ void orbit_step(Orbit<T>& o)
{
    const T p = ABS ? Real<T>::fabs(o.X) * Real<T>::fabs(o.Yd) : o.X * o.Yd;
    const T t = Real<T>::fma(T(-0.25), o.y2d, o.x2);
    o.X = t + o.cx;
    o.Yd = Real<T>::fma(T(2), p, o.cyd);
    o.x2 = o.X * o.X;
    o.y2d = o.Yd * o.Yd;
}

/* fp32: gfx950 has packed two-wide fp32 multiplies and fmas (v_pk_mul_f32, v_pk_fma_f32).  Each half is an
 * ordinary IEEE operation with its own rounding, so pairing operations of ONE pixel changes no bit:
 * {X', Yd'} = {fma(1, t, cx), fma(2, p, cyd)}  (fma(1, t, cx) is RN(t + cx)) and {x2', y2d'} = {X', Yd'}^2:
 * four VALU instructions per update instead of six.  The packed forms issue at half the rate of the scalar-width
 * ones (profiles/r01_ubench_valu_rates.txt: 74.7 T values/s either way), so this is not a 1.5x: what it buys is
 * issue slots and a shorter dependent chain -- measured +5.8 % on a 3840x2160 fp32 frame, +1.3 % on C3
 * (profiles/r01_packed_fp32_step.txt). */
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool ABS>
__device__ __forceinline__ void orbit_step_f32(Orbit<float>& o)
{
    /* p through an asm statement: left to itself the vectoriser pairs this product with the squares
     * (a second v_pk_mul_f32 with half its result unused) and then needs a v_mov to put p next to t */
    const float t = __builtin_fmaf(-0.25f, o.y2d, o.x2);
    float p;            /* t as a (textually unused) input orders the product after the fmac, so it can take y2d's register */
    if (ABS) asm("v_mul_f32_e64 %0, |%1|, |%2|" : "=v"(p) : "v"(o.X), "v"(o.Yd), "v"(t));
    else asm("v_mul_f32_e32 %0, %1, %2" : "=v"(p) : "v"(o.X), "v"(o.Yd), "v"(t));
    const f32x2 tp = {t, p}, c = {o.cx, o.cyd}, k = {1.0f, 2.0f};
    const f32x2 z = __builtin_elementwise_fma(k, tp, c);
    const f32x2 sq = z * z;
    o.X = z.x; o.Yd = z.y;
    o.x2 = sq.x; o.y2d = sq.y;
}
template <> __device__ __forceinline__ void orbit_step<float, false>(Orbit<float>& o) { orbit_step_f32<false>(o); }
template <> __device__ __forceinline__ void orbit_step<float, true>(Orbit<float>& o) { orbit_step_f32<true>(o); }

/* 4 |z|^2 = RN(4 x2 + y2d): the escape test compares it with 4 B^2.  Scaling by a power of two commutes
 * with rounding, so this is exactly 4 * RN(zx^2 + zy^2) and the test is the as-written one; written this
 * way the multiplier 4.0 is an inline constant of a three-operand v_fma_f64 (0.25 is not: the compiler
 * needed a register copy + v_fmac with a literal, one more instruction per tested update). */
template <typename T>
__device__ __forceinline__ T orbit_r2x4(const Orbit<T>& o)
{
    return Real<T>::fma(T(4), o.x2, o.y2d);
}

/* Runs the wave's 64 orbits over iterations [i0, i1).  esc_i: escape index, i1 if the lane is
 * still alive after update i1-1 (its orbit state is then the state after i1 updates);
 * esc_r2: |z|^2 at escape.  done_in: lanes that must not run (outside the frame / empty). */
/* PERIOD (with period_window != 0, and only when i1 is the sample's last iteration): cycle closing as in the lane
 * pool (pool_kernel) -- a lane whose state returns to its own state at the last snapshot can never escape and is
 * finished as "alive at i1" at once; a wave of interior samples then leaves the loop early. */
template <typename T, bool ABS = false, bool PERIOD = false>
__device__ __forceinline__ void escape_run(Orbit<T>& o, const T B2, const int i0, const int i1,
                                           const bool fast_ok, const bool start_fast, const uint64_t done_in,
                                           int& esc_i, T& esc_r2, const uint32_t period_window = 0u)
{
    esc_i = i1;
    esc_r2 = T(0);
    const T B2x4 = T(4) * B2;
    uint64_t done = done_in;
    int i = i0;                      /* wave-uniform: SGPR */
    bool fast = start_fast && fast_ok;
    T refX = __builtin_nan(""), refYd = __builtin_nan("");
    uint32_t snap_window = period_window, snap_closed = 1u;
    const uint32_t snap_cap = period_window > (((uint32_t)i1 >> 7) << 4) ? period_window : (((uint32_t)i1 >> 7) << 4);
    int next_snap = i0;
    /* where every running lane is known not to have escaped */
    auto close_cycles = [&]() {
        const bool hit = o.X == refX && o.Yd == refYd;       /* finished lanes sit at 0 == 0: masked by `done` */
        const uint64_t hm = __builtin_amdgcn_ballot_w64(hit) & ~done;
        if (hm != 0ull) {
            if (hit) { o.X = T(0); o.Yd = T(0); o.cx = T(0); o.cyd = T(0); o.x2 = T(0); o.y2d = T(0); }
            done |= hm;
            snap_closed += (uint32_t)__builtin_popcountll(hm);
        }
        if (i >= next_snap) {
            refX = o.X; refYd = o.Yd;
            if (snap_closed == 0u && snap_window < snap_cap) snap_window <<= 1;
            snap_closed = 0u;
            next_snap = i + (int)snap_window;
        }
    };

    while (i < i1) {
        if (done == ~0ull) break;    /* every lane finished: wave-uniform early-out */
        const int left = i1 - i;
        if (fast && left >= kFastBlock) {
            const Orbit<T> snap = o;
#pragma unroll
            for (int k = 0; k < kFastBlock; ++k) orbit_step<T, ABS>(o);
            const bool bad = !(orbit_r2x4(o) <= B2x4);
            if (__builtin_amdgcn_ballot_w64(bad) == 0ull) {
                i += kFastBlock;
                if constexpr (PERIOD) { if (period_window) close_cycles(); }
                continue;
            }
            o = snap;                /* roll back, replay tested */
            fast = false;
        }
        /* tested block: the loop carries one counter and one vector-compare branch (scalar instructions
         * share ONE issue port per CU: per-iteration scalar bookkeeping is what bounds escape-dense views) */
        int end = i + (left < kFastBlock ? left : kFastBlock);
        const uint64_t before = done;
        do {
            orbit_step<T, ABS>(o);
            const T r2x4 = orbit_r2x4(o);
            const bool e = r2x4 > B2x4;
            const uint64_t em = __builtin_amdgcn_ballot_w64(e);
            if (em != 0ull) {
                if (e) {
                    esc_i = i;
                    esc_r2 = T(0.25) * r2x4;
                    /* park at the fixed point z = 0 of c = 0: never "escapes" again */
                    o.X = T(0); o.Yd = T(0); o.cx = T(0); o.cyd = T(0); o.x2 = T(0); o.y2d = T(0);
                }
                done |= em;
                if (done == ~0ull) end = i;      /* single-exit loop: everybody finished -> this was the last update */
            }
            ++i;
        } while (i < end);
        if (done == ~0ull) return;
        fast = fast_ok && done == before;
        if constexpr (PERIOD) { if (period_window) close_cycles(); }
    }
}

/* The reference's loop with its 3-way orbit trap (shaders/mandelbrot.comp:157-170), for the colourings that need
 * minTrap or the z of the escape (orbit trap, stripes, interior style 2).
 *
 * Every iteration the shader takes min(length(z), min(|z.x|, |z.y|), length(z - c)) into minTrap.  A correctly
 * rounded square root is monotonic, so min_i sqrt(a_i) == sqrt(min_i a_i) bit for bit: the loop keeps the minima of
 * the SQUARED lengths and takes the two roots once, after it.  It runs in the scaled-imaginary form of escape_run
 * (Yd = 2 Im z, exact power-of-two scalings; see there), so the minima are kept scaled too and scaled back at the end:
 *     4 |z|^2     = fma(4, x2, y2d)                         (= 4 RN(zx^2 + zy^2), the dot() of length())
 *     |z.y|       = |Yd| / 2
 *     4 |z - c|^2 = fma(4, RN(dx^2), RN(dyd^2)),  dyd = Yd - cyd = 2 (zy - cy)   (= 4 RN(dx^2 + dy^2))
 * and sqrt(m / 4) == sqrt(m) / 2 exactly.  Unchecked blocks as in escape_run: a block in which some running lane
 * escaped is rolled back -- orbit AND minima -- and replayed with per-update tests, which stops every lane's minima
 * at its escape.  A finished lane is parked at NaN: every later minimum with its values is a no-op (minNum ignores
 * NaN) and it never tests as escaped; the dirty-block test looks at running lanes only. */
template <typename T>
__device__ __forceinline__ void escape_run_effects(Orbit<T>& o, const T B2, const int max_iter, const bool fast_ok,
                                                   const bool lane_runs, const bool need_trap,
                                                   int& esc_i, T& esc_zx, T& esc_zy, T& min_trap)
{
    esc_i = max_iter;
    esc_zx = T(0); esc_zy = T(0);
    const T B2x4 = T(4) * B2;
    const T kInf = (T)__builtin_inff(), kNaN = (T)__builtin_nanf("");
    T m_o4 = kInf, m_ax = T(1e20), m_ayd = T(2e20), m_c4 = kInf;
    uint32_t fin = lane_runs ? 0u : 1u;                          /* VGPR flag, as in the lane pool */
    if (!lane_runs) { o.X = kNaN; o.Yd = kNaN; o.x2 = kNaN; o.y2d = kNaN; }
    uint64_t done = __builtin_amdgcn_ballot_w64(fin != 0u);
    int i = 0;
    bool fast = false;
    auto accumulate = [&]() {
        m_o4 = Real<T>::fmin(m_o4, orbit_r2x4(o));
        m_ax = Real<T>::fmin(m_ax, Real<T>::fabs(o.X));
        m_ayd = Real<T>::fmin(m_ayd, Real<T>::fabs(o.Yd));
        const T dx = o.X - o.cx, dyd = o.Yd - o.cyd;
        m_c4 = Real<T>::fmin(m_c4, Real<T>::fma(T(4), dx * dx, dyd * dyd));
    };
    while (i < max_iter) {
        if (done == ~0ull) break;
        const int left = max_iter - i;
        if (fast && left >= kFastBlock) {
            const Orbit<T> snap = o;
            const T s_o4 = m_o4, s_ax = m_ax, s_ayd = m_ayd, s_c4 = m_c4;
            /* not fully unrolled: with its four running minima a 16-update straight line costs the kernel 157 VGPRs
             * (3 waves per SIMD); two updates per trip keep it at the tile kernel's usual budget */
            if (need_trap) {
#pragma unroll 2
                for (int k = 0; k < kFastBlock; ++k) { orbit_step<T, false>(o); accumulate(); }
            } else {
#pragma unroll
                for (int k = 0; k < kFastBlock; ++k) orbit_step<T, false>(o);
            }
            const bool bad = fin == 0u && !(orbit_r2x4(o) <= B2x4);
            if (__builtin_amdgcn_ballot_w64(bad) == 0ull) { i += kFastBlock; continue; }
            o = snap; m_o4 = s_o4; m_ax = s_ax; m_ayd = s_ayd; m_c4 = s_c4;
            fast = false;
        }
        const int end = i + (left < kFastBlock ? left : kFastBlock);
        const uint64_t before = done;
        for (; i < end; ++i) {
            orbit_step<T, false>(o);
            if (need_trap) accumulate();                         /* the escaping z is part of the minima, as written */
            const bool e = orbit_r2x4(o) > B2x4;
            const uint64_t em = __builtin_amdgcn_ballot_w64(e);
            if (em != 0ull) {
                if (e) {
                    esc_i = i; esc_zx = o.X; esc_zy = T(0.5) * o.Yd; fin = 1u;
                    o.X = kNaN; o.Yd = kNaN; o.x2 = kNaN; o.y2d = kNaN;
                }
                done |= em;
                if (done == ~0ull) { ++i; break; }
            }
        }
        if (done == ~0ull) break;
        fast = fast_ok && done == before;
    }
    if (fin == 0u) { esc_zx = o.X; esc_zy = T(0.5) * o.Yd; }      /* never escaped: the z after max_iter updates */
    const T m_axes = Real<T>::fmin(m_ax, T(0.5) * m_ayd);
    min_trap = Real<T>::fmin(T(1e20), Real<T>::fmin(T(0.5) * Real<T>::sqrt(m_o4), Real<T>::fmin(m_axes, T(0.5) * Real<T>::sqrt(m_c4))));
}

/* As-written Burning Ship loop with its trap / stripe accumulators (shaders/burning_ship.comp:228-256):
 * used when the colouring needs min_orbit_dist, stripe_value or the final z of an interior sample. */
template <typename T>
__device__ __forceinline__ void escape_run_ship_effects(T& zx, T& zy, const T cx, const T cy, const T B2,
                                                        const int max_iter, const uint64_t done_in,
                                                        const bool trap, const T trap_radius,
                                                        const bool stripes, const T stripe_density,
                                                        int& esc_i, T& esc_r2, T& min_dist, T& stripe_sum)
{
    esc_i = max_iter;
    esc_r2 = T(0);
    min_dist = T(1e10);
    stripe_sum = T(0);
    bool live = true;
    uint64_t done = done_in;
    for (int i = 0; i < max_iter; ++i) {
        if (done == ~0ull) break;
        if (live) {
            if (trap) {
                const T dist = Real<T>::sqrt(zx * zx + zy * zy);
                min_dist = Real<T>::fmin(min_dist, Real<T>::fabs(dist - trap_radius));
            }
            if (stripes) stripe_sum += Real<T>::sin(zy * stripe_density);
            const T ax = Real<T>::fabs(zx), ay = Real<T>::fabs(zy);
            const T x = ax * ax - ay * ay + cx;
            zy = T(2) * ax * ay + cy;
            zx = x;
            const T len_sq = zx * zx + zy * zy;
            if (len_sq > B2) { esc_i = i; esc_r2 = len_sq; live = false; }
        }
        done |= __builtin_amdgcn_ballot_w64(!live);
    }
}

/* FRACTAL: 0 Mandelbrot, 1 Julia, 2 Burning Ship.  Julia's c is a launch constant, so its survivor
 * records carry only z; the other two carry c per sample. */
template <int FRACTAL> struct RecFields { static constexpr int n = FRACTAL == 1 ? 2 : 4; };
template <int FRACTAL> struct Form {
    static constexpr bool abs_step = FRACTAL == 2;
    static constexpr bool per_sample_c = FRACTAL != 1;
};

__device__ __forceinline__ void diag_write(const LaunchArgs& A, uint32_t lane, uint64_t t0, uint32_t items, uint32_t claims)
{
    if (A.diag && lane == 0) {      /* diagnostics: per-wave timeline (100 MHz ticks) and work counts */
        const uint32_t wave_id = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
        uint64_t* d = A.diag + (size_t)wave_id * kDiagWords;
        d[0] = t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = items; d[3] = claims;
    }
}

/* ---- tile pass ---------------------------------------------------------------------------------
 * FRACTAL: 0 Mandelbrot, 1 Julia, 2 Burning Ship.  FPW_LOG2: log2 of the sub-tile width (3: 8x8, 4: 16x4, 6: 64x1).
 * EFFECTS: trap / stripe / interior-style variant (Mandelbrot and Burning Ship, never staged).
 * Runs iterations [0, A.i1); when A.i1 < max_iter the samples still alive go to A.out. */
template <typename T, int FRACTAL, int FPW_LOG2, bool EFFECTS, bool SSAA, bool PERIOD = false>
__global__ void __launch_bounds__(kBlockThreads)
tile_kernel(const LaunchArgs A)
{
    /* cycle closing in escape_run: always compiled into the SSAA variants, and into the PERIOD variant of the
     * one-sample kernel (an unstaged pass with the "periodicity" option on); the default one-sample kernel stays lean */
    constexpr bool CLOSE = SSAA || PERIOD;
    constexpr int FPW = 1 << FPW_LOG2;
    constexpr int FPH = kWave / FPW;
    constexpr int NF = RecFields<FRACTAL>::n;

    __shared__ LdsBlock S;
    __shared__ WaveRing<T, NF> rings[EFFECTS ? 1 : kWavesPerBlock];
    stage_constants(S, A);
    stage_interior<T, FRACTAL>(S, A);
    __shared__ double2 log2_lds[sizeof(T) == 8 ? kLog2Entries : 1];
    const LogTab<T> lg = stage_log2<T>(log2_lds, A);

    const int lane = threadIdx.x & (kWave - 1);
    const int lx = lane & (FPW - 1);
    const int ly = lane >> FPW_LOG2;

    /* viewport constants out of LDS (broadcast reads), narrowed as the reference narrows
     * them for its fp32 shaders (src/compute_effect_manager.h:85-90) */
    /* integers straight from the kernel arguments (wave-uniform SGPRs); through LDS they would be VGPR values
     * and everything derived from them would count as divergent */
    const int W = A.W, H = A.H, max_iter = A.max_iter;
    /* SSAA = false: the one-sample variant (the sample loop and its state fold away: fewer live SGPRs) */
    const int aa = SSAA ? (A.aa > 1 ? A.aa : 1) : 1;
    const T center_x = (T)S.center_x, center_y = (T)S.center_y, zoom = (T)S.zoom;
    const T bailout = (T)S.bailout;
    const T B2 = bailout * bailout;
    const T resx = (T)W, resy = (T)H;
    const T inv_w = sizeof(T) == 8 ? (T)A.inv_w_d : (T)A.inv_w_f;
    const T inv_h = sizeof(T) == 8 ? (T)A.inv_h_d : (T)A.inv_h_f;
    const T aspect = sizeof(T) == 8 ? (T)A.aspect_d : (T)A.aspect_f;      /* shaders/julia.comp:221 */
    /* planes nobody asked for are not computed (wave-uniform branches) */
    const bool want_rgb = A.rgba != nullptr;
    const bool want_nu = want_rgb || A.nu != nullptr;
    const bool staged = !EFFECTS && !SSAA && A.i1 < max_iter;        /* survivors continue in the stream pass */
    const int i1 = EFFECTS ? max_iter : A.i1;
    (void)inv_w; (void)aspect; (void)H;

    RingWriter<T, NF> writer;
    writer.init(&rings[EFFECTS ? 0 : (threadIdx.x >> 6)], A.out, (uint32_t)lane);

    WaveQueue q;
    q.init(A.q.heads, A.q.n_blk, (uint32_t)kShardBlock, A.q.run_shift, A.q.run_min, A.q.run_max, (uint32_t)lane, A.q.ns_log2);
    q.set_probes(A.q.flags);

    uint64_t diag_t0 = 0;
    uint32_t diag_items = 0, diag_claims = 0;
    if (A.diag) diag_t0 = __builtin_amdgcn_s_memrealtime();

    uint32_t begin, count, cur_shard;
    FR_STAMP_DECL
    for (;;) {
        FR_STAMP_BEGIN();
        const bool got_run = q.next(begin, count, cur_shard);
        FR_STAMP_END(0);
        if (!got_run) break;
        ++diag_claims;
        diag_items += count;
        for (uint32_t j = begin; j < begin + count; ++j) {
            /* shard-local index -> sub-tile id: blocks of kShardBlock sub-tiles dealt round-robin to the shards */
            const uint32_t blk = WaveQueue::block_of(j / kShardBlock, cur_shard, A.q.ns_log2);
            if (blk >= A.q.n_blk) continue;
            const uint32_t sid = blk * kShardBlock + (j % kShardBlock);
            if (sid >= A.q.n_items) continue;
            const uint32_t sty = A.q.nsx_shift >= 0 ? sid >> A.q.nsx_shift : sid / A.q.nsx;
            const uint32_t stx = sid - sty * A.q.nsx;
            const int px = (int)stx * FPW + lx;
            const int lrow = (int)sty * FPH + ly;               /* row inside this part's packed rows */
            const bool inside = px < W && lrow < A.rows_local;
            /* packed local row -> frame row (row strips dealt round-robin to parts) */
            int py = lrow;
            if (A.nparts != 1) {
                const int strip = lrow / A.rows_per_strip;
                py = (strip * A.nparts + A.part) * A.rows_per_strip + (lrow - strip * A.rows_per_strip);
            }
            const uint64_t outside_mask = __builtin_amdgcn_ballot_w64(!inside);
            const uint32_t pixel = (uint32_t)(A.out_frame ? py : lrow) * (uint32_t)W + (uint32_t)px;

            float acc[3] = {0.0f, 0.0f, 0.0f};
            T first_nu = T(0);
            int first_it = 0;
            bool alive = false;                                 /* staged: sample continues in the stream pass */

            const int nsamp = aa * aa;
            for (int s = 0; s < nsamp; ++s) {
                int it;
                T nu = T(0);
                float rgb[3] = {0.0f, 0.0f, 0.0f};
                if constexpr (FRACTAL == 0) {
                    /* shaders/mandelbrot.comp:222-226 sample offsets, :149-151 viewport map */
                    T uvx, uvy;
                    if (aa == 1 && A.exact_div_ok) {
                        /* same quotients as the as-written divides, without the divide (div_by) */
                        uvx = div_by<T>((T)px - T(0.5) * resx, resy, inv_h);
                        uvy = div_by<T>((T)py - T(0.5) * resy, resy, inv_h);
                    } else {
                        cold_path();
                        const int sy = s / aa, sx = s - sy * aa;
                        const T pxs = (T)px + (T)sx / (T)aa;
                        const T pys = (T)py + (T)sy / (T)aa;
                        uvx = (pxs - T(0.5) * resx) / resy;
                        uvy = (pys - T(0.5) * resy) / resy;
                    }
                    const T cx = center_x + uvx * zoom;
                    const T cy = center_y + uvy * zoom;
                    if constexpr (!EFFECTS) {
                        Orbit<T> o;
                        o.X = T(0); o.Yd = T(0); o.x2 = T(0); o.y2d = T(0);
                        o.cx = inside ? cx : T(0);
                        o.cyd = inside ? T(2) * cy : T(0);
                        /* |c| <= bailout for every live lane, else the first (tested) block
                         * retires the lane at i = 0 anyway; fast_ok also needs B^2 >= 4.5 */
                        T r2;
                        escape_run<T, false, CLOSE>(o, B2, 0, i1, A.fast_ok != 0, false, outside_mask, it, r2, A.period_window);
                        alive = staged && inside && it >= i1;
                        if (staged) {
                            const T rec[NF] = {o.X, o.Yd, o.cx, o.cyd};
                            writer.append(alive, pixel, (uint32_t)i1, rec);
                        }
                        if (!alive) shade<T, 0>(*kargs(), S, lg, it, r2, want_nu, want_rgb, nu, rgb);
                    } else {
                        T ezx, ezy, min_trap;
                        Orbit<T> o;
                        o.X = T(0); o.Yd = T(0); o.x2 = T(0); o.y2d = T(0);
                        o.cx = cx; o.cyd = T(2) * cy;
                        escape_run_effects<T>(o, B2, max_iter, A.fast_ok != 0, inside,
                                              A.trap_enabled != 0 || A.interior_style == 2, it, ezx, ezy, min_trap);
                        nu = (T)it;
                        if (it < max_iter) {
                            const T log_zn = Real<T>::log(ezx * ezx + ezy * ezy) / T(2);
                            const T mu = Real<T>::log(log_zn / Real<T>::ln2()) / Real<T>::ln2();
                            nu = (T)it + T(1) - mu;
                        }
                        T t = nu / (T)max_iter * (T)S.color_scale;
                        t = t < T(0) ? T(0) : (t > T(1) ? T(1) : t);
                        bool coloured = false;
                        if (it >= max_iter) {                                 /* :182-188 */
                            if (A.interior_style == 1) { coloured = true; }
                            else if (A.interior_style == 2) {
                                const float tf = expf(-(float)min_trap * 6.0f / fmaxf(S.trap_radius, 1e-6f));
                                palette_eval(kargs()->pal, S.pal, S.color_offset + tf * 0.3f, rgb);
                                coloured = true;
                            }
                        }
                        if (!coloured) {
                            palette_eval(kargs()->pal, S.pal, pal_arg(t + (T)S.color_offset), rgb);
                            if (A.trap_enabled) {                             /* :193-198 */
                                const float r = fmaxf(S.trap_radius, 1e-6f);
                                const float tf = expf(-(float)min_trap * 4.0f / r);
                                const float k = clamp01(tf * 0.8f);
                                rgb[0] = rgb[0] * (1.0f - k) + 1.0f * k;
                                rgb[1] = rgb[1] * (1.0f - k) + 0.8f * k;
                                rgb[2] = rgb[2] * (1.0f - k) + 0.4f * k;
                            }
                            if (A.stripe_enabled) {                           /* :201-205 */
                                const T angle = Real<T>::atan2(ezy, ezx);
                                const float sv = 0.5f + 0.5f * (float)Real<T>::sin(angle * (T)S.stripe_density + nu * T(0.3));
                                const float m = 0.7f * (1.0f - sv) + 1.3f * sv;
                                rgb[0] *= m; rgb[1] *= m; rgb[2] *= m;
                            }
                        }
                    }
                } else {
                    /* shaders/julia.comp:325 uv, :221-225 z0, :253-259 sample offsets (sx outer);
                     * shaders/burning_ship.comp:393, :322-325, :337-344 are the same map applied to c */
                    T uvx, uvy;
                    if (aa == 1 && A.exact_div_ok) {
                        uvx = div_by<T>((T)px, resx, inv_w);
                        uvy = div_by<T>((T)py, resy, inv_h);
                    } else {
                        cold_path();
                        uvx = (T)px / resx; uvy = (T)py / resy;
                        if (aa > 1) {
                            const int sx = s / aa, sy = s - sx * aa;
                            const T pixel_size = T(1) / resx;
                            const T sample_offset = pixel_size / (T)aa;
                            const T centre = sample_offset * (T)(aa - 1) * T(0.5);
                            uvx = uvx + ((T)sx * sample_offset - centre) / resx;
                            uvy = uvy + ((T)sy * sample_offset - centre) / resy;
                        }
                    }
                    const T z0x = center_x + (uvx - T(0.5)) * zoom * aspect;
                    const T z0y = center_y + (uvy - T(0.5)) * zoom;
                    if constexpr (FRACTAL == 2 && EFFECTS) {
                        T zx = T(0), zy = T(0), r2, min_dist, stripe_sum;
                        const bool stripes = A.stripe_enabled && A.interior_style == 2;
                        escape_run_ship_effects<T>(zx, zy, inside ? z0x : T(0), inside ? z0y : T(0), B2, max_iter,
                                                   outside_mask, A.trap_enabled != 0, (T)S.trap_radius,
                                                   stripes, (T)S.stripe_density, it, r2, min_dist, stripe_sum);
                        if (it < max_iter) {
                            shade<T, 2>(*kargs(), S, lg, it, r2, want_nu, want_rgb, nu, rgb);
                            if (A.trap_enabled && want_rgb) {                 /* burning_ship.comp:302-306 */
                                const float infl = 1.0f - clamp01((float)min_dist * 2.0f);
                                float tc[3];
                                palette_eval(kargs()->pal, S.pal, infl, tc);
                                const float k = infl * 0.3f;
                                for (int c = 0; c < 3; ++c) rgb[c] = rgb[c] * (1.0f - k) + tc[c] * k;
                            }
                        } else {                                              /* :259-293 interior styles */
                            nu = (T)max_iter;
                            float t = 0.0f, gain = 0.0f;
                            if (A.interior_style == 1 && A.trap_enabled) {
                                t = 1.0f - clamp01((float)min_dist * 5.0f); gain = 0.5f;
                            } else if (stripes) {
                                t = (float)((stripe_sum / (T)max_iter + T(1)) * T(0.5)); gain = 0.3f;
                            } else if (A.interior_style == 3) {
                                t = clamp01((float)Real<T>::sqrt(zx * zx + zy * zy) * 0.5f); gain = 0.4f;
                            }
                            if (gain != 0.0f && want_rgb) {
                                palette_eval(kargs()->pal, S.pal, t, rgb);
                                rgb[0] *= gain; rgb[1] *= gain; rgb[2] *= gain;
                            }
                        }
                    } else {
                        Orbit<T> o;
                        if constexpr (FRACTAL == 1) {
                            o.X = inside ? z0x : T(0);
                            o.Yd = inside ? T(2) * z0y : T(0);
                            o.cx = inside ? (T)S.julia_cx : T(0);
                            o.cyd = inside ? T(2) * (T)S.julia_cy : T(0);
                        } else {
                            o.X = T(0); o.Yd = T(0);
                            o.cx = inside ? z0x : T(0);
                            o.cyd = inside ? T(2) * z0y : T(0);
                        }
                        o.x2 = o.X * o.X;
                        o.y2d = o.Yd * o.Yd;
                        T r2;
                        escape_run<T, Form<FRACTAL>::abs_step, CLOSE>(o, B2, 0, i1, A.fast_ok != 0, false, outside_mask, it, r2, A.period_window);
                        alive = staged && inside && it >= i1;
                        if (staged) {
                            const T rec4[4] = {o.X, o.Yd, o.cx, o.cyd};
                            T rec[NF];
                            for (int k = 0; k < NF; ++k) rec[k] = rec4[k];
                            writer.append(alive, pixel, (uint32_t)i1, rec);
                        }
                        if (!alive) shade<T, FRACTAL>(*kargs(), S, lg, it, r2, want_nu, want_rgb, nu, rgb);
                    }
                }
                if (s == 0) { first_nu = nu; first_it = it; }
                acc[0] += rgb[0]; acc[1] += rgb[1]; acc[2] += rgb[2];
            }

            if (aa > 1) {
                const float n = (float)(aa * aa);
                acc[0] /= n; acc[1] /= n; acc[2] /= n;
            }
            if (want_rgb && (A.flags & FR_FLAG_POST_CHAIN))
                post_chain(acc, S.brightness, S.saturation, S.contrast, FRACTAL != 0);

            if (inside && !alive) {
                if (A.rgba) A.rgba[pixel] = make_float4(acc[0], acc[1], acc[2], 1.0f);
                if (A.nu) reinterpret_cast<T*>(A.nu)[pixel] = first_nu;
                if (A.iter) A.iter[pixel] = first_it;
            }
        }
    }
    if (staged) writer.finish();
    diag_write(A, (uint32_t)lane, diag_t0, diag_items, diag_claims);
#ifdef FR_STAMP
    st_acc[1] = writer.st_block;
#endif
    FR_STAMP_WRITE(A, lane);
}


/* MANDELBROT EFFECTS through the lean tile pass and the lane pool (kernel code FRACTAL = 3): orbit trap, trap-coloured
 * interior, stripe shading -- shaders/mandelbrot.comp:182-205.  The effects variant of the general tile kernel runs such
 * frames in lockstep, one sub-tile at a time, to max_iter, with four running minima per sample; this is its epilogue
 * restated operation for operation (library log of |z|^2, the division by max_iter, atan2 / sin in the kernel's precision),
 * so that both routes give the same planes bit for bit.
 *   STRIPES need the z of the sample's last update: (ezx, ezy) = z after the escaping update, or after max_iter updates for
 *   a sample that never escaped (the lean kernels' code-3 instantiations carry it).
 *   The ORBIT TRAP needs nothing: as the shader is written (:153-166) z starts at 0, the first update makes z = c exactly
 *   (0 * 0 - 0 * 0 + c.x), and the trap takes its minimum AFTER the update, over distToOrigin, distToAxes and
 *   distToC = length(z - c) -- which is 0 at i = 0, for every sample, in any precision.  minTrap is the constant 0 (the
 *   general kernel's four running minima compute exactly that, at 10 of its 16 instructions per update), the trap blend
 *   is a constant mix and the trap-coloured interior a constant colour. */
template <typename T, class ARGS>
__device__ __forceinline__ void shade_stripes(ARGS& A, const LdsBlock& S, const int it, const T ezx, const T ezy, T& nu, float rgb[3])
{
    const int max_iter = A.max_iter;
    const T min_trap = T(0);                                           /* :163-166, see above */
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    nu = (T)it;
    if (it < max_iter) {                                              /* :172-177, as written */
        const T log_zn = Real<T>::log(ezx * ezx + ezy * ezy) / T(2);
        const T mu = Real<T>::log(log_zn / Real<T>::ln2()) / Real<T>::ln2();
        nu = (T)it + T(1) - mu;
    }
    T t = nu / (T)max_iter * (T)S.color_scale;                        /* :179 */
    t = t < T(0) ? T(0) : (t > T(1) ? T(1) : t);
    if (it >= max_iter) {                                             /* :182-188 */
        if (A.interior_style == 1) return;                            /* black */
        if (A.interior_style == 2) {
            const float tf = expf(-(float)min_trap * 6.0f / fmaxf(S.trap_radius, 1e-6f));
            palette_eval(A.pal, S.pal, S.color_offset + tf * 0.3f, rgb);
            return;
        }
    }
    palette_eval(A.pal, S.pal, pal_arg(t + (T)S.color_offset), rgb);  /* :190 */
    if (A.trap_enabled) {                                             /* :193-198 */
        const float r = fmaxf(S.trap_radius, 1e-6f);
        const float tf = expf(-(float)min_trap * 4.0f / r);
        const float k = clamp01(tf * 0.8f);
        rgb[0] = rgb[0] * (1.0f - k) + 1.0f * k;
        rgb[1] = rgb[1] * (1.0f - k) + 0.8f * k;
        rgb[2] = rgb[2] * (1.0f - k) + 0.4f * k;
    }
    if (A.stripe_enabled) {                                           /* :201-205 */
        const T angle = Real<T>::atan2(ezy, ezx);
        const float sv = 0.5f + 0.5f * (float)Real<T>::sin(angle * (T)S.stripe_density + nu * T(0.3));
        const float m = 0.7f * (1.0f - sv) + 1.3f * sv;
        rgb[0] *= m; rgb[1] *= m; rgb[2] *= m;
    }
}

/* ---- control block + coordinate tables ---------------------------------------------------------------------------
 * One small launch in front of every render (replaces clear_words_kernel there): zeroes the queue heads / stream
 * counters and writes the two coordinate tables of the lean tile pass.  The viewport map is separable -- Re c depends on
 * the pixel's column only, Im c on its row only -- so the 15 fp64 operations (two conversions, two 3-op quotients, two
 * multiply-adds, the doubling) every pixel of the tile pass used to spend on it are W + H table entries instead, each
 * computed by the as-written expression of the shader (true IEEE divides: the exact-division check is not needed here):
 *   MAP 0  shaders/mandelbrot.comp:149-151   uv = (pix - 0.5 res) / res.y;  c = center + uv * zoom
 *   MAP 1  shaders/julia.comp:325, :221-225 (= burning_ship.comp:393, :322-325)
 *                                            uv = pix / res;  p = center + (uv - 0.5) * zoom * (aspect, 1)
 * T narrows center / zoom as the reference narrows them for its fp32 shaders (src/compute_effect_manager.h:85-90). */
/* the W + H table entries i = first, first + stride, ...
 * THROUGH: the entries are written through to memory (system-scope stores) -- for readers on other XCDs inside the SAME
 * launch (lean_prologue): the eight L2s of the device are not coherent with each other, ordinary stores stay in the
 * writer's, and the agent-scope release fence that would publish them writes back EVERYTHING that L2 holds dirty (the
 * previous frame's pixels: measured, +35 us on every frame). */
template <bool THROUGH, typename T>
__device__ __forceinline__ void put_entry(T* p, const T v)
{
    if constexpr (THROUGH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *p = v;
}
template <typename T, int MAP, bool THROUGH, class ARGS>
__device__ __forceinline__ void write_coordinate_tables(ARGS& A, const uint32_t first, const uint32_t stride, const uint32_t end)
{
    T* __restrict__ xs = reinterpret_cast<T*>(A.xs);
    T* __restrict__ yds = reinterpret_cast<T*>(A.yds);
    if (!xs) return;
    const T center_x = (T)A.center_x, center_y = (T)A.center_y, zoom = (T)A.zoom;
    const T aspect = sizeof(T) == 8 ? (T)A.aspect_d : (T)A.aspect_f;
    if (A.ssaa > 1) {
        /* the sample grid of a supersampled frame: column i = sample sx = i % aa of pixel column px = i / aa (rows alike),
         * by the shaders' own sample expressions -- mandelbrot.comp:222-226 (offsets (sx, sy) / aa, top-left anchored),
         * julia.comp:253-259 = burning_ship.comp:337-344 (centred offsets of 1 / (W aa), divided again by the size) -- as
         * the SSAA arm of tile_kernel evaluates them per sample */
        const int aa = A.ssaa;
        const T resx = (T)A.res_w, resy = (T)A.res_h;
        for (uint32_t i = first; i < end; i += stride) {
            const bool col = i < (uint32_t)A.W;
            const int g = col ? (int)i : (int)(i - (uint32_t)A.W);
            const int pq = g / aa, sq = g - pq * aa;                 /* pixel and sample index along this axis */
            if constexpr (MAP == 0) {
                const T ps = (T)pq + (T)sq / (T)aa;
                if (col) { const T uvx = (ps - T(0.5) * resx) / resy; put_entry<THROUGH>(&xs[g], (T)(center_x + uvx * zoom)); }
                else { const T uvy = (ps - T(0.5) * resy) / resy; put_entry<THROUGH>(&yds[g], (T)(T(2) * (center_y + uvy * zoom))); }
            } else {
                const T pixel_size = T(1) / resx;
                const T sample_offset = pixel_size / (T)aa;
                const T centre = sample_offset * (T)(aa - 1) * T(0.5);
                if (col) {
                    T uvx = (T)pq / resx;
                    uvx = uvx + ((T)sq * sample_offset - centre) / resx;
                    put_entry<THROUGH>(&xs[g], (T)(center_x + (uvx - T(0.5)) * zoom * aspect));
                } else {
                    T uvy = (T)pq / resy;
                    uvy = uvy + ((T)sq * sample_offset - centre) / resy;
                    put_entry<THROUGH>(&yds[g], (T)(T(2) * (center_y + (uvy - T(0.5)) * zoom)));
                }
            }
        }
        return;
    }
    const T resx = (T)A.W, resy = (T)A.H;
    for (uint32_t i = first; i < end; i += stride) {
        if (i < (uint32_t)A.W) {
            const int px = (int)i;
            if constexpr (MAP == 0) {
                const T uvx = ((T)px - T(0.5) * resx) / resy;
                put_entry<THROUGH>(&xs[px], (T)(center_x + uvx * zoom));
            } else {
                const T uvx = (T)px / resx;
                put_entry<THROUGH>(&xs[px], (T)(center_x + (uvx - T(0.5)) * zoom * aspect));
            }
        } else {
            const int py = (int)(i - (uint32_t)A.W);
            if constexpr (MAP == 0) {
                const T uvy = ((T)py - T(0.5) * resy) / resy;
                put_entry<THROUGH>(&yds[py], (T)(T(2) * (center_y + uvy * zoom)));
            } else {
                const T uvy = (T)py / resy;
                put_entry<THROUGH>(&yds[py], (T)(T(2) * (center_y + (uvy - T(0.5)) * zoom)));
            }
        }
    }
}

template <typename T, int MAP>
__global__ void __launch_bounds__(kBlockThreads)
prepare_kernel(const LaunchArgs A, uint32_t* __restrict__ ctrl, const uint32_t n_ctrl, const Feedback fb)
{
    forward_feedback(fb);
    const uint32_t stride = gridDim.x * kBlockThreads, first = blockIdx.x * kBlockThreads + threadIdx.x;
    for (uint32_t i = first; i < n_ctrl; i += stride) ctrl[i] = 0u;
    write_coordinate_tables<T, MAP, false>(A, first, stride, (uint32_t)(A.W + A.H));
}

/* PROLOGUE of the lean tile pass.  A render used to be prepare_kernel -> tile pass (-> lane pool): a 4.5 us launch and the
 * ~5 us it takes a dependent kernel to start behind it, in front of EVERY frame -- a fifth of a 1080p frame at max_iter 256
 * (the reference's interactive default), a quarter of a 512 x 512 one.  Here the first pro_n workgroups of the tile pass do
 * that work themselves (control words, the W + H coordinates, the feedback word) and publish it: each ends with a release
 * fence and two atomics on ONE word -- max(word, epoch << 4), then + 1 -- so the word reads (epoch << 4) | pro_n exactly when
 * all of them are through, whatever an earlier, failed launch left in it.  Every workgroup (the preparing ones too) waits
 * for that value before its first queue claim.  The control words are written and read by atomics only; the tables are
 * written through to memory (put_entry) and first read, on any XCD, after the wait.
 * No deadlock: workgroups start in index order, so the preparing ones are never behind a waiting one; the grid of a
 * persistent kernel is resident as a whole anyway.  And no unbounded wait: a workgroup that polls a million times (of the order of a second) sets the
 * context's error word (the host fails the render, FR_ERR_INTERNAL) and leaves without touching a queue.
 * The epoch is the host's per-context render count, so a captured launch cannot be replayed: the host takes the separate
 * prepare_kernel launch on capturing streams. */
template <typename T, int MAP>
__device__ __forceinline__ void lean_prologue_produce()
{
    KArgs K = kargs();
    if (blockIdx.x < K->pro_n) {
        const uint32_t stride = K->pro_n * kBlockThreads, first = blockIdx.x * kBlockThreads + threadIdx.x;
        for (uint32_t i = first; i * (uint32_t)kShardStrideWords < K->pro_ctrl_words; i += stride)
            __hip_atomic_store(K->pro_ctrl + (size_t)i * kShardStrideWords, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        {
            /* a preparing workgroup's entries are a contiguous run of whole 128-byte lines (the tables are one array, xs then
             * yds): a line written from two XCDs could sit in one's L2 with the other's half stale */
            const uint32_t n = (uint32_t)(K->W + K->H);
            const uint32_t chunk = ((n + K->pro_n - 1u) / K->pro_n + 31u) & ~31u;
            const uint32_t lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
            write_coordinate_tables<T, MAP, true>(*K, lo + threadIdx.x, (uint32_t)kBlockThreads, hi);
        }
        /* every store above is a write-through (atomic) store: complete when the counter says so -- no L2 write-back */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_max(K->pro_ready, K->pro_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(K->pro_ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        /* the previous render's cycle-closing verdict: nobody in this launch waits for it (its words belong to the lane
         * pool, a later launch), so it goes behind the publication -- its loads are a round trip to memory */
        Feedback fb;
        fb.dev_flag = K->pro_fb_flag; fb.host_word = K->pro_fb_host; fb.prev_seq = K->pro_prev_seq;
        forward_feedback(fb);                                            /* wave 0 of workgroup 0 */
    }
}
/* ... and the wait, after the workgroup has staged its constants (which the preparing workgroups do while their stores are
 * on their way) */
__device__ __forceinline__ bool lean_prologue_wait(const LaunchArgs& A, uint32_t* lds_word)
{
    if (threadIdx.x == 0) {
        const uint32_t target = A.pro_epoch + A.pro_n;
        uint32_t ok = 1u;
        for (uint32_t polls = 0; __hip_atomic_load(A.pro_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target; ++polls) {
            if (polls > (1u << 20)) {                                    /* of the order of a second: this is a bug, say so */
                if (A.out.overflow) __hip_atomic_store(A.out.overflow, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        *lds_word = ok;
    }
    /* no acquire fence (it would invalidate the whole L2, for every workgroup): no cache of this XCD can hold a line of the
     * tables -- the launch began with all of them invalidated, and nobody reads a table before this point */
    __syncthreads();
    return *lds_word != 0u;
}

/* ---- staged SSAA: the average of a pixel's samples ------------------------------------------------------------------
 * The sample grid (W aa x rows aa samples, packed local rows) was rendered like any frame -- lean tile pass + lane pool --
 * into scratch planes; a pixel is the sum of its aa x aa sample colours in the shader's order (mandelbrot.comp:219-230: sy
 * outer; julia.comp:253-262, burning_ship.comp:337-347: sx outer), divided by aa^2, then the post chain if asked for;
 * nu and iter are those of sample (0, 0), as in the SSAA arm of tile_kernel.  HBM-bound: 16 aa^2 B read + 16 B written per
 * pixel. */
struct SsaaArgs {
    const float4* s_rgba; const void* s_nu; const int32_t* s_iter;   /* sample planes (packed local sample rows) */
    float4* rgba; void* nu; int32_t* iter;                           /* the frame's planes */
    int32_t W, rows_local, aa, sx_outer;
    int32_t part, nparts, rows_per_strip, out_frame;                 /* where local row r lives in a whole-frame plane */
    uint32_t flags; float brightness, saturation, contrast; int32_t julia_floors;
};
template <typename T>
__global__ void __launch_bounds__(kBlockThreads)
ssaa_reduce_kernel(const SsaaArgs A)
{
    const size_t npx = (size_t)A.W * (size_t)A.rows_local;
    const int aa = A.aa;
    const size_t Ws = (size_t)A.W * (size_t)aa;
    for (size_t i = (size_t)blockIdx.x * kBlockThreads + threadIdx.x; i < npx; i += (size_t)gridDim.x * kBlockThreads) {
        const uint32_t lrow = (uint32_t)(i / (size_t)A.W), px = (uint32_t)(i - (size_t)lrow * A.W);
        uint32_t row = lrow;
        if (A.out_frame && A.nparts != 1) {
            const uint32_t strip = lrow / (uint32_t)A.rows_per_strip;
            row = (strip * (uint32_t)A.nparts + (uint32_t)A.part) * (uint32_t)A.rows_per_strip + (lrow - strip * (uint32_t)A.rows_per_strip);
        }
        const size_t o = (size_t)row * A.W + px;
        const size_t s00 = (size_t)lrow * aa * Ws + (size_t)px * aa;
        if (A.rgba) {
            float acc[3] = {0.0f, 0.0f, 0.0f};
            for (int s = 0; s < aa * aa; ++s) {
                const int a = s / aa, b = s - a * aa;                /* outer, inner */
                const int sy = A.sx_outer ? b : a, sx = A.sx_outer ? a : b;
                const float4 v = A.s_rgba[s00 + (size_t)sy * Ws + sx];
                acc[0] += v.x; acc[1] += v.y; acc[2] += v.z;
            }
            const float n = (float)(aa * aa);
            acc[0] /= n; acc[1] /= n; acc[2] /= n;
            if (A.flags & FR_FLAG_POST_CHAIN) post_chain(acc, A.brightness, A.saturation, A.contrast, A.julia_floors != 0);
            A.rgba[o] = make_float4(acc[0], acc[1], acc[2], 1.0f);
        }
        if (A.nu) reinterpret_cast<T*>(A.nu)[o] = reinterpret_cast<const T*>(A.s_nu)[s00];
        if (A.iter) A.iter[o] = A.s_iter[s00];
    }
}

/* ---- tile pass, lean form ------------------------------------------------------------------------------------------
 * The default path of every one-sample render without effects (8x8 sub-tiles): same queue, same survivor stream, same
 * per-lane arithmetic and therefore the same planes, bit for bit, as tile_kernel -- written for the instruction count.
 * The tile pass is a PER-PIXEL cost (C2: 10 updates per pixel on average, then a smooth count, a palette and a store),
 * and it is VALU-issue bound: every VALU instruction is ~4.3 cycles of its SIMD whatever its width (tools/ubench2.hip:
 * only a handful of 32-bit ops run at 2.3), scalar instructions ride along for free as long as they are fewer.  What
 * tile_kernel spends per 64-pixel sub-tile -- 239 VALU + 138 SALU on C2, 200 VALU on a view where every pixel escapes
 * at once -- is mostly not arithmetic: SGPR spills (v_readlane / v_writelane are VALU slots), selects and exec-mask
 * juggling around values that are wave-uniform but were read from LDS, parking moves, the viewport map.  Here:
 *   - coordinates come from the two tables of prepare_kernel (2 loads instead of 15 fp64 operations per pixel);
 *   - a finished lane is not parked at z = 0, c = 0 (12 moves per escape event in fp64): its escape threshold becomes
 *     NaN (one v_or), it iterates on whatever it holds and never compares as escaped again (escape_run_lean);
 *   - sub-tile indices are advanced incrementally along the run (one division per 16 sub-tiles at most);
 *   - the colour stage reads its control values from the kernel arguments (palette_eval);
 *   - a sub-tile that follows one in which no lane escaped starts in unchecked blocks (interior regions: 16 tested
 *     updates = 32 VALU + 64 SALU less per sub-tile). */
/* NP samples per lane (NP sub-tiles per wave and trip): the wave-uniform control -- loop counters, branches, exec-mask
 * handling, argument re-reads -- is paid once per NP x 64 pixels.  Scalar and vector issue hardly overlap in this code
 * (every few instructions one waits for the other: v_cmp -> branch, exec write -> VALU; measured: VALU and SALU + branch
 * cycles add up to ~85 % of the pass), so halving the scalar work per pixel is worth as much as removing vector work. */
template <typename T, int NP, bool ABS, bool PERIOD, bool ESCZ = false>
__device__ __forceinline__ int escape_run_lean(Orbit<T> (&o)[NP], const T B2x4, const int i1, const bool fast_ok, bool fast,
                                               const bool (&lane_off)[NP], int (&esc_i)[NP], T (&esc_r2x4)[NP],
                                               uint64_t (&done)[NP], const uint32_t period_window,
                                               const int exit_from, const uint32_t exit_cost,
                                               T (*esc_X)[NP] = nullptr, T (*esc_Yd)[NP] = nullptr)   /* ESCZ: z at the escape */
{
    using Bits = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
    constexpr Bits kNaNBits = sizeof(T) == 8 ? (Bits)0x7FF8000000000000ull : (Bits)0x7FC00000u;
    T thr[NP];                                             /* per-lane escape threshold; NaN: the lane is finished */
    uint64_t all = ~0ull;                                  /* AND of the done masks */
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        esc_i[p] = i1;
        esc_r2x4[p] = T(0);
        thr[p] = lane_off[p] ? (T)__builtin_nanf("") : B2x4;
        done[p] = __builtin_amdgcn_ballot_w64(lane_off[p]);
        all &= done[p];
    }
    int i = 0;                                             /* wave-uniform: SGPR */
    T refX[NP], refYd[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { refX[p] = __builtin_nan(""); refYd[p] = refX[p]; }
    uint32_t snap_window = period_window, snap_closed = 1u;
    const uint32_t snap_cap = period_window > (((uint32_t)i1 >> 7) << 4) ? period_window : (((uint32_t)i1 >> 7) << 4);
    int next_snap = 0;
    auto finish = [&](T& t) { Bits b; __builtin_memcpy(&b, &t, sizeof(T)); b |= kNaNBits; __builtin_memcpy(&t, &b, sizeof(T)); };
    /* PERIOD: see escape_run; a lane back at its own snapshot never escapes -> finished as "alive at i1" */
    auto close_cycles = [&]() {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bool hit = o[p].X == refX[p] && o[p].Yd == refYd[p];
            const uint64_t hm = __builtin_amdgcn_ballot_w64(hit) & ~done[p];     /* finished lanes hold garbage */
            if (hm != 0ull) {
                if (hit) finish(thr[p]);
                done[p] |= hm;
                snap_closed += (uint32_t)__builtin_popcountll(hm);
            }
        }
        if (i >= next_snap) {
#pragma unroll
            for (int p = 0; p < NP; ++p) { refX[p] = o[p].X; refYd[p] = o[p].Yd; }
            if (snap_closed == 0u && snap_window < snap_cap) snap_window <<= 1;
            snap_closed = 0u;
            next_snap = i + (int)snap_window;
        }
        all = ~0ull;
#pragma unroll
        for (int p = 0; p < NP; ++p) all &= done[p];
    };
    while (i < i1) {
        if (all == ~0ull) break;                           /* every lane finished: wave-uniform early-out */
        const int left = i1 - i;
        /* OCCUPANCY EXIT (staged passes).  The trip pays NP x 64 slots per update whatever the number of live samples; the
         * lane pool pays per live sample, plus `exit_cost` updates' worth of handling per record (stream write and read,
         * refill, half a stretch idle, a replay of its escape).  Once the live samples are few enough that handing them
         * over now is the cheaper side of that -- alive x (left + exit_cost) < slots x left -- the trip ends here and its
         * survivors go to the stream with the updates they have run (the pool honours every record's own count).  Same
         * operations per sample in the same order: only WHERE a sample's later updates run changes, never a pixel. */
        if (exit_cost != 0u && i >= exit_from) {
            uint32_t fin = 0;
#pragma unroll
            for (int p = 0; p < NP; ++p) fin += (uint32_t)__builtin_popcountll(done[p]);
            const uint32_t alive = (uint32_t)(NP * 64) - fin;
            if (alive * ((uint32_t)left + exit_cost) < (uint32_t)(NP * 64) * (uint32_t)left) break;
        }
        if (fast && left >= kFastBlock) {
            Orbit<T> snap[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) snap[p] = o[p];
#pragma unroll
            for (int k = 0; k < kFastBlock; ++k) {
#pragma unroll
                for (int p = 0; p < NP; ++p) orbit_step<T, ABS>(o[p]);
            }
            uint64_t badm = 0ull;
#pragma unroll
            for (int p = 0; p < NP; ++p) badm |= __builtin_amdgcn_ballot_w64(!(orbit_r2x4(o[p]) <= B2x4)) & ~done[p];
            if (badm == 0ull) {
                i += kFastBlock;
                if constexpr (PERIOD) { if (period_window) close_cycles(); }
                continue;
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) o[p] = snap[p];   /* roll back, replay tested */
            fast = false;
        }
        int end = i + (left < kFastBlock ? left : kFastBlock);
        bool event = false;
        do {
            T r2x4[NP];
            bool e[NP];
            uint64_t m[NP], em = 0ull;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                orbit_step<T, ABS>(o[p]);
                r2x4[p] = orbit_r2x4(o[p]);
                e[p] = r2x4[p] > thr[p];                   /* false for ever once thr is NaN */
                m[p] = __builtin_amdgcn_ballot_w64(e[p]);
                em |= m[p];
            }
            if (em != 0ull) {
                all = ~0ull;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (e[p]) {
                        esc_i[p] = i; esc_r2x4[p] = r2x4[p]; finish(thr[p]);
                        if constexpr (ESCZ) { (*esc_X)[p] = o[p].X; (*esc_Yd)[p] = o[p].Yd; }
                    }
                    done[p] |= m[p];
                    all &= done[p];
                }
                event = true;
                if (all == ~0ull) end = i;                 /* single-exit loop: this was the last update */
            }
            ++i;
        } while (i < end);
        if (all == ~0ull) break;
        fast = fast_ok && !event;
        if constexpr (PERIOD) { if (period_window) close_cycles(); }
    }
    return i;                                              /* updates the trip's unfinished samples have run */
}

/* WaveQueue with only its three words of state kept between calls (home shard, shards tried, head last seen) */
struct LeanQueue {
    uint32_t shard, tried, seen;
    __device__ __forceinline__ void init()
    {
        KArgs K = kargs();
        const uint32_t p = (K->q.flags >> kQueueProbeShift) & 0xFu, nsh = 1u << K->q.ns_log2;
        shard = (p == 0u || p >= nsh) ? WaveQueue::home_of(nsh) : (blockIdx.x & (nsh - 1u));
        tried = 0; seen = 0;
    }
    /* STREAM: the shards are the regions of the input stream (lengths written by the previous launch) */
    template <bool STREAM = false>
    __device__ __forceinline__ bool next(uint32_t lane, uint32_t& begin, uint32_t& count, uint32_t& sh, uint32_t even = 0u)
    {
        KArgs K = kargs();
        WaveQueue q;
        if constexpr (STREAM)
            q.init_lengths(K->q.heads, K->in.n_blocks, K->in.region_blocks, K->q.run_shift, K->q.run_min, K->q.run_max, lane, K->q.ns_log2);
        else
            q.init(K->q.heads, K->q.n_blk, (uint32_t)kShardBlock, K->q.run_shift, K->q.run_min, K->q.run_max, lane, K->q.ns_log2);
        q.run_even = even;
        const uint32_t p = (K->q.flags >> kQueueProbeShift) & 0xFu;
        if (p != 0u && p < q.ns) q.max_tries = p;
        q.shard = shard; q.tried = tried; q.seen = seen;
        const bool got = q.next(begin, count, sh);
        shard = q.shard; tried = q.tried; seen = q.seen;
        return got;
    }
};

/* RingWriter with the stream description re-read from the kernel arguments at every block (once per 64 survivors) */
template <typename T, int NF>
struct LeanWriter {
    uint32_t head, tail, home;
#ifdef FR_STAMP
    uint64_t st_block = 0;
#endif
    __device__ __forceinline__ void init() { head = tail = 0; home = WaveQueue::home_of(kargs()->out.nregions ? kargs()->out.nregions : (uint32_t)kShards); }
    __device__ __forceinline__ void write_block(WaveRing<T, NF>* ring, uint32_t lane, uint32_t nvalid)
    {
        KArgs K = kargs();
        RingWriter<T, NF> w;
        w.ring = ring; w.lane = lane;
        w.out.base = K->out.base; w.out.n_blocks = K->out.n_blocks; w.out.region_blocks = K->out.region_blocks;
        w.out.rotate = K->out.rotate; w.out.overflow = K->out.overflow; w.out.nregions = K->out.nregions; w.head = head; w.tail = tail; w.home = home;
        w.write_block(nvalid);
        head = w.head; home = w.home;
#ifdef FR_STAMP
        st_block += w.st_block;
#endif
    }
    __device__ __forceinline__ void append(WaveRing<T, NF>* ring, uint32_t lane, bool keep, uint32_t pixel, uint32_t done, const T (&v)[NF])
    {
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        if (m == 0ull) return;
        if (keep) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t slot = (tail + rank) & (kRingSlots - 1);
            ring->pix[slot] = pixel;
            ring->it[slot] = done;
#pragma unroll
            for (int k = 0; k < NF; ++k) ring->f[k][slot] = v[k];
        }
        tail += (uint32_t)__builtin_popcountll(m);
        __builtin_amdgcn_wave_barrier();
        if (tail - head >= 64u) write_block(ring, lane, 64u);
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void finish(WaveRing<T, NF>* ring, uint32_t lane)
    {
        const uint32_t left = tail - head;
        if (left > 0u) write_block(ring, lane, left);
    }
};

/* NP = 1: one 8x8 sub-tile per trip.  NP = 2: two horizontally adjacent ones (a 16x8 patch, lane (lx, ly) holds the
 * pixels (lx, ly) and (lx + 8, ly)); a trip whose second sub-tile does not exist (odd run, end of a row) runs with those
 * samples switched off like lanes outside the frame. */
template <typename T, int FRACTAL, bool PERIOD, int NP>
__global__ void __launch_bounds__(kBlockThreads)
tile_lean_kernel(const LaunchArgs A)
{
    constexpr int NF = RecFields<FRACTAL>::n;
    constexpr bool ABS = Form<FRACTAL>::abs_step;
    constexpr bool STRIPES = FRACTAL == 3;                    /* Mandelbrot + stripe shading: see shade_stripes */
    constexpr int FR = STRIPES ? 0 : FRACTAL;                 /* the fractal proper */

    __shared__ LdsBlock S;
    __shared__ WaveRing<T, NF> rings[kWavesPerBlock];
    stage_constants(S, A);
    /* (behind stage_constants: in front of it, its atomics stand between the argument block and the copy of its palette table
     * to LDS, the optimiser then keeps that table in scratch -- 176 B per lane, 78 VGPRs for the fp32 kernel) */
    if (A.pro_ready) lean_prologue_produce<T, FR == 0 ? 0 : 1>();
    stage_interior<T, FR>(S, A);
    __shared__ double2 log2_lds[sizeof(T) == 8 ? kLog2Entries : 1];
    const LogTab<T> lg = stage_log2<T>(log2_lds, A);
    __shared__ uint32_t pro_ok;
    if (A.pro_ready) {
        if (!lean_prologue_wait(A, &pro_ok)) return;
    }

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    WaveRing<T, NF>* const ring = &rings[threadIdx.x >> 6];
    const int max_iter = A.max_iter, i1 = A.i1;
    const bool staged = i1 < max_iter;                       /* survivors continue in the lane pool */
    const bool fast_ok = A.fast_ok != 0;
    const T B2x4 = sizeof(T) == 8 ? (T)A.b2x4_d : (T)A.b2x4_f;
    const bool want_rgb = A.rgba != nullptr;
    const bool want_nu = want_rgb || A.nu != nullptr;
    const T* __restrict__ xs = reinterpret_cast<const T*>(A.xs);
    const T* __restrict__ yds = reinterpret_cast<const T*>(A.yds);
    const uint32_t W = (uint32_t)A.W, rows_local = (uint32_t)A.rows_local, nsx = A.q.nsx;
    const bool out_frame = A.out_frame != 0;

    LeanWriter<T, NF> writer;
    writer.init();
    LeanQueue q;
    q.init();

    uint64_t diag_t0 = 0;
    uint32_t diag_items = 0, diag_claims = 0;
    if (kargs()->diag) diag_t0 = __builtin_amdgcn_s_memrealtime();

    /* frame row of the first row of sub-tile row sty (row strips dealt round-robin to parts; the host sends sharded
     * frames here only when a strip is a whole number of sub-tile rows) */
    auto first_row = [&](uint32_t sty) -> uint32_t {
        const uint32_t lrow0 = sty * 8u;
        KArgs K = kargs();
        if (K->nparts == 1) return lrow0;
        const uint32_t R = (uint32_t)K->rows_per_strip, strip = lrow0 / R;
        return (strip * (uint32_t)K->nparts + (uint32_t)K->part) * R + (lrow0 - strip * R);
    };

    uint32_t begin, count, cur_shard;
    FR_STAMP_DECL
    for (;;) {
        FR_STAMP_BEGIN();
        const bool got_run = q.next(lane, begin, count, cur_shard, NP == 2 ? 1u : 0u);
        FR_STAMP_END(0);
        if (!got_run) break;
        ++diag_claims;
        diag_items += count;
        /* the run, block by block (blocks of 16 consecutive sub-tiles are dealt round-robin to the shards); everything
         * here is wave-uniform: SALU */
        uint32_t j = begin;
        const uint32_t jend = begin + count;
        while (j < jend) {
            uint32_t n, sid, stx, sty;
            {
                KArgs K = kargs();
                const uint32_t blk = WaveQueue::block_of(j / kShardBlock, cur_shard, K->q.ns_log2);
                if (blk >= K->q.n_blk) break;
                uint32_t jb = (j | (uint32_t)(kShardBlock - 1)) + 1u;
                jb = jb < jend ? jb : jend;
                sid = blk * kShardBlock + (j % kShardBlock);
                n = jb - j;
                j = jb;
                const uint32_t n_items = K->q.n_items;
                if (sid >= n_items) break;
                n = sid + n > n_items ? n_items - sid : n;
                sty = K->q.nsx_shift >= 0 ? sid >> K->q.nsx_shift : sid / nsx;
                stx = sid - sty * nsx;
            }
            uint32_t py0 = first_row(sty);
            bool hint_fast = false;                          /* the previous trip lost no lane: start unchecked */
            do {
                /* sub-tiles of this trip: stx .. stx + np - 1 of row sty */
                uint32_t np = n < (uint32_t)NP ? n : (uint32_t)NP;
                np = stx + np > nsx ? nsx - stx : np;
                const uint32_t lrow = sty * 8u + ly;
                const bool row_in = lrow < rows_local;
                const T tyd = yds[row_in ? py0 + ly : 0u];     /* table reads: lanes outside the frame read entry 0 */
                uint32_t pixel[NP];
                bool inside[NP], lane_off[NP];
                Orbit<T> o[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const uint32_t px = (stx + (uint32_t)p) * 8u + lx;
                    inside[p] = (uint32_t)p < np && px < W && row_in;
                    lane_off[p] = !inside[p];
                    pixel[p] = (out_frame ? py0 + ly : lrow) * W + px;       /* where the pixel's planes entries are */
                    const T tx = xs[inside[p] ? px : 0u];
                    if constexpr (FR == 1) {
                        o[p].X = tx; o[p].Yd = tyd;
                        o[p].cx = (T)S.julia_cx; o[p].cyd = T(2) * (T)S.julia_cy;
                    } else {
                        o[p].X = T(0); o[p].Yd = T(0);
                        o[p].cx = tx; o[p].cyd = tyd;
                    }
                    o[p].x2 = o[p].X * o[p].X;
                    o[p].y2d = o[p].Yd * o[p].Yd;
                }

                int it[NP];
                T r2x4[NP];
                uint64_t done[NP];
                T esc_X[NP], esc_Yd[NP];                         /* STRIPES: z at the escape */
                int i_end;
                if constexpr (STRIPES)
                    i_end = escape_run_lean<T, NP, ABS, PERIOD, true>(o, B2x4, i1, fast_ok, hint_fast && fast_ok, lane_off, it, r2x4, done,
                                                                      PERIOD ? A.period_window : 0u, A.exit_from, staged ? A.exit_cost : 0u,
                                                                      &esc_X, &esc_Yd);
                else
                    i_end = escape_run_lean<T, NP, ABS, PERIOD>(o, B2x4, i1, fast_ok, hint_fast && fast_ok, lane_off, it, r2x4, done,
                                                                PERIOD ? A.period_window : 0u, A.exit_from, staged ? A.exit_cost : 0u);
                bool lost = false, need_any = false;
                bool alive[NP], need[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    lost = lost || done[p] != __builtin_amdgcn_ballot_w64(lane_off[p]);
                    alive[p] = staged && inside[p] && it[p] >= i1;
                    need[p] = inside[p] && !alive[p];
                    need_any = need_any || __builtin_amdgcn_ballot_w64(need[p]) != 0ull;
                }
                hint_fast = !lost;
                if (staged) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const T rec4[4] = {o[p].X, o[p].Yd, o[p].cx, o[p].cyd};
                        T rec[NF];
                        for (int k = 0; k < NF; ++k) rec[k] = rec4[k];
                        writer.append(ring, lane, alive[p], pixel[p], (uint32_t)i_end, rec);
                    }
                }
                if (need_any) {
                    /* every lane runs the colour stage of all its samples (what a lane without a finished sample
                     * computes is dropped): one pass through its wave-uniform branches for NP x 64 pixels */
                    KArgs K = kargs();
                    T nu[NP];
                    float rgb[NP][3];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        if constexpr (STRIPES) {
                            /* a sample that never escaped: its z after max_iter updates (one-pass frames; a staged pass hands
                             * such samples on) */
                            const bool esc = it[p] < i1;
                            shade_stripes<T>(*K, S, it[p], esc ? esc_X[p] : o[p].X, T(0.5) * (esc ? esc_Yd[p] : o[p].Yd), nu[p], rgb[p]);
                        } else {
                            shade<T, FR>(*K, S, lg, it[p], T(0.25) * r2x4[p], want_nu, want_rgb, nu[p], rgb[p]);
                        }
                        if (want_rgb && (K->flags & FR_FLAG_POST_CHAIN))
                            post_chain(rgb[p], S.brightness, S.saturation, S.contrast, FR != 0);
                    }
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        if (need[p]) {
                            if (K->rgba) K->rgba[pixel[p]] = make_float4(rgb[p][0], rgb[p][1], rgb[p][2], 1.0f);
                            if (K->nu) reinterpret_cast<T*>(K->nu)[pixel[p]] = nu[p];
                            if (K->iter) K->iter[pixel[p]] = it[p];
                        }
                    }
                }
                stx += np;
                n -= np;
                if (stx == nsx) { stx = 0; ++sty; py0 = first_row(sty); hint_fast = false; }
            } while (n != 0u);
        }
    }
    if (staged) writer.finish(ring, lane);
    {
        KArgs K = kargs();
        if (K->diag && lane == 0) {      /* diagnostics: per-wave timeline (100 MHz ticks) and work counts */
            const uint32_t wave_id = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
            uint64_t* d = K->diag + (size_t)wave_id * kDiagWords;
            d[0] = diag_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = diag_items; d[3] = diag_claims;
        }
    }
#ifdef FR_STAMP
    st_acc[1] = writer.st_block;
#endif
    FR_STAMP_WRITE(A, lane);
}

/* ---- lane pool ------------------------------------------------------------------------------------
 * Persistent LANES: a lane that finishes its sample is refilled with the next survivor record of the wave's
 * reserve (blocks of 64 records claimed a run at a time from the region queues of the survivor stream),
 * so a wave stays full whatever the spread of escape times (a 64-pixel tile of a Julia dust runs
 * 25 % full when it must wait for its slowest lane) and the frame balances at pixel granularity.
 *
 *   - wclock: iterations this wave has run (wave-uniform, SGPR).  A lane refilled at wclock = s with a record that
 *     has run d updates has deadline = s + max_iter - d; it escapes with index  wclock - (deadline - max_iter)  or is
 *     interior when wclock reaches its deadline.  The earliest deadline is kept wave-uniform, so reaching it costs one
 *     scalar compare per iteration and nothing per lane.
 *   - unchecked blocks of 16 updates stay legal at ANY alignment: if the block is clean (no lane
 *     escaped in all 16 updates) then a lane whose deadline fell inside the block did not escape
 *     before its deadline either -> interior, exactly.  A dirty block is rolled back and replayed
 *     tested, where deadlines are honoured to the iteration.
 *   - finished lanes wait (parked at z = 0, c = 0) until `refill_at` lanes are idle, then they are
 *     shaded, stored (16-byte scattered stores; neighbours in tile order finish close in time and
 *     merge in L2) and refilled together, so the per-pixel code runs reasonably full.
 * Results are bit-identical to the tile pass: same per-lane operation sequence. */
/* DEFERRED LOCATION of escapes (round 4).  An unchecked stretch that ends with some lanes beyond the bailout used to be
 * followed by locate_escapes: the WHOLE wave replayed the stretch, tested, for the two or three lanes that needed their
 * escape index -- and after two such stretches in a row the wave fell back to per-update tests for everybody (6 VALU + 4-6
 * scalar instructions and two branches per update instead of 6 VALU): on escape-dense survivors (the C5 view, the C3 dust)
 * 41 % of all updates ran tested and another 14 % were replayed.  Now the escaped lanes' stretch-start states go into a
 * per-wave ring in LDS {pixel, iteration index at the stretch's start, X, Yd (, cx, cyd)}, the lanes are free for the next
 * refill at once, the wave STAYS in unchecked stretches, and whenever 64 entries are queued lane l takes entry l and the
 * wave runs ONE tested replay at full occupancy (at most the longest stretch among them: 16-64 updates), shades and stores.
 * Same operations on the same values in the same order as the stretch itself: the planes do not change by a bit. */
constexpr int kDeferSlots = 128;         /* < 64 entries wait, a stretch adds at most 64 */
template <typename T, int NF>
struct DeferRing {
    uint32_t pix[kDeferSlots];
    int32_t i0[kDeferSlots];             /* index of the stretch's first update in the sample's own count */
    T f[NF][kDeferSlots];                /* X, Yd (, cx, cyd) at the stretch's start */
};

template <typename T>
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o < v ? o : v;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

/* Lanes are refilled with survivor records of the tile pass (A.in) and run the remaining iterations [A.i0, max_iter).
 * (A fresh-pixel form of this kernel -- lanes refilled straight from the pixel index, no tile pass -- was the first lane
 * pool; it lost to tile pass + pool on every workload and left the product in round 3: DESIGN.md section 7.) */
template <typename T, int FRACTAL, bool PERIOD = false>
__global__ void __launch_bounds__(kBlockThreads)
/* VGPRs are handed out in granules of 16 on gfx950 (measured in round 3: a 68-VGPR fp64 instantiation ran 6 workgroups per
 * CU, not the 7 its count suggests).  Round 3 held the plain fp64 instantiation at 64 (8 waves per SIMD; +-0 to -2 %); with
 * the deferred-escape ring (22 KB of LDS per workgroup: 7 at most) it takes 70 and runs 6 -- budgets of 5, 6 and "8" measured
 * the same (profiles/r04_pool_deferred_escapes.txt). */
pool_kernel(const LaunchArgs A)
{
    constexpr int NF = RecFields<FRACTAL>::n;
    constexpr size_t kBlockBytes = RingWriter<T, NF>::kBlockBytes;
    constexpr bool STRIPES = FRACTAL == 3;                /* Mandelbrot + stripe shading (shade_stripes): a finished lane keeps the
                                                           * z of its last update, and no lane runs past its deadline */
    constexpr int FR = STRIPES ? 0 : FRACTAL;
    static_assert(!(STRIPES && PERIOD), "a closed cycle has no z after max_iter updates");

    __shared__ LdsBlock S;
    stage_constants(S, A);
    stage_interior<T, FR>(S, A);
    __shared__ double2 log2_lds[sizeof(T) == 8 ? kLog2Entries : 1];
    const LogTab<T> lg = stage_log2<T>(log2_lds, A);
    __shared__ DeferRing<T, NF> defer_rings[kWavesPerBlock];
    DeferRing<T, NF>& D = defer_rings[threadIdx.x >> 6];
    uint32_t dhead = 0, dtail = 0;       /* wave-uniform entry counters */
    bool finishing = false;              /* every lane retired and the queue dry: only the ring is left */

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const int max_iter = A.max_iter;                     /* wave-uniform: straight from the kernel arguments */
    const T bailout = (T)S.bailout;
    const T B2 = bailout * bailout;
    const T B2x4 = T(4) * B2;
    const bool want_rgb = A.rgba != nullptr;
    const bool want_nu = want_rgb || A.nu != nullptr;
    const bool fast_ok = A.fast_ok != 0;
    const uint32_t refill_at = A.pool_refill_at;

    /* queue parameters, stream description and output planes are re-read from the kernel arguments where they are used
     * (kargs()): held in SGPRs across the iteration loops they were what the PERIOD variant spilled (23 SGPRs) */
    LeanQueue q;
    q.init();

    uint64_t diag_t0 = 0;
    uint32_t diag_items = 0, diag_claims = 0, diag_dry = 0;
    if (A.diag) diag_t0 = __builtin_amdgcn_s_memrealtime();

    /* per-lane state */
    uint32_t pixel = kInvalidPixel;      /* kInvalidPixel: lane is free */
    uint32_t fin = 0;                    /* 1: finished, waiting to be shaded and stored (a VGPR flag, not a
                                          * lane mask: a divergent bool carried through the iteration loops costs
                                          * mask-merging scalar instructions in every iteration) */
    Orbit<T> o;
    o.X = o.Yd = o.cx = o.cyd = o.x2 = o.y2d = T(0);
    uint32_t deadline = 0;
    int esc_i = 0;
    T esc_r2 = T(0);
    T esc_X = T(0), esc_Yd = T(0);       /* STRIPES: z after the lane's last update (escape, or its max_iter-th) */
    /* PERIOD: the lane's own state at the last snapshot (NaN: none since its refill) and "it came back to it" */
    T refX = __builtin_nan(""), refYd = __builtin_nan("");
    uint32_t cyc = 0;
    uint32_t next_snap = 0;              /* wave-uniform: clock of the next snapshot */
    uint32_t snap_window = A.period_window, snap_closed = 1u;   /* current window; lanes closed since the last snapshot */
    const uint32_t snap_cap = A.period_window > (((uint32_t)A.max_iter >> 7) << 4) ? A.period_window : (((uint32_t)A.max_iter >> 7) << 4);
    /* PERIOD, adaptive stride (see close_cycles): the wave's unchecked stretches are `stride` updates long once a cycle has
     * told it what stride to compare at (0: blocks of 16); clock of the last snapshot; smallest offset a lane closed at
     * since then */
    uint32_t stride = 0, snap_time = 0, min_hit = 0xFFFFFFFFu;
    bool learning = false;               /* a cycle was seen somewhere inside a stretch of several blocks: one block per
                                          * stretch until the next one tells the offset exactly */
    /* wave-uniform state */
    uint32_t wclock = 0;
    uint32_t next_deadline = 0;          /* a lower bound of the earliest deadline among running lanes */
    bool have_running = false;
    uint32_t res_begin = 0, res_count = 0, res_shard = 0, res_next = 0;    /* reserve: run of 64-item groups, cursor in items */
    uint32_t life_avg = 0;                    /* PERIOD: smoothed lifetime (updates since its refill) of retired lanes */
    /* PERIOD: what looking for cycles costs a wave that finds none (see `look`): stretches the wave stays ALERT (compares
     * Re z and Im z, block by block) after it last saw a lane back at its snapshot; "this wave has closed a cycle" */
    uint32_t alert = 0, n_closed = 0;
    bool ever_closed = false;
    bool dry = false, fast = fast_ok;    /* a dirty stretch costs a ring entry per escaped lane: no reason to start tested */

    FR_STAMP_DECL
    for (;;) {
        /* ---- deferred escapes: 64 queued (or the wave is leaving) -> one tested replay at full occupancy ---- */
        while (dtail - dhead >= 64u || (finishing && dtail != dhead)) {
            const uint32_t count = dtail - dhead < 64u ? dtail - dhead : 64u;
            const bool have = lane < count;
            const uint32_t slot = (dhead + lane) & (uint32_t)(kDeferSlots - 1);
            Orbit<T> t;
            t.X = have ? D.f[0][slot] : T(0);
            t.Yd = have ? D.f[1][slot] : T(0);
            if constexpr (Form<FRACTAL>::per_sample_c) { t.cx = have ? D.f[2][slot] : T(0); t.cyd = have ? D.f[3][slot] : T(0); }
            else { t.cx = have ? (T)S.julia_cx : T(0); t.cyd = have ? T(2) * (T)S.julia_cy : T(0); }
            t.x2 = t.X * t.X;
            t.y2d = t.Yd * t.Yd;
            const uint32_t rpix = D.pix[slot];
            const int ri0 = D.i0[slot];
            bool open = have;
            uint32_t ek = 0u;
            T er = T(0), eX = T(0), eYd = T(0);
            uint64_t pending = __builtin_amdgcn_ballot_w64(have);
            uint32_t k = 0;
            do {
                orbit_step<T, Form<FRACTAL>::abs_step>(t);
                const T r = orbit_r2x4(t);
                const bool e = open && r > B2x4;
                if (e) {
                    ek = k; er = r; open = false;
                    if constexpr (STRIPES) { eX = t.X; eYd = t.Yd; }
                }
                pending &= ~__builtin_amdgcn_ballot_w64(e);
                ++k;
            } while (pending != 0ull && k < (uint32_t)max_iter);
#ifdef FR_STAMP_TESTED
            st_acc[3] += k;              /* diagnostic: updates of deferred replays */
#endif
            if (have) {
                /* an escape at or past the sample's last update is no escape: it ran its max_iter updates */
                const int idx = ri0 + (int)ek;
                const bool esc = !open && idx < max_iter;
                const int r_it = esc ? idx : max_iter;
                const T r_r2 = esc ? T(0.25) * er : T(0);
                T nu;
                float rgb[3];
                /* (STRIPES: a stretch never crosses a deadline, so a deferred lane did escape before its max_iter-th update) */
                if constexpr (STRIPES) shade_stripes<T>(*kargs(), S, r_it, eX, T(0.5) * eYd, nu, rgb);
                else shade<T, FR>(*kargs(), S, lg, r_it, r_r2, want_nu, want_rgb, nu, rgb);
                KArgs K = kargs();
                if (want_rgb && (K->flags & FR_FLAG_POST_CHAIN))
                    post_chain(rgb, S.brightness, S.saturation, S.contrast, FR != 0);
                if (K->rgba) K->rgba[rpix] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
                if (K->nu) reinterpret_cast<T*>(K->nu)[rpix] = nu;
                if (K->iter) K->iter[rpix] = r_it;
            }
            dhead += count;
            __builtin_amdgcn_wave_barrier();
        }
        if (finishing) break;
        /* ---- retire: shade and store the finished lanes ---- */
        FR_STAMP_BEGIN();
        const uint64_t finm = __builtin_amdgcn_ballot_w64(fin != 0u);
        if (finm != 0ull) {
            if (PERIOD && ever_closed) {
                /* lifetime of (the first of) the lanes retired now: its deadline was set to refill clock + remaining updates.
                 * (Read with the lane's number, in uniform control flow: a readfirstlane inside the divergent block below
                 * left the wave running on with the retiring lanes' EXEC mask -- the rest of the loop saw 17 lanes.) */
                const uint32_t life = (uint32_t)__builtin_amdgcn_readlane((int)(wclock - (deadline - ((uint32_t)max_iter - (uint32_t)A.i0))),
                                                                          (int)__builtin_ctzll(finm));
                life_avg = life_avg == 0u ? life : (3u * life_avg + life) >> 2;
            }
            if (fin != 0u) {
                T nu;
                float rgb[3];
                if constexpr (STRIPES) shade_stripes<T>(*kargs(), S, esc_i, esc_X, T(0.5) * esc_Yd, nu, rgb);
                else shade<T, FR>(*kargs(), S, lg, esc_i, esc_r2, want_nu, want_rgb, nu, rgb);
                KArgs K = kargs();
                if (want_rgb && (K->flags & FR_FLAG_POST_CHAIN))
                    post_chain(rgb, S.brightness, S.saturation, S.contrast, FR != 0);
                if (K->rgba) K->rgba[pixel] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
                if (K->nu) reinterpret_cast<T*>(K->nu)[pixel] = nu;
                if (K->iter) K->iter[pixel] = esc_i;
                pixel = kInvalidPixel;
                fin = 0u;
            }
        }
        FR_STAMP_END(3);
        /* ---- refill the free lanes from the reserve ---- */
        FR_STAMP_BEGIN();
        for (;;) {
            const uint64_t freem = __builtin_amdgcn_ballot_w64(pixel == kInvalidPixel);
            if (freem == 0ull || dry) break;
            if (res_next == res_count * 64u) {
#ifdef FR_STAMP
                const uint64_t stq = __builtin_amdgcn_s_memtime();
                const bool got_q = q.next<true>(lane, res_begin, res_count, res_shard);
#ifndef FR_STAMP_TESTED
                st_acc[0] += __builtin_amdgcn_s_memtime() - stq;
#endif
                if (!got_q) {
#else
                if (!q.next<true>(lane, res_begin, res_count, res_shard)) {
#endif
                    dry = true;
                    /* diagnostics: when this wave found the queue dry, in 100 MHz ticks since its start (bits 32..) */
                    diag_dry = A.diag ? (uint32_t)(__builtin_amdgcn_s_memrealtime() - diag_t0) : 0u;
                    break;
                }
                res_next = 0;
                ++diag_claims;
                diag_items += res_count;
            }
            const uint32_t nfree = (uint32_t)__builtin_popcountll(freem);
            const uint32_t avail = res_count * 64u - res_next;
            const uint32_t n = nfree < avail ? nfree : avail;
            if (pixel == kInvalidPixel) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(freem >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)freem, 0u));
                if (rank < n) {
                    const uint32_t t = res_next + rank;
                    const uint32_t j = res_begin + (t >> 6), l = t & 63u;
                    {
                        /* record l of block j of region res_shard */
                        KArgs K = kargs();
                        const uint8_t* b = K->in.base + ((size_t)res_shard * K->in.region_blocks + j) * kBlockBytes;
                        const T* fields = reinterpret_cast<const T*>(b + RingWriter<T, NF>::kHeaderBytes);
                        const uint32_t pix = reinterpret_cast<const uint32_t*>(b)[l];
                        if (pix != kInvalidPixel) {
                            pixel = pix;
                            const uint32_t done = reinterpret_cast<const uint32_t*>(b)[64 + l];
                            o.X = fields[l];
                            o.Yd = fields[64 + l];
                            if constexpr (Form<FRACTAL>::per_sample_c) { o.cx = fields[128 + l]; o.cyd = fields[192 + l]; }
                            else { o.cx = (T)S.julia_cx; o.cyd = T(2) * (T)S.julia_cy; }
                            o.x2 = o.X * o.X;
                            o.y2d = o.Yd * o.Yd;
                            deadline = wclock + ((uint32_t)max_iter - done);
                            if constexpr (PERIOD) { refX = __builtin_nan(""); refYd = refX; cyc = 0u; }
                        }
                    }
                }
            }
            res_next += n;
        }
#ifdef FR_STAMP
        {   /* the records must have arrived before the clock is read: touch them */
            const T touch = o.X + o.Yd + o.cx + o.cyd;
            asm volatile("" :: "v"(touch));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#endif
        FR_STAMP_END(2);
        const uint64_t active = __builtin_amdgcn_ballot_w64(pixel != kInvalidPixel);
        if (active == 0ull) { finishing = true; continue; }  /* queue dry and every lane retired: flush the ring, leave */
        const uint32_t nactive = (uint32_t)__builtin_popcountll(active);
        /* a wave that can no longer refill is on the critical path of the launch: give it issue priority */
        if (dry) __builtin_amdgcn_s_setprio(3);
        /* a record has run AT MOST A.i0 updates (exactly that many unless its trip of the tile pass left early), so no
         * refilled lane's deadline lies before wclock + max_iter - i0, and that is the latest deadline among the lanes that
         * ran their A.i0: the earliest one only changes when it is reached -- no reduction here.  (Lanes with fewer updates
         * behind them have later deadlines: where only such lanes were running, the new ones may come first -- the clock stops
         * at the bound, finds nobody due and looks for the earliest deadline then.) */
        {
            const uint32_t bound = wclock + (uint32_t)(max_iter - A.i0);
            if (!have_running || (int32_t)(bound - next_deadline) < 0) next_deadline = bound;
            have_running = true;
        }

        /* ---- iterate until `goal` lanes have finished ---- */
        /* How many idle lanes to wait for.  A retire + refill costs the wave about 40 updates' worth of instructions
         * whatever the number of lanes it serves; waiting for k lanes that finish tau updates apart idles k^2 tau / 2
         * lane-updates.  Per lane served that is 40 / k + k tau / 128, smallest at k = 72 / sqrt(tau): where lanes finish
         * in quick succession (escape-dense records, or 64 interior lanes reaching one deadline) the configured
         * threshold (24) is right, where they trickle out -- cycle closing on a deep view: one lane per ~150 updates --
         * waiting for 24 keeps a fifth of the wave idle.  tau is taken from how long the lanes retired lately had lived
         * (64 lanes of lifetime L finish L / 64 apart). */
        uint32_t want = refill_at;
        if (PERIOD && ever_closed) {       /* (a wave that has closed nothing retires lanes by escape and deadline only) */
            /* tau = (smoothed lifetime of the lanes retired lately) / 64 */
            const uint32_t tau = life_avg >> 6;
            want = tau >= 288u ? 4u : (tau >= 128u ? 6u : (tau >= 72u ? 8u : (tau >= 32u ? 12u : (tau >= 14u ? 18u : refill_at))));
            if (want > refill_at) want = refill_at;
        }
        const uint32_t goal = (dry || want > nactive) ? nactive : want;             /* queue dry: run the rest out */
        uint32_t newly = 0;
        /* lanes whose deadline is reached are interior; then find the next earliest deadline */
        auto reach_deadline = [&](bool at_or_past) {
            const bool running = pixel != kInvalidPixel && fin == 0u;
            const bool hit = running && (at_or_past ? (int32_t)(wclock - deadline) >= 0 : deadline == wclock);
            if (hit) {
                esc_i = max_iter; esc_r2 = T(0); fin = 1u;
                if constexpr (STRIPES) { esc_X = o.X; esc_Yd = o.Yd; }          /* z after exactly max_iter updates */
                o.X = T(0); o.Yd = T(0); o.cx = T(0); o.cyd = T(0); o.x2 = T(0); o.y2d = T(0);
            }
            const uint32_t nhit = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit));
            newly += nhit;
            const uint32_t rel = wave_min_u32<T>((running && !hit) ? deadline - wclock : 0xFFFFFFFFu);
            have_running = rel != 0xFFFFFFFFu;
            next_deadline = wclock + (have_running ? rel : (uint32_t)max_iter);
        };
        /* PERIOD.  A running lane whose state equals -- as numbers -- its own state at the last snapshot is on a
         * cycle: the update is a deterministic function of (z, c), so it repeats for ever and the lane can never
         * escape.  It is interior with exactly the result the remaining iterations would give (index max_iter;
         * the plain colourings do not look at the final z).  Called where every running lane is known not to
         * have escaped; also takes the next snapshot when its time has come.
         *
         * WHERE to compare.  Comparing after every unchecked block sees a cycle of period p only at offsets (from the
         * snapshot) that are multiples of lcm(16, p).  Short cycles (the period-1/2/3 components of shallow views) show
         * within a block or three; the cycles of a deep view do not: in the C4 view the floating-point orbits settle on
         * cycles of 78, 39, 156, ... updates (tools/cycle_study.py), lcm(16, 78) = 624, and with the window that needs
         * plus the wait for the next snapshot a lane ran ~1 100 updates too long, every lane, every refill.  So the
         * wave LEARNS its stride: `hit_off` is the offset at which a lane was first seen back at its snapshot (a
         * multiple of its period, and in block mode of 16); from 256 up the wave switches to stretches of hit_off / 16
         * updates with one comparison each -- every period whose 16-fold divides hit_off shows within hit_off, its
         * first multiple of the stride typically a period or two away (78 at stride 39) -- and shrinks the window to
         * twice the smallest offset anything closed at.  A window that closes nothing doubles (both modes); at the cap
         * the stride is forgotten and learned again.  Any equality is a proof, whatever the stride: only WHEN a cycle
         * is seen changes, never a pixel. */
        auto close_cycles = [&](const uint32_t hit_off) {
            const bool running = pixel != kInvalidPixel && fin == 0u;
            const bool hit = running && cyc != 0u;
            if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
                if (hit) {
                    esc_i = max_iter; esc_r2 = T(0); fin = 1u; cyc = 0u;
                    o.X = T(0); o.Yd = T(0); o.cx = T(0); o.cyd = T(0); o.x2 = T(0); o.y2d = T(0);
                }
                const uint32_t nhit = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit));
                newly += nhit;
                snap_closed += nhit;
                n_closed += nhit;
                ever_closed = true;
                /* seen, but not at which block of the stretch: worth finding out only if the offset may be a long one
                 * (a short period closes within a few blocks whatever the stride) */
                if (hit_off == 0u) learning = stride == 0u && wclock - snap_time >= 256u + 4u * (uint32_t)kFastBlock;
                else {
                    learning = false;
                    if (hit_off < min_hit) min_hit = hit_off;
                    if (stride == 0u && hit_off >= 256u && hit_off <= (snap_cap >> 1) + 16u * (uint32_t)kFastBlock) {
                        /* learn.  Offsets counted in whole blocks are multiples of 16 AND of the period: a sixteenth is a
                         * stride whose multiples meet the period's within hit_off.  An offset that is not (tested
                         * stretches since the snapshot) is still a multiple of the period: compare every hit_off. */
                        stride = (hit_off & 15u) ? hit_off : hit_off >> 4;
                        snap_window = hit_off;
                        next_snap = wclock;                          /* re-align: stretches are counted from a fresh snapshot */
                    }
                }
            }
            if ((int32_t)(wclock - next_snap) >= 0) {
                /* a slot that is not running gets a NaN snapshot: parked at z = 0 it would equal its own snapshot for ever,
                 * and the cheap first look (Re z alone, see `look`) would fire in every block */
                refX = running ? o.X : (T)__builtin_nanf(""); refYd = o.Yd;
                if (snap_closed == 0u) {
                    /* a window that closed nothing is doubled (up to max_iter / 8); at the cap a learned stride is dropped */
                    if (snap_window < snap_cap) snap_window <<= 1;
                    else if (stride != 0u) { stride = 0u; snap_window = A.period_window; }
                } else if (stride != 0u && min_hit != 0xFFFFFFFFu) {
                    const uint32_t w2 = min_hit << 1;
                    snap_window = w2 < stride ? stride : (w2 > snap_cap ? snap_cap : w2);
                }
                snap_closed = 0u;
                min_hit = 0xFFFFFFFFu;
                snap_time = wclock;
                next_snap = wclock + snap_window;
            }
        };
        /* WHAT looking costs where nothing closes (the C3 Julia dust, the C5 view: +7 % / +4.4 % in round 2; measured piece
         * by piece in round 3, profiles/r03_periodicity_cost.txt: the per-block comparison 5 %, the lifetime bookkeeping of
         * the refill threshold 1-3 %, everything else 2.5 %).  It is not the two compares, it is what hangs on them: a branch
         * (or a select) on a mask the vector unit has only just produced, in every block.  So a wave is QUIET until it has a
         * reason not to be: per block ONE v_cmp of Re z against the snapshot whose mask is ORed into a scalar pair -- nothing
         * waits for it -- and one scalar test of that pair per stretch.  Re z back at its snapshot is the reason (for a
         * sample not on a cycle it does not happen): the wave turns ALERT for the next 64 stretches -- every block compares
         * Re z and Im z and the hit is taken at its block, as before -- and every lane seen renews that.  A quiet wave
         * therefore sees a cycle one period (one lcm(16, p) for the long cycles of deep views) later than an alert one
         * would, once; tested stretches do not look at all (an escape-dense neighbourhood: the wave returns to unchecked
         * blocks as soon as 16 updates pass without an escape).  No sampling, no skipped stretches: a comparison that is
         * skipped with a fixed phase misses a cycle of period 78 for ever (tried; the C4 view lost all its closures). */
        bool saw = false;              /* some lane of the current stretch was seen back at its snapshot */
        uint64_t hint = 0ull;          /* quiet wave: lanes whose Re z equalled the snapshot's at some block of the stretch */
        auto look = [&](uint32_t& seen) {          /* alert */
            if (__builtin_amdgcn_ballot_w64(o.X == refX) != 0ull) {
                cold_path();
                seen |= (o.X == refX && o.Yd == refYd) ? 1u : 0u;
                saw = true;
            }
        };
        uint32_t clean = 0;            /* tested updates since the last escape */
        uint32_t streak = 0;           /* clean unchecked blocks in a row */
        /* Every wave must reach its exit whatever happens to the bookkeeping above: a running lane finishes within
         * max_iter updates, so a stretch loop that has run max_iter + 4096 updates without reaching its goal is a bug.
         * It then says so (the host fails the render: FR_ERR_INTERNAL) and retires what it holds as interior. */
        const uint32_t watchdog = wclock + (uint32_t)max_iter + 4096u;
        while (newly < goal) {
            if ((int32_t)(wclock - watchdog) > 0) {
                if (lane == 0 && A.out.overflow) __hip_atomic_store(A.out.overflow, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef FR_WATCHDOG_DUMP
                if (A.diag) {
                    uint64_t first = 1;
                    if (lane == 0) first = atomicAdd((unsigned long long*)&A.diag[15], 1ull);
                    first = __builtin_amdgcn_readfirstlane((int)first);
                    if (first == 0) {
                        if (lane == 0) {
                            A.diag[0] = wclock; A.diag[1] = next_deadline; A.diag[2] = have_running; A.diag[3] = goal; A.diag[4] = newly;
                            A.diag[5] = nactive; A.diag[6] = stride; A.diag[7] = fast; A.diag[8] = watchdog; A.diag[9] = snap_window;
                            A.diag[10] = next_snap; A.diag[11] = dry; A.diag[12] = life_avg; A.diag[13] = refill_at; A.diag[14] = (uint64_t)max_iter | ((uint64_t)A.i0 << 32);
                        }
                        A.diag[16 + lane] = ((uint64_t)pixel << 32) | (uint64_t)(uint32_t)(deadline - wclock);
                        A.diag[80 + lane] = ((uint64_t)fin << 32) | (uint64_t)cyc;
                    }
                }
#endif
                if (pixel != kInvalidPixel && fin == 0u) {
                    esc_i = max_iter; esc_r2 = T(0); fin = 1u;
                    o.X = T(0); o.Yd = T(0); o.cx = T(0); o.cyd = T(0); o.x2 = T(0); o.y2d = T(0);
                }
                break;
            }
            if (fast) {
                if constexpr (STRIPES) {
                    /* where the z of a sample that never escapes is read (striped interior, style 0) no stretch may cross a
                     * deadline: that z must be the one after exactly max_iter updates.  Within the longest stretch (64 updates)
                     * of the earliest deadline the wave runs tested blocks, which stop at it */
                    if (A.stripe_enabled && A.interior_style == 0 &&
                        (uint32_t)(next_deadline - wclock) < 4u * (uint32_t)kFastBlock) goto tested_stretch;
                }
                const T sX = o.X, sYd = o.Yd;
                /* after 2 (6) clean stretches in a row the wave runs 2 (4) blocks per snapshot / test (a half, a
                 * quarter of that overhead on the long interior runs that dominate deep views); a dirty one resets it */
                uint32_t reps = streak >= 6u ? 4u : (streak >= 2u ? 2u : 1u);
                if constexpr (PERIOD) { if (learning) reps = 1u; }
                uint32_t len = reps * (uint32_t)kFastBlock;          /* updates of this stretch */
                uint32_t seen = 0u;                                  /* PERIOD: lanes seen back at their snapshot */
                saw = false;
                hint = 0ull;
                constexpr bool looking = PERIOD;
                bool strided = false, exact = reps == 1u;            /* exact: a lane seen back was seen at the stretch's end */
                if constexpr (PERIOD) strided = stride != 0u;
                if (strided) {
                    /* learned stride: a whole number of strides, at least 64 updates, ONE comparison */
                    len = stride * (stride >= 64u ? 1u : (63u + stride) / stride);
                    exact = true;
                    for (uint32_t b = len >> 4; b != 0u; --b) {
#pragma unroll
                        for (int k = 0; k < kFastBlock; ++k) orbit_step<T, Form<FRACTAL>::abs_step>(o);
                    }
                    for (uint32_t r = len & 15u; r != 0u; --r) orbit_step<T, Form<FRACTAL>::abs_step>(o);
                    if (looking) look(seen);
                } else if (looking && alert != 0u) {
                    for (uint32_t rep = 0; rep < reps; ++rep) {
#pragma unroll
                        for (int k = 0; k < kFastBlock; ++k) orbit_step<T, Form<FRACTAL>::abs_step>(o);
                        look(seen);
                    }
                } else {
                    /* quiet (or no cycle closing at all): no branch inside the stretch */
                    for (uint32_t rep = 0; rep < reps; ++rep) {
#pragma unroll
                        for (int k = 0; k < kFastBlock; ++k) orbit_step<T, Form<FRACTAL>::abs_step>(o);
                        if (looking) hint |= __builtin_amdgcn_ballot_w64(o.X == refX);
                    }
                }
                const bool bad = !(orbit_r2x4(o) <= B2x4);
                const uint64_t badm = __builtin_amdgcn_ballot_w64(bad);
                bool ring_full = false;
                if (badm != 0ull) {
                    /* dirty stretch: the escaped lanes' stretch-start states go to the ring (located 64 at a time, see
                     * DeferRing), the lanes are free; everybody else's progress counts and the wave stays unchecked */
                    if (bad) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(badm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)badm, 0u));
                        const uint32_t slot = (dtail + rank) & (uint32_t)(kDeferSlots - 1);
                        D.pix[slot] = pixel;
                        D.i0[slot] = (int)(wclock - (deadline - (uint32_t)max_iter));
                        D.f[0][slot] = sX;
                        D.f[1][slot] = sYd;
                        if constexpr (Form<FRACTAL>::per_sample_c) { D.f[2][slot] = o.cx; D.f[3][slot] = o.cyd; }
                        pixel = kInvalidPixel;
                        o.X = T(0); o.Yd = T(0); o.cx = T(0); o.cyd = T(0); o.x2 = T(0); o.y2d = T(0);
                    }
                    const uint32_t nbad = (uint32_t)__builtin_popcountll(badm);
                    dtail += nbad;
                    newly += nbad;
                    streak = 0;
                    ring_full = dtail - dhead >= 64u;
                    __builtin_amdgcn_wave_barrier();
#ifdef FR_STAMP_TESTED
                    st_acc[2] += len;        /* diagnostic: updates of dirty unchecked stretches */
#endif
                } else {
                    ++streak;
                }
                wclock += len;
                /* lanes at or past their deadline that are still running never escaped -> interior */
                if ((int32_t)(wclock - next_deadline) >= 0) reach_deadline(true);
                /* (a lane that escaped inside the stretch is finished: close_cycles only looks at running lanes) */
                if constexpr (PERIOD) {
                    alert = (saw || hint != 0ull) ? 64u : (alert != 0u ? alert - 1u : 0u);
                    /* nothing seen and no snapshot due (the usual case where nothing closes): two scalar tests */
                    if (saw || (int32_t)(wclock - next_snap) >= 0) { cyc |= seen; close_cycles(exact ? wclock - snap_time : 0u); }
                }
                if (ring_full) break;        /* 64 deferred escapes queued: replay them (top of the loop) before more arrive */
                continue;
            }
        tested_stretch:
            /* tested stretch: up to the next deadline, at most one block.  The loop carries a countdown and
             * one vector-compare branch; goal and deadline are only looked at where they can change (on an
             * escape event / after the stretch).  On escape-dense views such as the C3 Julia dust the scalar
             * bookkeeping of per-iteration tests made the pass SALU-bound (147 M scalar against 98 M vector
             * instructions per frame, one scalar issue port per CU). */
            uint32_t n = next_deadline - wclock;                 /* >= 1: deadlines lie ahead of the clock */
            if (n > (uint32_t)kFastBlock) n = (uint32_t)kFastBlock;
            uint32_t k = 0;
            bool escaped = false;
            while (k < n) {
                orbit_step<T, Form<FRACTAL>::abs_step>(o);
                const T r2x4 = orbit_r2x4(o);
                const bool e = r2x4 > B2x4;
                const uint64_t em = __builtin_amdgcn_ballot_w64(e);
                ++k;
                if (em != 0ull) {
                    if (e) {
                        esc_i = (int)(wclock + k - 1u - (deadline - (uint32_t)max_iter));
                        esc_r2 = T(0.25) * r2x4;
                        fin = 1u;
                        if constexpr (STRIPES) { esc_X = o.X; esc_Yd = o.Yd; }
                        o.X = T(0); o.Yd = T(0); o.cx = T(0); o.cyd = T(0); o.x2 = T(0); o.y2d = T(0);
                    }
                    newly += (uint32_t)__builtin_popcountll(em);
                    escaped = true;
                    if (newly >= goal) n = k;        /* single-exit loop: goal reached -> this was the last update */
                }
            }
            wclock += k;
#ifdef FR_STAMP_TESTED
            st_acc[0] += k;              /* diagnostic: updates this wave ran TESTED */
#endif
            clean = escaped ? 0u : clean + k;
            if (wclock == next_deadline) reach_deadline(false);
            /* back to unchecked blocks after a block's worth of updates without an escape */
            if (clean >= (uint32_t)kFastBlock) { fast = fast_ok; clean = 0; }
        }
#ifdef FR_STAMP
        st_acc[1] = wclock;          /* pool: updates this wave has run (x 64 lanes = the lane-updates it paid for) */
#endif
    }
    if constexpr (PERIOD) {
        if (lane == 0 && A.closed_flag) {
            uint32_t* w = A.closed_flag + (blockIdx.x & (uint32_t)(kFeedbackShards - 1)) * kShardStrideWords;
            if (n_closed != 0u) atomicAdd(w, n_closed);
            atomicAdd(w + 1, diag_items * 64u);                 /* records taken in (whole blocks claimed) */
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && A.closed_flag)        /* "this render's pool looked" */
            __hip_atomic_store(A.closed_flag + kFeedbackShards * kShardStrideWords, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (A.diag && lane == 0) {      /* as diag_write, with the dry time packed above the dequeue count */
        const uint32_t wave_id = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
        uint64_t* d = A.diag + (size_t)wave_id * kDiagWords;
        d[0] = diag_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = diag_items;
        d[3] = (uint64_t)diag_claims | ((uint64_t)diag_dry << 32);
    }
    FR_STAMP_WRITE(A, lane);
}

/* ---- Deep_Zoom: the reference's perturbation shader ------------------------------------------------
 * shaders/test_deep_zoom.comp restated operation for operation (fp32, float-float centre/zoom, explicit
 * fma in dd_mul_sf as the shader writes it), quirks included -- see DESIGN.md.  The reference orbit
 * (fp64 on the host, narrowed to float pairs as src/deep_zoom_system.cpp:102-110 uploads it) is read
 * with a wave-uniform index, i.e. through the scalar cache.  Sub-tiles from the same sharded queue. */
struct DeepZoomArgs {
    float cx_hi, cx_lo, cy_hi, cy_lo, zoom_hi, zoom_lo;
    float bailout, color_offset, color_scale;
    int32_t palette_mode, max_iter, ref_iter;
    int32_t W, H, rows_local, part, nparts, rows_per_strip, out_frame;
    const float2* orbit;
    float4* rgba;
    float* nu;
    int32_t* iter;
    QueueArgs q;
};

struct FF { float hi, lo; };
__device__ __forceinline__ FF dd_add_dd(FF a, FF b)                /* :31-39 */
{
    const float s = a.hi + b.hi;
    const float v = s - a.hi;
    const float t = ((b.hi - v) + (a.hi - (s - v))) + (a.lo + b.lo);
    FF r; r.hi = s + t; r.lo = t - (r.hi - s); return r;
}
__device__ __forceinline__ FF dd_mul_sf(FF a, float b)             /* :41-48 */
{
    const float p = a.hi * b;
    const float e = __builtin_fmaf(a.hi, b, -p);
    const float lo = __builtin_fmaf(a.lo, b, e);
    FF r; r.hi = p + lo; r.lo = lo - (r.hi - p); return r;
}
__device__ __forceinline__ float fractf(float x) { return x - floorf(x); }

__device__ __forceinline__ void deep_zoom_color(const DeepZoomArgs& A, float iter, float zx, float zy,
                                                float rgb[3], float& smooth)            /* get_color, :75-103 */
{
    float lenz = sqrtf(zx * zx + zy * zy);
    lenz = fmaxf(lenz, 1e-12f);
    const float log_zn = logf(lenz);
    const float nu = logf(log_zn / logf(2.0f)) / logf(2.0f);
    smooth = iter + 1.0f - nu;
    const float t = smooth * A.color_scale + A.color_offset;
    if (A.palette_mode == 0) {                                      /* hsv2rgb(fract(t*0.05), 0.8, 0.9), :66-70 */
        const float h = fractf(t * 0.05f), s = 0.8f, v = 0.9f;
        const float K[4] = {1.0f, 2.0f / 3.0f, 1.0f / 3.0f, 3.0f};
        for (int c = 0; c < 3; ++c) {
            const float p = fabsf(fractf(h + K[c]) * 6.0f - K[3]);
            const float q = clamp01(p - K[0]);
            rgb[c] = v * (K[0] * (1.0f - s) + q * s);
        }
    } else if (A.palette_mode == 1) {
        const float s = fractf(t * 0.03f);
        rgb[0] = 0.0f * (1.0f - s) + 1.0f * s; rgb[1] = 0.1f * (1.0f - s) + 1.0f * s; rgb[2] = 0.3f * (1.0f - s) + 1.0f * s;
    } else if (A.palette_mode == 2) {
        const float s = fractf(t * 0.04f);
        rgb[0] = 0.1f * (1.0f - s) + 1.0f * s; rgb[1] = 0.0f * (1.0f - s) + 0.8f * s; rgb[2] = 0.0f * (1.0f - s) + 0.0f * s;
    } else {
        const float s = fractf(t * 0.02f);
        rgb[0] = rgb[1] = rgb[2] = s;
    }
}

template <int FPW_LOG2>
__global__ void __launch_bounds__(kBlockThreads)
deep_zoom_kernel(const DeepZoomArgs A)
{
    constexpr int FPW = 1 << FPW_LOG2;
    constexpr int FPH = kWave / FPW;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const int lx = (int)(lane & (FPW - 1)), ly = (int)(lane >> FPW_LOG2);
    const int W = A.W, H = A.H, max_iter = A.max_iter, ref_iter = A.ref_iter;
    const float bailout = fmaxf(2.0f, A.bailout);                   /* :114 */
    const float bailout_sq = bailout * bailout;
    const FF center_x = {A.cx_hi, A.cx_lo}, center_y = {A.cy_hi, A.cy_lo}, zoom = {A.zoom_hi, A.zoom_lo};
    const float aspect = (float)W / (float)H;                       /* :125 */
    const FF pixel_size = dd_mul_sf(zoom, 4.0f / (float)H);          /* :128 */
    const int n_ref = max_iter < ref_iter ? max_iter : ref_iter;

    WaveQueue q;
    q.init(A.q.heads, A.q.n_blk, (uint32_t)kShardBlock, A.q.run_shift, A.q.run_min, A.q.run_max, lane, A.q.ns_log2);
    uint32_t begin, count, cur_shard;
    while (q.next(begin, count, cur_shard)) {
        for (uint32_t j = begin; j < begin + count; ++j) {
            const uint32_t blk = WaveQueue::block_of(j / kShardBlock, cur_shard, A.q.ns_log2);
            if (blk >= A.q.n_blk) continue;
            const uint32_t sid = blk * kShardBlock + (j % kShardBlock);
            if (sid >= A.q.n_items) continue;
            const uint32_t sty = sid / A.q.nsx, stx = sid - sty * A.q.nsx;
            const int px = (int)stx * FPW + lx;
            const int lrow = (int)sty * FPH + ly;
            const bool inside = px < W && lrow < A.rows_local;
            int py = lrow;
            if (A.nparts != 1) {
                const int strip = lrow / A.rows_per_strip;
                py = (strip * A.nparts + A.part) * A.rows_per_strip + (lrow - strip * A.rows_per_strip);
            }
            const float uvx = (float)px / (float)W, uvy = (float)py / (float)H;          /* :118 */
            const float offset_x = (uvx - 0.5f) * aspect;                               /* :131-132 */
            const float offset_y = (uvy - 0.5f);
            const FF dc_x = dd_mul_sf(pixel_size, offset_x), dc_y = dd_mul_sf(pixel_size, offset_y);   /* :135-136 */
            const FF c_x_dd = dd_add_dd(center_x, dc_x), c_y_dd = dd_add_dd(center_y, dc_y);          /* :139-140 */
            const float delta_x = dc_x.hi + dc_x.lo, delta_y = dc_y.hi + dc_y.lo;                      /* :143 */
            const float c_fx = c_x_dd.hi + c_x_dd.lo, c_fy = c_y_dd.hi + c_y_dd.lo;

            float dzx = 0.0f, dzy = 0.0f;
            bool live = inside;
            int esc_i = max_iter;
            float ezx = 0.0f, ezy = 0.0f;
            /* perturbed iteration against the reference orbit, :153-173 */
            auto perturb = [&](const float2 zr, const int i) {
                const float mx = zr.x * dzx - zr.y * dzy, my = zr.x * dzy + zr.y * dzx;
                const float t1x = mx * 2.0f, t1y = my * 2.0f;
                const float t2x = dzx * dzx - dzy * dzy, t2y = 2.0f * dzx * dzy;
                const float ndx = t1x + t2x + delta_x, ndy = t1y + t2y + delta_y;
                if (live) {
                    dzx = ndx; dzy = ndy;
                    const float zfx = zr.x + dzx, zfy = zr.y + dzy;
                    if (zfx * zfx + zfy * zfy > bailout_sq) { live = false; esc_i = i; ezx = zfx; ezy = zfy; }
                }
            };
            /* four reference points per scalar load and per "anybody still alive" test: the orbit comes through
             * the scalar cache (wave-uniform index), whose latency would otherwise sit in every update */
            int i = 0;
            for (; i + 4 <= n_ref; i += 4) {
                if (__builtin_amdgcn_ballot_w64(live) == 0ull) break;
                const float2 z0 = A.orbit[i], z1 = A.orbit[i + 1], z2 = A.orbit[i + 2], z3 = A.orbit[i + 3];
                perturb(z0, i); perturb(z1, i + 1); perturb(z2, i + 2); perturb(z3, i + 3);
            }
            if (__builtin_amdgcn_ballot_w64(live) != 0ull)
                for (; i < n_ref; ++i) perturb(A.orbit[i], i);
            /* continue in plain fp32 for the remaining iterations, :181-203 */
            float zx, zy;
            if (ref_iter > 0) { const float2 zl = A.orbit[ref_iter - 1]; zx = zl.x + dzx; zy = zl.y + dzy; }
            else { zx = c_fx; zy = c_fy; }
            auto plain = [&](const int k) {
                const float z2x = zx * zx - zy * zy, z2y = 2.0f * zx * zy;
                const float nx = z2x + c_fx, ny = z2y + c_fy;
                if (live) {
                    zx = nx; zy = ny;
                    if (zx * zx + zy * zy > bailout_sq) { live = false; esc_i = k; ezx = zx; ezy = zy; }
                }
            };
            int k = n_ref;
            for (; k + 4 <= max_iter; k += 4) {
                if (__builtin_amdgcn_ballot_w64(live) == 0ull) break;
                plain(k); plain(k + 1); plain(k + 2); plain(k + 3);
            }
            if (__builtin_amdgcn_ballot_w64(live) != 0ull)
                for (; k < max_iter; ++k) plain(k);
            if (inside) {
                float rgb[3] = {0.0f, 0.0f, 0.0f};
                float smooth = (float)max_iter;
                if (esc_i < max_iter && !((float)esc_i >= (float)max_iter - 0.5f))      /* :76 */
                    deep_zoom_color(A, (float)esc_i, ezx, ezy, rgb, smooth);
                const size_t o = (size_t)(A.out_frame ? py : lrow) * (size_t)W + (size_t)px;
                if (A.rgba) A.rgba[o] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
                if (A.nu) A.nu[o] = smooth;
                if (A.iter) A.iter[o] = esc_i;
            }
        }
    }
}

/* ---- exports: RGBA f32 -> packed RGB8 / RGB16, flipped (src/vk_engine.cpp:1344-1371, :2054-2073) ---------------------
 * HBM-bound by design: 16 B read + 3 (6) B written per pixel.  A thread converts FOUR consecutive pixels of an output
 * row -- four 16-byte loads in flight, 12 (24) output bytes stored as three dwords (dwordx2s): whole 128-byte lines per
 * wave-instruction instead of single bytes at a stride of three.  Frames whose width is not a multiple of four, and output
 * pointers that are not dword (dwordx2) aligned, take the one-pixel-per-thread form (`quads` = 0).
 *
 * The 8-bit byte is EXACTLY the reference loop's (uint8)(powf(aces(v), 1/2.2f) * 255.0f): the gamma is estimated with the
 * hardware exp2(log2 a / 2.2) (within an ulp or two of powf: the estimated byte is off by one only next to a truncation
 * edge) and then corrected against the thresholds of the powf form -- thr[b] = {t[b], t[b + 1]}, t[b] the smallest float
 * whose byte is >= b (fr_export8_thresholds, host libm), 2 KB staged in LDS, one ds_read_b64 and two compares per channel.
 * aces() itself is the same IEEE operations as the host's (contraction off, correctly rounded divide). */
__device__ __forceinline__ float half_round(float f) { return __half2float(__float2half_rn(f)); }

/* aces() with the clamp as ONE v_med3_f32 instead of two compares and two selects.  It differs from clamp01() only for
 * -0.0 (kept by the ternaries, +0.0 or -0.0 here) and NaN inputs, whose byte is 0 either way. */
__device__ __forceinline__ float aces_med3(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;       /* src/vk_engine.cpp:1344-1351 */
    return __builtin_amdgcn_fmed3f((x * (a * x + b)) / (x * (c * x + d) + e), 0.0f, 1.0f);
}

template <bool HALF>
__device__ __forceinline__ uint32_t to_u8(const float2* thr, float f)
{
    if (HALF) f = half_round(f);
    f = aces_med3(f);                                                         /* :1366 */
    const float est = pow01(f, 1.0f / 2.2f);                                  /* :1367; f is in [0, 1], so is est */
    uint32_t b = (uint32_t)(est * 255.0f);                                    /* :1368 (truncation); <= 255 */
    const float2 t = thr[b & 255u];
    b -= f < t.x ? 1u : 0u;
    b += f >= t.y ? 1u : 0u;
    return b;
}
template <bool HALF>
__device__ __forceinline__ uint32_t to_u16(float f)
{
    if (HALF) f = half_round(f);
    f = __builtin_amdgcn_fmed3f(f, 0.0f, 1.0f);                               /* :2068 (NaN -> 0 either way) */
    return (uint32_t)(f * 65535.0f);                                          /* :2069 */
}

/* Walks the quads (four consecutive pixels of an output row) q = first, first + stride, ... of a W x H frame: (y, x4) kept
 * incrementally -- a 64-bit q / (W / 4) per trip was a fifth of the kernel's instructions. */
struct QuadWalk {
    uint32_t y, x, dy, dx, wq;
    __device__ __forceinline__ QuadWalk(uint32_t first, uint32_t stride, uint32_t wq_) : wq(wq_)
    {
        y = first / wq; x = first - y * wq;
        dy = stride / wq; dx = stride - dy * wq;
    }
    __device__ __forceinline__ void step() { x += dx; y += dy; if (x >= wq) { x -= wq; ++y; } }
};

template <bool HALF>
__global__ void __launch_bounds__(kBlockThreads)
export_rgb8_kernel(const float4* __restrict__ rgba, uint8_t* __restrict__ rgb8, int W, int H, int quads_ok,
                   const float2* __restrict__ thr_global)
{
    __shared__ float2 thr[256];
    thr[threadIdx.x] = thr_global[threadIdx.x];                               /* kBlockThreads == 256 entries */
    __syncthreads();
    const uint32_t n = (uint32_t)W * (uint32_t)H;                             /* < 2^31 (fr_params_validate) */
    const uint32_t stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
    if (quads_ok) {
        const uint32_t quads = n >> 2;
        uint32_t* out = reinterpret_cast<uint32_t*>(rgb8);                    /* 12 bytes per quad: 4-byte aligned */
        QuadWalk w(first, stride, (uint32_t)W >> 2);
        for (uint32_t q = first; q < quads; q += stride, w.step()) {
            const float4* src = rgba + (size_t)((uint32_t)H - 1u - w.y) * (uint32_t)W + (w.x << 2);   /* :1359 flip */
            const float4 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
            const uint32_t b[12] = {to_u8<HALF>(thr, p0.x), to_u8<HALF>(thr, p0.y), to_u8<HALF>(thr, p0.z),
                                    to_u8<HALF>(thr, p1.x), to_u8<HALF>(thr, p1.y), to_u8<HALF>(thr, p1.z),
                                    to_u8<HALF>(thr, p2.x), to_u8<HALF>(thr, p2.y), to_u8<HALF>(thr, p2.z),
                                    to_u8<HALF>(thr, p3.x), to_u8<HALF>(thr, p3.y), to_u8<HALF>(thr, p3.z)};
            uint32_t* o = out + (size_t)q * 3;
            o[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            o[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
            o[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
        }
        return;
    }
    QuadWalk w(first, stride, (uint32_t)W);                                   /* single pixels: the same walk, W per row */
    for (uint32_t idx = first; idx < n; idx += stride, w.step()) {
        const float4 v = rgba[(size_t)((uint32_t)H - 1u - w.y) * (uint32_t)W + w.x];
        uint8_t* o = rgb8 + (size_t)idx * 3;
        o[0] = (uint8_t)to_u8<HALF>(thr, v.x);
        o[1] = (uint8_t)to_u8<HALF>(thr, v.y);
        o[2] = (uint8_t)to_u8<HALF>(thr, v.z);
    }
}

template <bool HALF>
__global__ void __launch_bounds__(kBlockThreads)
export_rgb16_kernel(const float4* __restrict__ rgba, uint16_t* __restrict__ rgb16, int W, int H, int quads_ok)
{
    const uint32_t n = (uint32_t)W * (uint32_t)H;
    const uint32_t stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
    if (quads_ok) {
        const uint32_t quads = n >> 2;
        uint2* out = reinterpret_cast<uint2*>(rgb16);                         /* 24 bytes per quad: 8-byte aligned */
        QuadWalk w(first, stride, (uint32_t)W >> 2);
        for (uint32_t q = first; q < quads; q += stride, w.step()) {
            const float4* src = rgba + (size_t)((uint32_t)H - 1u - w.y) * (uint32_t)W + (w.x << 2);   /* :2058 flip */
            const float4 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
            const uint32_t h[12] = {to_u16<HALF>(p0.x), to_u16<HALF>(p0.y), to_u16<HALF>(p0.z),
                                    to_u16<HALF>(p1.x), to_u16<HALF>(p1.y), to_u16<HALF>(p1.z),
                                    to_u16<HALF>(p2.x), to_u16<HALF>(p2.y), to_u16<HALF>(p2.z),
                                    to_u16<HALF>(p3.x), to_u16<HALF>(p3.y), to_u16<HALF>(p3.z)};
            uint2* o = out + (size_t)q * 3;
            o[0] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            o[1] = make_uint2(h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            o[2] = make_uint2(h[8] | (h[9] << 16), h[10] | (h[11] << 16));
        }
        return;
    }
    QuadWalk w(first, stride, (uint32_t)W);
    for (uint32_t idx = first; idx < n; idx += stride, w.step()) {
        const float4 v = rgba[(size_t)((uint32_t)H - 1u - w.y) * (uint32_t)W + w.x];
        uint16_t* o = rgb16 + (size_t)idx * 3;
        o[0] = (uint16_t)to_u16<HALF>(v.x);
        o[1] = (uint16_t)to_u16<HALF>(v.y);
        o[2] = (uint16_t)to_u16<HALF>(v.z);
    }
}

}  // namespace fr
