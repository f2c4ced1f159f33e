/*
 * fr_oracle.c -- CPU restatement of the reference's escape-time hot path.
 *
 * TEST INFRASTRUCTURE ONLY; see fr_oracle.h.  Parity pin: the reference holds no tests or
 * golden vectors; the fp32 variants are checked against vectors obtained by executing the
 * reference's compiled shaders (tests/golden/spv_frames.npz, tests/test_spv_golden.py).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "fr_oracle.h"

#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct fro_sample {
    int32_t iter;
    double  nu, zre, zim;
    int64_t executed;
    float   rgb[3];
} fro_sample;

/* ------------------------------------------------------------------ GLSL helpers */
static inline float fract_f(float t) { return t - floorf(t); }
static inline float clamp01_f(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
/* GLSL mix(x, y, a) = x*(1-a) + y*a */
static inline void mix3(const float a[3], const float b[3], float k, float out[3])
{
    for (int c = 0; c < 3; ++c) out[c] = a[c] * (1.0f - k) + b[c] * k;
}
/* GLSL smoothstep(0, 1, t) */
static inline float smoothstep01(float t)
{
    t = clamp01_f((t - 0.0f) / (1.0f - 0.0f));
    return t * t * (3.0f - 2.0f * t);
}

/* five-knot ramps.  "quarters": breaks at .25/.5/.75, every segment mixes
 * (shaders/mandelbrot.comp:84-87).  "fifths": breaks at .2/.4/.6/.8 and the last
 * segment is the constant c5 (shaders/mandelbrot.comp:67-71). */
static void ramp_quarters(const float k[5][3], float t, float out[3])
{
    if (t < 0.25f)      mix3(k[0], k[1], t * 4.0f, out);
    else if (t < 0.5f)  mix3(k[1], k[2], (t - 0.25f) * 4.0f, out);
    else if (t < 0.75f) mix3(k[2], k[3], (t - 0.5f) * 4.0f, out);
    else                mix3(k[3], k[4], (t - 0.75f) * 4.0f, out);
}
static void ramp_fifths(const float k[5][3], float t, float out[3])
{
    if (t < 0.2f)       mix3(k[0], k[1], t * 5.0f, out);
    else if (t < 0.4f)  mix3(k[1], k[2], (t - 0.2f) * 5.0f, out);
    else if (t < 0.6f)  mix3(k[2], k[3], (t - 0.4f) * 5.0f, out);
    else if (t < 0.8f)  mix3(k[3], k[4], (t - 0.6f) * 5.0f, out);
    else { out[0] = k[4][0]; out[1] = k[4][1]; out[2] = k[4][2]; }
}

/* ------------------------------------------------------------------ palettes */
/* shaders/mandelbrot.comp:60-141 */
static void palette_mandelbrot(int32_t mode, float t, float out[3])
{
    static const float fire[5][3]     = {{0.0f,0.0f,0.1f},{0.8f,0.0f,0.0f},{1.0f,0.3f,0.0f},{1.0f,0.9f,0.0f},{1.0f,1.0f,0.95f}};
    static const float electric[5][3] = {{0.0f,0.0f,0.05f},{0.0f,0.1f,0.4f},{0.0f,0.5f,1.0f},{0.3f,0.8f,1.0f},{0.8f,1.0f,1.0f}};
    static const float nebula[5][3]   = {{0.02f,0.00f,0.05f},{0.15f,0.00f,0.25f},{0.00f,0.40f,0.60f},{0.00f,0.90f,1.00f},{0.90f,0.95f,1.00f}};
    static const float solar[5][3]    = {{0.1f,0.0f,0.1f},{0.5f,0.0f,0.2f},{0.9f,0.3f,0.0f},{1.0f,0.8f,0.3f},{1.0f,1.0f,0.9f}};
    static const float ocean[5][3]    = {{0.0f,0.05f,0.08f},{0.0f,0.3f,0.5f},{0.0f,0.7f,0.9f},{0.2f,0.9f,1.0f},{0.9f,1.0f,1.0f}};
    t = fract_f(t);                                               /* :130 */
    switch (mode) {
    case 1: ramp_quarters(electric, smoothstep01(t), out); return;             /* :74-88 */
    case 2: out[0] = out[1] = out[2] = t; return;                              /* :90-92 */
    case 3: ramp_quarters(nebula, fract_f(t), out); return;                    /* :94-105 */
    case 4: ramp_quarters(solar, powf(fract_f(t), 0.9f), out); return;         /* :107-118 */
    case 5: ramp_quarters(ocean, powf(fract_f(t), 0.85f), out); return;        /* :120-131 */
    case 0: default: ramp_fifths(fire, powf(t, 0.7f), out); return;            /* :60-72, :139 */
    }
}

/* shaders/julia.comp:20-181 */
static void palette_julia(int32_t mode, float t, float out[3])
{
    static const float ultra_fire[5][3] = {{0.0f,0.0f,0.1f},{0.8f,0.0f,0.0f},{1.0f,0.3f,0.0f},{1.0f,0.9f,0.0f},{1.0f,1.0f,0.95f}};
    static const float electric[5][3]   = {{0.0f,0.0f,0.05f},{0.0f,0.1f,0.4f},{0.0f,0.5f,1.0f},{0.3f,0.8f,1.0f},{0.8f,1.0f,1.0f}};
    static const float ocean[5][3]      = {{0.0f,0.0f,0.1f},{0.0f,0.1f,0.3f},{0.0f,0.4f,0.7f},{0.0f,0.7f,1.0f},{0.5f,1.0f,1.0f}};
    static const float sunset[5][3]     = {{0.1f,0.0f,0.2f},{0.5f,0.1f,0.3f},{1.0f,0.3f,0.2f},{1.0f,0.7f,0.3f},{1.0f,0.95f,0.7f}};
    static const float cosmic[5][3]     = {{0.0f,0.0f,0.0f},{0.2f,0.0f,0.4f},{0.4f,0.0f,0.6f},{0.8f,0.3f,0.9f},{1.0f,0.7f,1.0f}};
    static const float gold[5][3]       = {{0.1f,0.05f,0.0f},{0.4f,0.2f,0.0f},{0.8f,0.5f,0.1f},{1.0f,0.8f,0.3f},{1.0f,1.0f,0.9f}};
    static const float vapor[5][3]      = {{0.1f,0.0f,0.2f},{0.5f,0.0f,0.5f},{1.0f,0.0f,0.8f},{0.0f,0.8f,1.0f},{1.0f,0.5f,1.0f}};
    static const float forest[5][3]     = {{0.0f,0.05f,0.0f},{0.0f,0.2f,0.1f},{0.1f,0.5f,0.2f},{0.3f,0.8f,0.4f},{0.8f,1.0f,0.6f}};
    static const float lava[5][3]       = {{0.1f,0.0f,0.0f},{0.6f,0.0f,0.0f},{1.0f,0.2f,0.0f},{1.0f,0.6f,0.0f},{1.0f,1.0f,0.5f}};
    t = fract_f(t);                                               /* :163 */
    switch (mode) {
    case 1: ramp_quarters(electric, smoothstep01(t), out); return;             /* :37-51 */
    case 2: ramp_quarters(ocean, smoothstep01(t), out); return;                /* :54-68 */
    case 3: ramp_fifths(sunset, t, out); return;                               /* :71-84 */
    case 4: {                                                                  /* :87-101 */
        const float w = powf(t, 0.8f);
        if (w < 0.3f)      mix3(cosmic[0], cosmic[1], w / 0.3f, out);
        else if (w < 0.5f) mix3(cosmic[1], cosmic[2], (w - 0.3f) / 0.2f, out);
        else if (w < 0.7f) mix3(cosmic[2], cosmic[3], (w - 0.5f) / 0.2f, out);
        else               mix3(cosmic[3], cosmic[4], (w - 0.7f) / 0.3f, out);
        return;
    }
    case 5: ramp_quarters(gold, smoothstep01(t), out); return;                 /* :104-118 */
    case 6: ramp_quarters(vapor, t, out); return;                              /* :121-132 */
    case 7: ramp_quarters(forest, t, out); return;                             /* :135-146 */
    case 8: {                                                                  /* :149-163 */
        const float w = powf(t, 0.6f);
        if (w < 0.2f)      mix3(lava[0], lava[1], w * 5.0f, out);
        else if (w < 0.4f) mix3(lava[1], lava[2], (w - 0.2f) * 5.0f, out);
        else if (w < 0.7f) mix3(lava[2], lava[3], (w - 0.4f) / 0.3f, out);
        else               mix3(lava[3], lava[4], (w - 0.7f) / 0.3f, out);
        return;
    }
    case 9: out[0] = out[1] = out[2] = t; return;                              /* :166-168 */
    case 0: default: ramp_fifths(ultra_fire, powf(t, 0.7f), out); return;      /* :20-34, :178 */
    }
}

void fro_palette(int32_t shader, int32_t mode, float t, float rgb[3])
{
    if (shader == 0) palette_mandelbrot(mode, t, rgb);
    else             palette_julia(mode, t, rgb);
}

/* ------------------------------------------------------------------ post chain */
/* aces_tonemap, shaders/mandelbrot.comp:38-45 == src/vk_engine.cpp:1344-1351 */
static inline float aces(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return clamp01_f((x * (a * x + b)) / (x * (c * x + d) + e));
}

void fro_post_chain(float rgb[3], float brightness, float saturation, float contrast,
                    int32_t julia_floors)
{
    if (julia_floors) {                                           /* shaders/julia.comp:319-322 */
        brightness = fmaxf(brightness, 0.1f);
        saturation = fmaxf(saturation, 0.0f);
        contrast = fmaxf(contrast, 0.1f);
    }
    /* enhance_color, shaders/mandelbrot.comp:48-54 */
    float c[3];
    for (int k = 0; k < 3; ++k) c[k] = rgb[k] * brightness;
    for (int k = 0; k < 3; ++k) c[k] = (c[k] - 0.5f) * contrast + 0.5f;
    const float gray = c[0] * 0.299f + c[1] * 0.587f + c[2] * 0.114f;
    for (int k = 0; k < 3; ++k) c[k] = clamp01_f(gray * (1.0f - saturation) + c[k] * saturation);
    /* aces + gamma, :234-235 */
    for (int k = 0; k < 3; ++k) rgb[k] = powf(aces(c[k]), 1.0f / 2.2f);
}

/* ------------------------------------------------------------------ per-sample code, two precisions */
#define FN(name) name##_f32
#define REAL float
#define LOG logf
#define SQRT sqrtf
#define FMIN fminf
#define FABS fabsf
#define FLOOR floorf
#define ATAN2 atan2f
#define SIN sinf
#define PAL_ARG(t) (t)
#include "fr_oracle_sample.inc"
#undef PAL_ARG
#undef FN
#undef REAL
#undef LOG
#undef SQRT
#undef FMIN
#undef FABS
#undef FLOOR
#undef ATAN2
#undef SIN

#define FN(name) name##_f64
#define REAL double
#define LOG log
#define SQRT sqrt
#define FMIN fmin
#define FABS fabs
#define FLOOR floor
#define ATAN2 atan2
#define SIN sin
#define PAL_ARG(t) ((float)((t) - floor(t)))
#include "fr_oracle_sample.inc"
#undef PAL_ARG
#undef FN
#undef REAL
#undef LOG
#undef SQRT
#undef FMIN
#undef FABS
#undef FLOOR
#undef ATAN2
#undef SIN

/* ------------------------------------------------------------------ Deep_Zoom (perturbation) */
/* shaders/test_deep_zoom.comp, the shader the host loads for FractalType::Deep_Zoom
 * (src/compute_effect_manager.cpp:133-135).  fp32 throughout, with float-float ("double-double" in
 * the shader's words) centre and zoom.  Restated WITH its quirks:
 *   - the view is  4*zoom/H  c-plane units high (pixel_size = zoom*4/H multiplies a NORMALISED
 *     offset, :128-136), H times smaller than the Mandelbrot shader's mapping;
 *   - the perturbation loop pairs dz_{i+1} with z_ref[i] (z_full = z_ref[i] + dz_{i+1}, :154-165);
 *   - the reference orbit is the fp64 orbit narrowed to float (src/deep_zoom_system.cpp:102-110);
 *   - no glitch detection, no tonemap/gamma, its own 4 palettes (:73-103). */
typedef struct { float hi, lo; } ff_t;
static ff_t dd_add_dd(ff_t a, ff_t b)                              /* :31-39 */
{
    const float s = a.hi + b.hi;
    const float v = s - a.hi;
    const float t = ((b.hi - v) + (a.hi - (s - v))) + (a.lo + b.lo);
    ff_t r; r.hi = s + t; r.lo = t - (r.hi - s); return r;
}
static ff_t dd_mul_sf(ff_t a, float b)                            /* :41-48 */
{
    const float p = a.hi * b;
    const float e = fmaf(a.hi, b, -p);
    float lo = fmaf(a.lo, b, e);
    ff_t r; r.hi = p + lo; r.lo = lo - (r.hi - p); return r;
}
static ff_t ff_split(double v)                                     /* src/compute_effect_manager.h:252-257 */
{
    ff_t r; r.hi = (float)v; r.lo = (float)(v - (double)r.hi); return r;
}
static void hsv2rgb(float h, float s, float v, float out[3])       /* :66-70 */
{
    const float K[4] = {1.0f, 2.0f / 3.0f, 1.0f / 3.0f, 3.0f};
    for (int c = 0; c < 3; ++c) {
        const float p = fabsf(fract_f(h + K[c]) * 6.0f - K[3]);
        const float q = clamp01_f(p - K[0]);
        out[c] = v * (K[0] * (1.0f - s) + q * s);
    }
}
static void dz_get_color(const fro_params* P, float iter, float max_iter, float zx, float zy,
                         float rgb[3], float* smooth)               /* :75-103 */
{
    if (iter >= max_iter - 0.5f) { rgb[0] = rgb[1] = rgb[2] = 0.0f; *smooth = max_iter; return; }
    float lenz = sqrtf(zx * zx + zy * zy);
    lenz = fmaxf(lenz, 1e-12f);
    const float log_zn = logf(lenz);
    const float nu = logf(log_zn / logf(2.0f)) / logf(2.0f);
    const float smooth_iter = iter + 1.0f - nu;
    *smooth = smooth_iter;
    const float t = smooth_iter * P->color_scale + P->color_offset;
    const int palette = P->palette_mode;
    if (palette == 0) hsv2rgb(fract_f(t * 0.05f), 0.8f, 0.9f, rgb);
    else if (palette == 1) {
        const float s = fract_f(t * 0.03f);
        const float a[3] = {0.0f, 0.1f, 0.3f}, b[3] = {1.0f, 1.0f, 1.0f};
        mix3(a, b, s, rgb);
    } else if (palette == 2) {
        const float s = fract_f(t * 0.04f);
        const float a[3] = {0.1f, 0.0f, 0.0f}, b[3] = {1.0f, 0.8f, 0.0f};
        mix3(a, b, s, rgb);
    } else { const float s = fract_f(t * 0.02f); rgb[0] = rgb[1] = rgb[2] = s; }
}

static void pixel_deep_zoom(const fro_params* P, const float* orbit, int32_t ref_iter,
                            int32_t px, int32_t py, int32_t W, int32_t H, fro_sample* o)   /* main(), :107-207 */
{
    const int32_t max_iter = P->max_iterations;
    const float bailout = fmaxf(2.0f, P->bailout);                 /* :114 */
    const float bailout_sq = bailout * bailout;
    const float uvx = (float)px / (float)W, uvy = (float)py / (float)H;        /* :118 */
    const ff_t center_x = ff_split(P->center_x), center_y = ff_split(P->center_y), zoom = ff_split(P->zoom);
    const float aspect = (float)W / (float)H;                     /* :125 */
    const ff_t pixel_size = dd_mul_sf(zoom, 4.0f / (float)H);      /* :128 */
    const float offset_x = (uvx - 0.5f) * aspect;                 /* :131-132 */
    const float offset_y = (uvy - 0.5f);
    const ff_t dc_x = dd_mul_sf(pixel_size, offset_x), dc_y = dd_mul_sf(pixel_size, offset_y);   /* :135-136 */
    const ff_t c_x_dd = dd_add_dd(center_x, dc_x), c_y_dd = dd_add_dd(center_y, dc_y);          /* :139-140 */
    const float delta_x = dc_x.hi + dc_x.lo, delta_y = dc_y.hi + dc_y.lo;                        /* :143 */
    float dzx = 0.0f, dzy = 0.0f;                                  /* :146 */
    const int32_t n_ref = max_iter < ref_iter ? max_iter : ref_iter;
    o->executed = 0;
    for (int32_t i = 0; i < n_ref; ++i) {                          /* :153-173 */
        const float zrx = orbit[2 * i], zry = orbit[2 * i + 1];
        const float mx = zrx * dzx - zry * dzy, my = zrx * dzy + zry * dzx;       /* c_mul(z_ref, dz) */
        const float t1x = mx * 2.0f, t1y = my * 2.0f;
        const float t2x = dzx * dzx - dzy * dzy, t2y = 2.0f * dzx * dzy;
        dzx = t1x + t2x + delta_x;
        dzy = t1y + t2y + delta_y;
        const float zfx = zrx + dzx, zfy = zry + dzy;
        o->executed++;
        if (zfx * zfx + zfy * zfy > bailout_sq) {
            float sm;
            dz_get_color(P, (float)i, (float)max_iter, zfx, zfy, o->rgb, &sm);
            o->iter = i; o->nu = (double)sm; o->zre = zfx; o->zim = zfy;
            return;
        }
    }
    float zx = 0.0f, zy = 0.0f;                                    /* :181-188 */
    const float c_fx = c_x_dd.hi + c_x_dd.lo, c_fy = c_y_dd.hi + c_y_dd.lo;
    if (ref_iter > 0) { zx = orbit[2 * (ref_iter - 1)] + dzx; zy = orbit[2 * (ref_iter - 1) + 1] + dzy; }
    else { zx = c_fx; zy = c_fy; }
    for (int32_t i = n_ref; i < max_iter; ++i) {                   /* :190-203 */
        const float z2x = zx * zx - zy * zy, z2y = 2.0f * zx * zy;
        zx = z2x + c_fx; zy = z2y + c_fy;
        o->executed++;
        if (zx * zx + zy * zy > bailout_sq) {
            float sm;
            dz_get_color(P, (float)i, (float)max_iter, zx, zy, o->rgb, &sm);
            o->iter = i; o->nu = (double)sm; o->zre = zx; o->zim = zy;
            return;
        }
    }
    o->iter = max_iter; o->nu = (double)max_iter; o->zre = zx; o->zim = zy;
    o->rgb[0] = o->rgb[1] = o->rgb[2] = 0.0f;                      /* :206 */
}

/* reference orbit as the shader sees it: fp64 orbit narrowed to float pairs; *len_out = reference_iterations */
static float* deep_zoom_orbit(const fro_params* P, int32_t* len_out)
{
    *len_out = 0;
    if (!P->use_perturbation) return NULL;                        /* src/deep_zoom_system.cpp:364 */
    double* xy = (double*)malloc((size_t)P->max_iterations * 2 * sizeof(double));
    if (!xy) return NULL;
    const int32_t n = fro_reference_orbit(P->center_x, P->center_y, P->max_iterations, xy);
    float* f = (float*)malloc((size_t)n * 2 * sizeof(float));
    if (f) { for (int32_t i = 0; i < 2 * n; ++i) f[i] = (float)xy[i]; *len_out = n; }
    free(xy);
    return f;
}

/* restatement of the colour stage alone, for checking fr_colorize_async and the multi-GPU exchange */
void fro_colorize(const fro_params* p, int64_t n, const double* nu, float* rgba)
{
    for (int64_t i = 0; i < n; ++i) {
        float rgb[3];
        if (p->precision == 1) colour_of_nu_f64(p, nu[i], rgb);
        else colour_of_nu_f32(p, (float)nu[i], rgb);
        rgba[4 * i + 0] = rgb[0]; rgba[4 * i + 1] = rgb[1]; rgba[4 * i + 2] = rgb[2]; rgba[4 * i + 3] = 1.0f;
    }
}

int32_t fro_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int64_t fro_render_rows(const fro_params* p, int32_t W, int32_t H,
                        int32_t y0, int32_t y1,
                        float* rgba, double* nu, int32_t* iter,
                        double* zre, double* zim, int32_t threads)
{
    int64_t total = 0;
    int32_t ref_iter = 0;
    float* orbit = p->fractal == 5 ? deep_zoom_orbit(p, &ref_iter) : NULL;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    /* one row per task, dynamic: rows through the set cost max_iter per pixel */
    #pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+:total)
    for (int32_t y = y0; y < y1; ++y) {
        for (int32_t x = 0; x < W; ++x) {
            fro_sample first;
            float rgb[3];
            int64_t executed;
            if (p->fractal == 5) {
                pixel_deep_zoom(p, orbit, ref_iter, x, y, W, H, &first);
                rgb[0] = first.rgb[0]; rgb[1] = first.rgb[1]; rgb[2] = first.rgb[2];
                executed = first.executed;
            }
            else if (p->precision == 0) pixel_f32(p, x, y, W, H, &first, rgb, &executed);
            else                   pixel_f64(p, x, y, W, H, &first, rgb, &executed);
            total += executed;
            const int64_t o = (int64_t)(y - y0) * W + x;
            if (rgba) { rgba[4*o] = rgb[0]; rgba[4*o+1] = rgb[1]; rgba[4*o+2] = rgb[2]; rgba[4*o+3] = 1.0f; }
            if (nu)   nu[o] = first.nu;
            if (iter) iter[o] = first.iter;
            if (zre)  zre[o] = first.zre;
            if (zim)  zim[o] = first.zim;
        }
    }
    free(orbit);
    return total;
}

/* ------------------------------------------------------------------ push constants */
/* src/compute_effect_manager.h:84-113 (Mandelbrot), :115-140 (Julia) */
void fro_pack_push_constants(const fro_params* p, float out[20])
{
    memset(out, 0, 20 * sizeof(float));
    if (p->fractal == 5) {                                        /* src/compute_effect_manager.h:240-324 */
        const ff_t cx = ff_split(p->center_x), cy = ff_split(p->center_y), z = ff_split(p->zoom);
        int32_t n = 0;
        float* orb = deep_zoom_orbit(p, &n);
        free(orb);
        out[0] = cx.hi; out[1] = cx.lo; out[2] = cy.hi; out[3] = cy.lo;
        out[4] = z.hi; out[5] = z.lo; out[6] = (float)p->max_iterations; out[7] = p->use_perturbation ? 1.0f : 0.0f;
        out[8] = p->color_offset; out[9] = p->color_scale; out[10] = (float)p->bailout; out[11] = (float)p->palette_mode;
        out[12] = (float)p->aa; out[13] = (float)n; out[14] = 0.0f; out[15] = 3.0f;   /* series approx off, order 3 (defaults) */
        return;
    }
    out[0] = (float)p->center_x;
    out[1] = (float)p->center_y;
    out[2] = (float)p->zoom;
    out[3] = (float)p->max_iterations;
    if (p->fractal == 0 || p->fractal == 2) {                     /* Burning Ship packs like Mandelbrot, :142-171 */
        out[4] = p->color_offset; out[5] = p->color_scale; out[6] = (float)p->bailout; out[7] = (float)p->palette_mode;
        out[8] = (float)p->aa; out[9] = (float)p->interior_style;
        out[10] = p->orbit_trap_enabled ? 1.0f : 0.0f; out[11] = p->orbit_trap_radius;
        out[12] = p->stripe_density; out[13] = p->stripe_enabled ? 1.0f : 0.0f;
        out[14] = p->brightness; out[15] = p->saturation;
        out[16] = p->contrast;
    } else {
        out[4] = (float)p->julia_c_real; out[5] = (float)p->julia_c_imag; out[6] = (float)p->bailout; out[7] = (float)p->color_offset;
        out[8] = (float)p->aa; out[9] = (float)p->color_scale; out[10] = p->brightness; out[11] = p->saturation;
        out[12] = p->contrast; out[13] = (float)p->palette_mode;
    }
}

/* ------------------------------------------------------------------ reference orbit */
/* src/deep_zoom_system.cpp:378-424.  z*z via std::complex<double>::operator* is the
 * textbook (ac-bd, ad+bc) for finite operands. */
int32_t fro_reference_orbit(double cx, double cy, int32_t max_iter, double* out_xy)
{
    double zr = 0.0, zi = 0.0;                                    /* :383 */
    int32_t escape_iter = max_iter;                               /* :390 */
    for (int32_t i = 0; i < max_iter; i++) {                      /* :391 */
        out_xy[2*i] = zr; out_xy[2*i+1] = zi;                     /* :392 store BEFORE iterating */
        const double mag = hypot(zr, zi);                         /* :396 std::abs */
        if (mag > 2.0) { escape_iter = i; break; }                /* :397-401 */
        if (mag > 1e10 || isnan(mag) || isinf(mag)) { escape_iter = i; break; }   /* :404-408 */
        const double nr = zr * zr - zi * zi;                      /* :411 z = z*z + c */
        const double ni = zr * zi + zi * zr;
        zr = nr + cx; zi = ni + cy;
    }
    return escape_iter < max_iter ? escape_iter + 1 : max_iter;   /* :422-424 */
}

/* ------------------------------------------------------------------ 8-bit export */
static uint16_t float_to_half_rne(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t absx = x & 0x7FFFFFFFu;
    if (absx >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((absx > 0x7F800000u) ? 0x200u : 0));
    if (absx >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);          /* rounds to inf */
    if (absx < 0x33000001u) return (uint16_t)sign;                         /* rounds to zero */
    int32_t e = (int32_t)(absx >> 23) - 127;
    uint32_t m = (absx & 0x7FFFFFu) | 0x800000u;
    uint32_t shift, hm;
    if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); hm = 0; }          /* subnormal half */
    else         { shift = 13; hm = (uint32_t)(e + 15) << 10; }
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    if (e >= -14) q -= 0x400u;                                             /* drop implicit bit; carry propagates into exponent */
    return (uint16_t)(sign | (hm + q));
}

/* lambda half_to_float, src/vk_engine.cpp:1313-1341 */
static float half_to_float(uint16_t h)
{
    uint16_t h_exp = (h & 0x7C00u), h_sig = (h & 0x03FFu);
    const uint32_t f_sgn = (uint32_t)(h & 0x8000u) << 16;
    uint32_t f_exp, f_sig;
    if (h_exp == 0x7C00u) { f_exp = 0x7F800000u >> 23; f_sig = (uint32_t)h_sig << 13; }
    else if (h_exp != 0)  { f_exp = (uint32_t)(h_exp >> 10) + 112; f_sig = (uint32_t)h_sig << 13; }
    else if (h_sig != 0) {
        int shift = 0;
        while ((h_sig & 0x0400u) == 0) { h_sig <<= 1; shift++; }
        h_sig &= 0x03FFu;
        f_exp = (uint32_t)(113 - shift); f_sig = (uint32_t)h_sig << 13;
    } else { f_exp = 0; f_sig = 0; }
    const uint32_t f = f_sgn | (f_exp << 23) | f_sig;
    float out; memcpy(&out, &f, 4);
    return out;
}

/* powf of the 8-bit export.  The reference calls its C runtime's powf (MSVC's; src/vk_engine.cpp:1367), and runtimes differ in
 * the last bit of powf: glibc's is one ulp off the correctly rounded result at a = 0x3C364A2A, where that decides between byte
 * 32 and byte 33.  The checker therefore uses the one value that does not depend on a runtime -- the CORRECTLY ROUNDED
 * single-precision power -- computed as the double-precision pow (error < 1 ulp of double) rounded to float, which is the
 * correctly rounded float unless a^e lies within 2^-29 (relative) of a rounding boundary of float; the library's table is
 * generated from a 200-bit power (tools/gen_export8_table.py), and the exhaustive scan below holds this function against it
 * at every float of [0, 1]. */
static inline float powf_cr(float a, float e) { return (float)pow((double)a, (double)e); }

/* src/vk_engine.cpp:1355-1371 */
void fro_export_rgb8(const float* rgba, int32_t W, int32_t H, uint8_t* rgb8, int32_t through_half)
{
    const float gamma = 1.0f / 2.2f;
    for (int32_t y = 0; y < H; y++) {
        const int32_t flipped = H - 1 - y;                        /* :1359 */
        for (int32_t x = 0; x < W; x++) {
            const int64_t src = ((int64_t)flipped * W + x) * 4, dst = ((int64_t)y * W + x) * 3;
            for (int c = 0; c < 3; c++) {
                float v = rgba[src + c];
                if (through_half) v = half_to_float(float_to_half_rne(v));
                v = aces(v);                                       /* :1366 */
                v = powf_cr(v, gamma);                             /* :1367 */
                rgb8[dst + c] = (uint8_t)(v * 255.0f);             /* :1368 */
            }
        }
    }
}

/* Exhaustive scan of the byte of the 8-bit export, (uint8)(powf(a, 1/2.2f) * 255.0f) (src/vk_engine.cpp:1367-1368), over
 * EVERY float a of [0, 1] (bit patterns 0 .. 0x3F800000): first[b] = the smallest bit pattern whose byte is b
 * (0xFFFFFFFF if no float maps to b); returns the number of places where the byte DEcreases from one float to the next
 * (0 = the byte is monotone in a, which is what lets a threshold table describe it).  Checker only: the library builds its
 * table by bisection (fr_export8_thresholds) and the tests hold it against this scan. */
int64_t fro_export8_scan(uint32_t first[256])
{
    const float gamma = 1.0f / 2.2f;
    const uint32_t last = 0x3F800000u;
    const uint32_t chunk = 1u << 20;
    const uint32_t nchunks = last / chunk + 1u;
    int64_t violations = 0;
    for (int b = 0; b < 256; b++) first[b] = 0xFFFFFFFFu;
#pragma omp parallel
    {
        uint32_t mine[256];
        int64_t bad = 0;
        for (int b = 0; b < 256; b++) mine[b] = 0xFFFFFFFFu;
#pragma omp for schedule(dynamic, 1)
        for (uint32_t c = 0; c < nchunks; c++) {
            const uint32_t lo = c * chunk;
            uint32_t hi = lo + chunk - 1u;
            if (hi > last) hi = last;
            uint32_t prev = 0;
            if (lo > 0) { const uint32_t pb = lo - 1u; float a; memcpy(&a, &pb, 4); prev = (uint32_t)(uint8_t)(powf_cr(a, gamma) * 255.0f); }
            for (uint32_t bits = lo; ; bits++) {
                float a;
                memcpy(&a, &bits, 4);
                const uint32_t by = (uint32_t)(uint8_t)(powf_cr(a, gamma) * 255.0f);
                if (by < prev) bad++;
                if (bits < mine[by]) mine[by] = bits;
                prev = by;
                if (bits == hi) break;
            }
        }
#pragma omp critical
        {
            violations += bad;
            for (int b = 0; b < 256; b++) if (mine[b] < first[b]) first[b] = mine[b];
        }
    }
    return violations;
}
