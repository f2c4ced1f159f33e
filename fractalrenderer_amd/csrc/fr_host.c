/*
 * fr_host.c -- host side of the C ABI that needs no GPU: parameter block, validation,
 * push-constant packing, row-strip arithmetic, palette knot tables, reference orbit.
 * Plain C11 (the reference's host is C++; the hot path's host logic is small enough
 * to stay in C behind the C ABI).
 */
#include "fr_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

/* ---- errors ---------------------------------------------------------------------------------- */
static _Thread_local char g_err[512];

int fr_set_error(int status, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return status;
}

const char* fr_last_error(void) { return g_err; }

const char* fr_status_string(int status)
{
    switch (status) {
    case FR_OK: return "ok";
    case FR_ERR_INVALID_ARG: return "invalid argument";
    case FR_ERR_NO_DEVICE: return "no HIP device";
    case FR_ERR_HIP: return "HIP runtime error";
    case FR_ERR_UNSUPPORTED: return "fractal type outside the hot path";
    case FR_ERR_IO: return "I/O error";
    case FR_ERR_PARSE: return "parse error";
    case FR_ERR_NOMEM: return "out of memory";
    case FR_ERR_INTERNAL: return "internal error (frame incomplete)";
    default: return "unknown status";
    }
}

void fr_version(int* major, int* minor)
{
    if (major) *major = FR_VERSION_MAJOR;
    if (minor) *minor = FR_VERSION_MINOR;
}

/* ---- parameters ------------------------------------------------------------------------------ */

/* FractalState member initialisers, src/fractal_state.h:18-51,77-79 */
int fr_params_default(fr_params* p)
{
    if (!p) return fr_set_error(FR_ERR_INVALID_ARG, "params is NULL");
    memset(p, 0, sizeof *p);
    p->fractal_type = FR_FRACTAL_MANDELBROT;
    p->precision = FR_PRECISION_F64;
    p->center_x = -0.5;  p->center_y = 0.0;  p->zoom = 3.0;          /* :18-20 */
    p->max_iterations = 256;                                          /* :21 */
    p->julia_c_real = (double)-0.7f;  p->julia_c_imag = (double)0.27015f;   /* :29-30 (floats) */
    p->bailout = 4.0f;  p->antialiasing_samples = 1;                  /* :36-37 */
    p->palette_mode = 0;  p->color_offset = 0.0f;  p->color_scale = 1.0f;   /* :40-42 */
    p->interior_style = 0;                                            /* :47 */
    p->orbit_trap_enabled = 0;  p->orbit_trap_radius = 0.5f;          /* :48-49 */
    p->stripe_enabled = 0;  p->stripe_density = 10.0f;                /* :50-51 */
    p->color_brightness = 1.0f;  p->color_saturation = 1.0f;  p->color_contrast = 1.0f;   /* :77-79 */
    p->flags = 0;
    p->use_perturbation = 0;                                          /* :86 */
    return FR_OK;
}

/* FractalState::reset(), src/fractal_state.h:135-153 */
int fr_params_reset(fr_params* p)
{
    if (!p) return fr_set_error(FR_ERR_INVALID_ARG, "params is NULL");
    p->center_x = -0.5;  p->center_y = 0.0;  p->zoom = 1.5;  p->max_iterations = 256;   /* :137-140 */
    p->color_brightness = 1.0f;  p->color_saturation = 1.0f;  p->color_contrast = 1.0f; /* :145-147 */
    return FR_OK;
}

int fr_params_validate(const fr_params* p, uint32_t width, uint32_t height)
{
    if (!p) return fr_set_error(FR_ERR_INVALID_ARG, "params is NULL");
    if (width == 0 || height == 0)
        return fr_set_error(FR_ERR_INVALID_ARG, "width and height must be > 0 (got %ux%u)", width, height);
    if ((uint64_t)width * (uint64_t)height >= (1ull << 31))
        return fr_set_error(FR_ERR_INVALID_ARG, "frame %ux%u has 2^31 pixels or more", width, height);
    if (p->fractal_type < 0 || p->fractal_type > FR_FRACTAL_DEEP_ZOOM)
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown fractal_type %d", p->fractal_type);
    if (p->fractal_type == FR_FRACTAL_MANDELBULB || p->fractal_type == FR_FRACTAL_PHOENIX)
        return fr_set_error(FR_ERR_UNSUPPORTED,
                            "fractal_type %d is outside the hot path (Mandelbrot, JuliaSet, BurningShip and Deep_Zoom only)",
                            p->fractal_type);
    if (p->precision != FR_PRECISION_F32 && p->precision != FR_PRECISION_F64)
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown precision %d", p->precision);
    if (p->max_iterations < 1 || p->max_iterations > (1 << 24))
        return fr_set_error(FR_ERR_INVALID_ARG, "max_iterations %d outside [1, 2^24]", p->max_iterations);
    if (!isfinite(p->zoom) || p->zoom == 0.0)
        return fr_set_error(FR_ERR_INVALID_ARG, "zoom must be finite and non-zero");
    if (!isfinite(p->center_x) || !isfinite(p->center_y))
        return fr_set_error(FR_ERR_INVALID_ARG, "centre must be finite");
    if (!isfinite(p->julia_c_real) || !isfinite(p->julia_c_imag))
        return fr_set_error(FR_ERR_INVALID_ARG, "julia c must be finite");
    if (!isfinite(p->bailout) || !(p->bailout > 0.0f))
        return fr_set_error(FR_ERR_INVALID_ARG, "bailout must be finite and > 0");
    if (p->antialiasing_samples < 0 || p->antialiasing_samples > 16)
        return fr_set_error(FR_ERR_INVALID_ARG, "antialiasing_samples %d outside [0, 16]", p->antialiasing_samples);
    return FR_OK;
}

/* ComputeEffect::update_from_state, src/compute_effect_manager.h:84-113 / :115-140 */
int fr_pack_push_constants(const fr_params* p, float out[20])
{
    if (!p || !out) return fr_set_error(FR_ERR_INVALID_ARG, "params/out is NULL");
    for (int i = 0; i < 20; ++i) out[i] = 0.0f;                       /* "= {}", :81 */
    switch (p->fractal_type) {
    case FR_FRACTAL_MANDELBROT:
    case FR_FRACTAL_BURNING_SHIP:                                            /* same layout, :142-171 */
        out[0] = (float)p->center_x;  out[1] = (float)p->center_y;           /* data1, :86-91 */
        out[2] = (float)p->zoom;      out[3] = (float)p->max_iterations;
        out[4] = p->color_offset;     out[5] = p->color_scale;               /* data2, :92-97 */
        out[6] = (float)p->bailout;   out[7] = (float)p->palette_mode;
        out[8] = (float)p->antialiasing_samples;  out[9] = (float)p->interior_style;   /* data3, :98-103 */
        out[10] = p->orbit_trap_enabled ? 1.0f : 0.0f;  out[11] = p->orbit_trap_radius;
        out[12] = p->stripe_density;  out[13] = p->stripe_enabled ? 1.0f : 0.0f;       /* data4, :104-109 */
        out[14] = p->color_brightness;  out[15] = p->color_saturation;
        out[16] = p->color_contrast;                                         /* data5, :110-113 */
        return FR_OK;
    case FR_FRACTAL_JULIA:
        out[0] = (float)p->center_x;  out[1] = (float)p->center_y;           /* data1, :116-121 */
        out[2] = (float)p->zoom;      out[3] = (float)p->max_iterations;
        out[4] = (float)p->julia_c_real;  out[5] = (float)p->julia_c_imag;   /* data2, :122-127 */
        out[6] = (float)p->bailout;   out[7] = (float)p->color_offset;
        out[8] = (float)p->antialiasing_samples;  out[9] = (float)p->color_scale;      /* data3, :128-133 */
        out[10] = p->color_brightness;  out[11] = p->color_saturation;
        out[12] = p->color_contrast;  out[13] = (float)p->palette_mode;      /* data4, :134-138 */
        return FR_OK;                                                        /* data5 = 0, :139 */
    case FR_FRACTAL_DEEP_ZOOM: {                                             /* :236-324 */
        /* split_double, :252-257: hi = float(v), lo = float(v - double(hi)) */
        const float cxh = (float)p->center_x, cxl = (float)(p->center_x - (double)cxh);
        const float cyh = (float)p->center_y, cyl = (float)(p->center_y - (double)cyh);
        const float zh = (float)p->zoom, zl = (float)(p->zoom - (double)zh);
        out[0] = cxh; out[1] = cxl; out[2] = cyh; out[3] = cyl;             /* data1 */
        out[4] = zh;  out[5] = zl;  out[6] = (float)p->max_iterations;      /* data2 */
        out[7] = p->use_perturbation ? 1.0f : 0.0f;
        out[8] = p->color_offset;  out[9] = p->color_scale;                  /* data3 */
        out[10] = (float)p->bailout;  out[11] = (float)p->palette_mode;
        out[12] = (float)p->antialiasing_samples;                            /* data4: samples, reference_iterations, */
        out[13] = (float)fr_deep_zoom_reference_length(p);                   /*        use_series_approx (0), series_order (3) */
        out[14] = 0.0f;  out[15] = 3.0f;
        return FR_OK;                                                        /* data5 = 0 */
    }
    default:
        return fr_set_error(FR_ERR_UNSUPPORTED, "push-constant packing: fractal_type %d is outside the hot path",
                            p->fractal_type);
    }
}

/* reference_iterations of a Deep_Zoom render: the trimmed length of the fp64 orbit at the centre,
 * or 0 when perturbation is off (compute_reference_orbit returns early, src/deep_zoom_system.cpp:364) */
int32_t fr_deep_zoom_reference_length(const fr_params* p)
{
    if (!p || !p->use_perturbation || p->max_iterations < 1) return 0;
    double zr = 0.0, zi = 0.0;
    for (int32_t i = 0; i < p->max_iterations; ++i) {
        const double mag = hypot(zr, zi);
        if (mag > 2.0 || mag > 1e10 || isnan(mag) || isinf(mag)) return i + 1;
        const double re = zr * zr - zi * zi, im = zr * zi + zi * zr;
        zr = re + p->center_x;
        zi = im + p->center_y;
    }
    return p->max_iterations;
}

/* ---- row strips -------------------------------------------------------------------------------- */
static void shard_normalise(const fr_shard* s, uint32_t height, uint32_t* part, uint32_t* nparts, uint32_t* R)
{
    *nparts = (s && s->nparts) ? s->nparts : 1u;
    *part = s ? s->part : 0u;
    *R = (s && s->rows_per_strip) ? s->rows_per_strip : (*nparts == 1u ? height : 1u);
    if (*R == 0) *R = 1;
}

uint32_t fr_shard_rows(const fr_shard* s, uint32_t height)
{
    uint32_t part, nparts, R;
    shard_normalise(s, height, &part, &nparts, &R);
    if (part >= nparts || height == 0) return 0;
    const uint32_t nstrips = (height + R - 1) / R;                 /* last strip may be short */
    if (part >= nstrips) return 0;
    const uint32_t mine = (nstrips - part + nparts - 1) / nparts;   /* strips part, part+nparts, ... */
    const uint32_t last = part + (mine - 1) * nparts;               /* my last strip's index */
    uint32_t rows = mine * R;
    if (last == nstrips - 1) rows -= nstrips * R - height;          /* trim the short strip */
    return rows;
}

uint32_t fr_shard_global_row(const fr_shard* s, uint32_t height, uint32_t local_row)
{
    uint32_t part, nparts, R;
    shard_normalise(s, height, &part, &nparts, &R);
    if (local_row >= fr_shard_rows(s, height)) return UINT32_MAX;
    const uint32_t strip = local_row / R;
    return (strip * nparts + part) * R + (local_row - strip * R);
}

/* ---- palette knot tables ------------------------------------------------------------------------ */
static void set_knots(fr_palette_table* t, const float k[5][3])
{
    for (int i = 0; i < 5; ++i) {
        t->knot[i][0] = k[i][0]; t->knot[i][1] = k[i][1]; t->knot[i][2] = k[i][2]; t->knot[i][3] = 0.0f;
    }
    for (int c = 0; c < 4; ++c) t->knot[5][c] = t->knot[4][c];
}
/* breaks .25/.5/.75, factor *4 (e.g. shaders/mandelbrot.comp:84-87) */
static void ramp_quarters(fr_palette_table* t)
{
    static const float lo[4] = {0.0f, 0.25f, 0.5f, 0.75f};
    t->nseg = 4; t->last_const = 0;
    for (int i = 0; i < 4; ++i) { t->seg_lo[i] = lo[i]; t->seg_k[i] = 4.0f; t->seg_div[i] = 0; }
    t->seg_lo[4] = 1.0f; t->seg_k[4] = 1.0f; t->seg_div[4] = 0;
}
/* breaks .2/.4/.6/.8, factor *5, last segment constant (shaders/mandelbrot.comp:67-71) */
static void ramp_fifths(fr_palette_table* t)
{
    static const float lo[5] = {0.0f, 0.2f, 0.4f, 0.6f, 0.8f};
    t->nseg = 5; t->last_const = 1;
    for (int i = 0; i < 5; ++i) { t->seg_lo[i] = lo[i]; t->seg_k[i] = 5.0f; t->seg_div[i] = 0; }
}

static void palette_table_fill(int shader, int mode, fr_palette_table* t);

void fr_palette_table_build(int shader, int mode, fr_palette_table* t)
{
    palette_table_fill(shader, mode, t);
    t->any_div = 0;
    for (int i = 0; i < t->nseg && i < 5; ++i) t->any_div |= t->seg_div[i] != 0;
}

static void palette_table_fill(int shader, int mode, fr_palette_table* t)
{
    /* knots: shaders/mandelbrot.comp:60-131 and shaders/julia.comp:20-163 */
    static const float fire[5][3]       = {{0.0f,0.0f,0.1f},{0.8f,0.0f,0.0f},{1.0f,0.3f,0.0f},{1.0f,0.9f,0.0f},{1.0f,1.0f,0.95f}};
    static const float electric[5][3]   = {{0.0f,0.0f,0.05f},{0.0f,0.1f,0.4f},{0.0f,0.5f,1.0f},{0.3f,0.8f,1.0f},{0.8f,1.0f,1.0f}};
    static const float nebula[5][3]     = {{0.02f,0.00f,0.05f},{0.15f,0.00f,0.25f},{0.00f,0.40f,0.60f},{0.00f,0.90f,1.00f},{0.90f,0.95f,1.00f}};
    static const float solar[5][3]      = {{0.1f,0.0f,0.1f},{0.5f,0.0f,0.2f},{0.9f,0.3f,0.0f},{1.0f,0.8f,0.3f},{1.0f,1.0f,0.9f}};
    static const float ocean_m[5][3]    = {{0.0f,0.05f,0.08f},{0.0f,0.3f,0.5f},{0.0f,0.7f,0.9f},{0.2f,0.9f,1.0f},{0.9f,1.0f,1.0f}};
    static const float ocean_j[5][3]    = {{0.0f,0.0f,0.1f},{0.0f,0.1f,0.3f},{0.0f,0.4f,0.7f},{0.0f,0.7f,1.0f},{0.5f,1.0f,1.0f}};
    static const float sunset[5][3]     = {{0.1f,0.0f,0.2f},{0.5f,0.1f,0.3f},{1.0f,0.3f,0.2f},{1.0f,0.7f,0.3f},{1.0f,0.95f,0.7f}};
    static const float cosmic[5][3]     = {{0.0f,0.0f,0.0f},{0.2f,0.0f,0.4f},{0.4f,0.0f,0.6f},{0.8f,0.3f,0.9f},{1.0f,0.7f,1.0f}};
    static const float gold[5][3]       = {{0.1f,0.05f,0.0f},{0.4f,0.2f,0.0f},{0.8f,0.5f,0.1f},{1.0f,0.8f,0.3f},{1.0f,1.0f,0.9f}};
    static const float vapor[5][3]      = {{0.1f,0.0f,0.2f},{0.5f,0.0f,0.5f},{1.0f,0.0f,0.8f},{0.0f,0.8f,1.0f},{1.0f,0.5f,1.0f}};
    static const float forest[5][3]     = {{0.0f,0.05f,0.0f},{0.0f,0.2f,0.1f},{0.1f,0.5f,0.2f},{0.3f,0.8f,0.4f},{0.8f,1.0f,0.6f}};
    static const float lava[5][3]       = {{0.1f,0.0f,0.0f},{0.6f,0.0f,0.0f},{1.0f,0.2f,0.0f},{1.0f,0.6f,0.0f},{1.0f,1.0f,0.5f}};

    memset(t, 0, sizeof *t);
    t->warp = FR_WARP_NONE; t->warp_exp = 1.0f;
    if (shader == 0) {
        /* get_palette_color, shaders/mandelbrot.comp:129-141: modes outside 0..5 are fire */
        switch (mode) {
        case 1: t->warp = FR_WARP_SMOOTHSTEP; ramp_quarters(t); set_knots(t, electric); return;
        case 2: t->warp = FR_WARP_GRAY; ramp_quarters(t); set_knots(t, fire); return;
        case 3: ramp_quarters(t); set_knots(t, nebula); return;            /* fract(fract(t)) == fract(t) */
        case 4: t->warp = FR_WARP_POW; t->warp_exp = 0.9f; ramp_quarters(t); set_knots(t, solar); return;
        case 5: t->warp = FR_WARP_POW; t->warp_exp = 0.85f; ramp_quarters(t); set_knots(t, ocean_m); return;
        default: t->warp = FR_WARP_POW; t->warp_exp = 0.7f; ramp_fifths(t); set_knots(t, fire); return;
        }
    }
    /* get_palette_color, shaders/julia.comp:162-181: modes outside 0..9 are ultra_fire */
    switch (mode) {
    case 1: t->warp = FR_WARP_SMOOTHSTEP; ramp_quarters(t); set_knots(t, electric); return;
    case 2: t->warp = FR_WARP_SMOOTHSTEP; ramp_quarters(t); set_knots(t, ocean_j); return;
    case 3: ramp_fifths(t); set_knots(t, sunset); return;
    case 4:                                                                /* :87-101 */
        t->warp = FR_WARP_POW; t->warp_exp = 0.8f; set_knots(t, cosmic);
        t->nseg = 4; t->last_const = 0;
        t->seg_lo[0] = 0.0f; t->seg_lo[1] = 0.3f; t->seg_lo[2] = 0.5f; t->seg_lo[3] = 0.7f; t->seg_lo[4] = 1.0f;
        t->seg_k[0] = 0.3f; t->seg_k[1] = 0.2f; t->seg_k[2] = 0.2f; t->seg_k[3] = 0.3f; t->seg_k[4] = 1.0f;
        t->seg_div[0] = t->seg_div[1] = t->seg_div[2] = t->seg_div[3] = 1;
        return;
    case 5: t->warp = FR_WARP_SMOOTHSTEP; ramp_quarters(t); set_knots(t, gold); return;
    case 6: ramp_quarters(t); set_knots(t, vapor); return;
    case 7: ramp_quarters(t); set_knots(t, forest); return;
    case 8:                                                                /* :149-163 */
        t->warp = FR_WARP_POW; t->warp_exp = 0.6f; set_knots(t, lava);
        t->nseg = 4; t->last_const = 0;
        t->seg_lo[0] = 0.0f; t->seg_lo[1] = 0.2f; t->seg_lo[2] = 0.4f; t->seg_lo[3] = 0.7f; t->seg_lo[4] = 1.0f;
        t->seg_k[0] = 5.0f; t->seg_k[1] = 5.0f; t->seg_k[2] = 0.3f; t->seg_k[3] = 0.3f; t->seg_k[4] = 1.0f;
        t->seg_div[0] = 0; t->seg_div[1] = 0; t->seg_div[2] = 1; t->seg_div[3] = 1;
        return;
    case 9: t->warp = FR_WARP_GRAY; ramp_quarters(t); set_knots(t, fire); return;
    default: t->warp = FR_WARP_POW; t->warp_exp = 0.7f; ramp_fifths(t); set_knots(t, fire); return;
    }
}

/* ---- 8-bit export: exact bytes --------------------------------------------------------------------
 * The reference's byte is (uint8)(powf(a, 1/2.2f) * 255.0f) with a = aces(v) in [0, 1] (src/vk_engine.cpp:1366-1368).
 * As a function of a it is monotone non-decreasing (checked exhaustively, over every float of [0, 1], by
 * tests/test_host.py::test_export8_thresholds_against_an_exhaustive_scan), so it is fully described by 255
 * thresholds: t[b] = the smallest float whose byte is >= b.  The export kernel keeps its fast exp2(log2 a / 2.2) estimate
 * and corrects it by comparing a with t[estimate] and t[estimate + 1]: the bytes are then those of the powf form, for every
 * input.  t[0] = 0, t[256] = +inf.
 *
 * WHICH powf.  The last bit of powf depends on the C runtime (the reference builds against MSVC's; glibc's is one ulp off
 * the correctly rounded value at a = 0x3C364A2A, exactly where byte 32 becomes byte 33), so thresholds found by bisection
 * with the deployment host's libm -- rounds 2 and 3 -- differ between hosts and from the reference's by a float here and
 * there.  Since round 4 the table is BAKED IN: generated once from the correctly rounded single-precision power
 * (tools/gen_export8_table.py, mpmath at 200 bits), the same on every host.  "Exact" means exact with respect to that
 * definition; the host-libm bisection is kept (fr_export8_thresholds_host_powf) so that the tests can report where a host
 * differs. */
static const uint32_t kExport8Table[256] = {
#include "fr_export8_table.inc"
};

void fr_export8_thresholds(float t[257])
{
    memcpy(t, kExport8Table, sizeof(kExport8Table));
    t[256] = INFINITY;
}

static uint32_t export8_byte(uint32_t bits)
{
    float a;
    memcpy(&a, &bits, sizeof(a));
    return (uint32_t)(powf(a, 1.0f / 2.2f) * 255.0f);
}

void fr_export8_thresholds_host_powf(float t[257])
{
    t[0] = 0.0f;
    for (uint32_t b = 1; b < 256; ++b) {
        uint32_t lo = 0u, hi = 0x3F800000u;           /* byte(0.0) = 0 < b <= 255 = byte(1.0) */
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (export8_byte(mid) >= b) hi = mid; else lo = mid;
        }
        memcpy(&t[b], &hi, sizeof(float));
    }
    t[256] = INFINITY;
}

/* ---- deep-zoom reference orbit -------------------------------------------------------------------
 * DeepZoomManager::compute_reference_orbit, src/deep_zoom_system.cpp:378-424.  A single-point,
 * inherently sequential fp64 recurrence: host code in the reference and here. */
int fr_reference_orbit(double cx, double cy, int32_t max_iter, double* out_xy, int32_t* out_len)
{
    if (!out_xy || !out_len || max_iter < 1)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_reference_orbit: bad argument");
    double zr = 0.0, zi = 0.0;
    int32_t escape_iter = max_iter;
    for (int32_t i = 0; i < max_iter; ++i) {
        out_xy[2 * i] = zr;                                   /* :392 stored before the update */
        out_xy[2 * i + 1] = zi;
        const double mag = hypot(zr, zi);                     /* :396 */
        if (mag > 2.0 || mag > 1e10 || isnan(mag) || isinf(mag)) { escape_iter = i; break; }   /* :397-408 */
        const double re = zr * zr - zi * zi, im = zr * zi + zi * zr;   /* :411 complex z*z */
        zr = re + cx;
        zi = im + cy;
    }
    *out_len = escape_iter < max_iter ? escape_iter + 1 : max_iter;    /* :422-424 */
    return FR_OK;
}
