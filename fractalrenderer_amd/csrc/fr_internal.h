/*
 * fr_internal.h -- declarations shared by the C host side (fr_host.c, fr_franim.c)
 * and the HIP side (fr_device.hip).  Not installed; the public ABI is
 * include/fractalrenderer_amd.h.
 */
#ifndef FR_INTERNAL_H
#define FR_INTERNAL_H

#include "fractalrenderer_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* thread-local last-error message (fr_host.c) */
int fr_set_error(int status, const char* fmt, ...)
#if defined(__GNUC__)
    __attribute__((format(printf, 2, 3)))
#endif
    ;

int32_t fr_deep_zoom_reference_length(const fr_params* p);

/* the context's own stream (hipStream_t) and device ordinal: fr_node.cpp orders RCCL transfers behind the renders */
void* fr_ctx_stream_handle(fr_ctx* ctx);
int   fr_ctx_device(const fr_ctx* ctx);

/* thresholds of the 8-bit export (fr_host.c): t[b] = smallest float a in [0, 1] with (uint8)(powf(a, 1/2.2f) * 255) >= b, powf
 * being the correctly rounded single-precision power (a baked table: the same on every host); _host_powf: the same by
 * bisection with the deployment host's libm, for the tests to report where a host's powf differs */
void fr_export8_thresholds(float t[257]);
void fr_export8_thresholds_host_powf(float t[257]);

/* ---- palette knot table: what the kernels stage into LDS ------------------------------
 * Every palette of the two shaders is "warp t, then a 5-knot piece-wise linear ramp"
 * (shaders/mandelbrot.comp:60-141, shaders/julia.comp:20-181).  The table holds the ramp
 * exactly as written (break points, the per-segment multiply/divide constant, the knots),
 * so the device evaluation performs the shader's own float operations. */
enum {
    FR_WARP_NONE       = 0,   /* w = t                        */
    FR_WARP_POW        = 1,   /* w = pow(t, e)                */
    FR_WARP_SMOOTHSTEP = 2,   /* w = smoothstep(0, 1, t)      */
    FR_WARP_GRAY       = 3    /* colour = vec3(t), no ramp    */
};

typedef struct fr_palette_table {
    int32_t warp;
    float   warp_exp;
    int32_t nseg;            /* 4 or 5                                              */
    int32_t last_const;      /* 1: last segment returns knot[4] (fire-style ramps)  */
    float   seg_lo[5];       /* segment k covers w in [seg_lo[k], seg_lo[k+1])      */
    float   seg_k[5];        /* mix factor = (w - seg_lo[k]) * seg_k[k]  or / seg_k[k] */
    int32_t seg_div[5];      /* 1: divide by seg_k, 0: multiply                      */
    int32_t any_div;         /* some segment divides (lets the kernels skip the divide otherwise) */
    float   knot[6][4];      /* RGB knots (4th lane padding); knot[k], knot[k+1] bound segment k */
} fr_palette_table;

/* shader: 0 = shaders/mandelbrot.comp numbering, 1 = shaders/julia.comp numbering */
void fr_palette_table_build(int shader, int palette_mode, fr_palette_table* out);

#ifdef __cplusplus
}
#endif
#endif
