/*
 * fr_oracle.h -- CPU restatement of the reference's escape-time hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY PIN: franklynch/FractalRenderer ships no tests, golden images or
 * known-answer vectors for this path, and its only implementation of it is
 * GLSL/Vulkan (fp32), which no driver in this image can run.  It does ship its
 * shaders COMPILED (shaders/<name>.comp.spv), so the fp32 variants are pinned against
 * OUTPUTS OF THOSE BINARIES: tests/golden/make_spv_golden.py executes them,
 * one invocation per pixel, through the SPIR-V interpreter
 * tests/golden/spirv_interp.py (ours; IEEE binary32, one rounding per
 * instruction) and commits what they write as tests/golden/spv_frames.npz
 * (70 cases over mandelbrot / julia / burning_ship / test_deep_zoom);
 * tests/test_spv_golden.py requires this restatement to reproduce them --
 * escape indices bit-exact, written texels within 5e-6.  The fp64 variants
 * have no counterpart in the reference (it has no fp64 shader): they are the
 * same source compiled with REAL=double and are additionally pinned by
 *   (i)  analytic known answers (tests/test_oracle_kat.py),
 *   (ii) an independent numpy restatement (oracle/np_restatement.py) and an
 *        mpmath high-precision check of nu, and
 *   (iii) the reference's own data artefact FR/.franim for the animation rows.
 * What is NOT claimed: that a Vulkan driver's code generation (fma
 * contraction, fast transcendental approximations) yields the same bits as
 * the interpreter; that latitude is what the colour tolerances are for.
 *
 * Path shorthand: shaders/ = /root/reference/FractalRenderer/shaders/,
 *                 src/     = /root/reference/FractalRenderer/src/.
 *
 * Every per-pixel function evaluates the shader formulas in the as-written
 * operation order with NO fma contraction (build with -ffp-contract=off):
 *     x  = (zx*zx - zy*zy) + cx
 *     y  = ((2*zx)*zy) + cy
 *     r2 = x*x + y*y
 */
#ifndef FR_ORACLE_H
#define FR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Plain mirror of the hot-path fields of FractalState (src/fractal_state.h:16-91).
 * Deliberately NOT the product's fr_params: the oracle shares no header with
 * the thing it checks. */
typedef struct fro_params {
    int32_t fractal;        /* 0 Mandelbrot, 1 Julia, 2 Burning Ship, 5 Deep_Zoom (FractalType, src/fractal_state.h:6-14) */
    int32_t precision;      /* 0 = fp32 (what the shaders do), 1 = fp64 */
    double  center_x, center_y, zoom;
    int32_t max_iterations;
    float   bailout;        /* compared as |z|^2 > bailout^2 */
    double  julia_c_real, julia_c_imag;   /* float in FractalState; double is a superset (fp32 path narrows) */
    int32_t aa;             /* antialiasing_samples */
    int32_t palette_mode;
    float   color_offset, color_scale;
    int32_t interior_style;
    int32_t orbit_trap_enabled;
    float   orbit_trap_radius;
    int32_t stripe_enabled;
    float   stripe_density;
    float   brightness, saturation, contrast;
    int32_t post_chain;     /* 0: linear colour, 1: enhance->ACES->gamma (shaders/mandelbrot.comp:233-235) */
    int32_t use_perturbation;   /* Deep_Zoom (fractal 5): 1 = reference orbit computed and used, 0 = empty orbit */
} fro_params;

/* Render rows [y0, y1) of a W x H frame.  Any output pointer may be NULL.
 *   rgba : (y1-y0)*W*4 float   row-major, row 0 = y0, alpha = 1
 *   nu   : (y1-y0)*W   double  smooth iteration count of sample (0,0); max_iter for interior
 *                               (fp32 mode: the float value widened exactly)
 *   iter : (y1-y0)*W   int32   index i of the escaping update; max_iter for interior
 *   zre, zim : z at escape (or final z for interior), widened to double
 * threads <= 0 -> use all OpenMP threads; 1 -> serial.
 * Returns the exact number of executed iterations (sum over samples of
 * i+1 for escaped, max_iter for interior). */
int64_t fro_render_rows(const fro_params* p, int32_t W, int32_t H,
                        int32_t y0, int32_t y1,
                        float* rgba, double* nu, int32_t* iter,
                        double* zre, double* zim, int32_t threads);

/* Palette functions exactly as the two shaders define them.
 * shader = 0: shaders/mandelbrot.comp:60-141, 1: shaders/julia.comp:20-181. */
void fro_palette(int32_t shader, int32_t mode, float t, float rgb[3]);

/* enhance_color -> aces_tonemap -> pow(1/2.2)  (shaders/mandelbrot.comp:38-54,233-235).
 * julia_floors != 0 applies the Julia shader's max() floors (shaders/julia.comp:319-322). */
void fro_post_chain(float rgb[3], float brightness, float saturation, float contrast,
                    int32_t julia_floors);

/* ComputeEffect::update_from_state push-constant packing (src/compute_effect_manager.h:84-140). */
void fro_pack_push_constants(const fro_params* p, float out[20]);

/* DeepZoomManager::compute_reference_orbit fp64 loop (src/deep_zoom_system.cpp:378-424).
 * Writes up to max_iter points (re,im interleaved) and returns the trimmed orbit length. */
int32_t fro_reference_orbit(double cx, double cy, int32_t max_iter, double* out_xy);

/* second tonemap + gamma + u8 truncation + vertical flip of the 8-bit export
 * (src/vk_engine.cpp:1344-1371) applied to a float RGBA image (the fp16 round
 * trip of the reference's storage image is applied when through_half != 0). */
void fro_export_rgb8(const float* rgba, int32_t W, int32_t H, uint8_t* rgb8,
                     int32_t through_half);

/* byte of the 8-bit export scanned over every float of [0, 1]: first[b] = smallest bit pattern with byte b;
 * returns the number of monotonicity violations (see fr_oracle.c) */
int64_t fro_export8_scan(uint32_t first[256]);

/* Colour stage alone: nu (as fro_render_rows returns it: doubles, fp32 values widened exactly) -> RGBA f32,
 * for the plain colourings whose colour is a function of nu (see include/fractalrenderer_amd.h,
 * fr_colorize_supported).  Applies the post chain when p->post_chain. */
void fro_colorize(const fro_params* p, int64_t n, const double* nu, float* rgba);

int32_t fro_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
