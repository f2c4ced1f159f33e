/*
 * client.c -- a plain C11 caller of libfractalrenderer_amd.so through include/fractalrenderer_amd.h only
 * (no Python, no torch): what a maintainer's binding sees.  Built and run by the tests:
 *   client host <franim> <out.png>   host-side entry points only (no GPU needed)
 *   client gpu  <out.png>            the render path through FR_MEM_HOST buffers
 * Exit code 0 = every check passed; otherwise the failing line is printed.
 */
#include "fractalrenderer_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "%s:%d: check failed: %s (last error: %s)\n", __FILE__, __LINE__, #cond, fr_last_error()); return 1; } } while (0)

static int host_part(const char* franim, const char* png)
{
    fr_params p;
    fr_params_default(&p);
    CHECK(p.center_x == -0.5 && p.center_y == 0.0 && p.zoom == 3.0 && p.max_iterations == 256 && p.bailout == 4.0f);
    CHECK(fr_params_validate(&p, 64, 64) == FR_OK);
    p.max_iterations = 0;
    CHECK(fr_params_validate(&p, 64, 64) == FR_ERR_INVALID_ARG && strlen(fr_last_error()) > 0);
    fr_params_default(&p);
    float pc[20];
    CHECK(fr_pack_push_constants(&p, pc) == FR_OK && pc[0] == -0.5f && pc[2] == 3.0f && pc[3] == 256.0f && pc[6] == 4.0f);

    fr_shard sh = {1u, 3u, 4u};
    CHECK(fr_shard_rows(&sh, 30) == 10u);                       /* strips 1, 4, 7 of 8 (the 8th is ragged: 2 rows to part 1) */
    CHECK(fr_shard_global_row(&sh, 30, 0) == 4u && fr_shard_global_row(&sh, 30, 4) == 16u);

    fr_anim* a = NULL;
    CHECK(fr_anim_load(franim, &a) == FR_OK);
    fr_anim_info info;
    CHECK(fr_anim_get_info(a, &info) == FR_OK && info.keyframe_count >= 2 && info.target_fps > 0);
    CHECK(fr_anim_frame_count(a) == (int32_t)(info.duration * (float)info.target_fps));
    fr_params base, at;
    fr_params_default(&base);
    CHECK(fr_anim_state_at(a, 0.0f, &base, &at) == FR_OK && at.zoom > 0.0);
    fr_anim_free(a);

    static uint8_t img[16 * 8 * 3];
    for (int i = 0; i < (int)sizeof(img); ++i) img[i] = (uint8_t)(i * 7);
    CHECK(fr_write_png(png, 16, 8, 8, img, NULL, 0, 0) == FR_OK);
    FILE* f = fopen(png, "rb");
    CHECK(f != NULL);
    unsigned char sig[8];
    CHECK(fread(sig, 1, 8, f) == 8 && sig[1] == 'P' && sig[2] == 'N' && sig[3] == 'G');
    fclose(f);
    char path[64];
    CHECK(fr_frame_path("out", 42, path, sizeof(path)) == FR_OK && strcmp(path, "out/frame_000042.png") == 0);
    int major = -1, minor = -1;
    fr_version(&major, &minor);
    CHECK(strcmp(fr_status_string(FR_OK), fr_status_string(FR_ERR_HIP)) != 0 && major >= 0 && minor >= 0);
    return 0;
}

static int gpu_part(const char* png)
{
    fr_ctx* ctx = NULL;
    CHECK(fr_ctx_create(0, &ctx) == FR_OK && fr_ctx_compute_units(ctx) > 0);
    enum { W = 96, H = 64 };
    static float rgba[W * H * 4];
    static double nu[W * H];
    static int32_t it[W * H];
    fr_params p;
    fr_params_default(&p);
    p.max_iterations = 300;
    fr_output out = {rgba, nu, it, FR_MEM_HOST, FR_LAYOUT_PACKED};
    CHECK(fr_render(ctx, &p, W, H, &out) == FR_OK);
    /* pixel (W/2, H/2) maps exactly to the centre (-0.5, 0): inside the main cardioid */
    CHECK(it[(H / 2) * W + W / 2] == 300 && nu[(H / 2) * W + W / 2] == 300.0);
    /* pixel (0, 0): c = -0.5 + (0 - W/2)/H*3 - 1.5i, iterated here as the shader writes it
     * (shaders/mandelbrot.comp:149-177): update, then test |z|^2 > bailout^2 = 16 */
    {
        const double cx = -0.5 + (0.0 - 0.5 * W) / H * 3.0, cy = 0.0 + (0.0 - 0.5 * H) / H * 3.0;
        double zx = 0.0, zy = 0.0;
        int i = 0;
        for (; i < 300; ++i) {
            const double x = zx * zx - zy * zy + cx, y = 2.0 * zx * zy + cy;
            zx = x; zy = y;
            if (zx * zx + zy * zy > 16.0) break;
        }
        const double expect = (double)i + 1.0 - log2(0.5 * log2(zx * zx + zy * zy));
        CHECK(i < 300 && it[0] == i && fabs(nu[0] - expect) < 1e-12);
    }
    for (int i = 0; i < W * H; ++i) CHECK(rgba[4 * i + 3] == 1.0f);
    /* conjugate symmetry: rows y and H - y */
    for (int y = 1; y < H / 2; ++y)
        for (int x = 0; x < W; ++x) CHECK(it[y * W + x] == it[(H - y) * W + x]);
    CHECK(fr_ctx_last_kernel_ms(ctx) < 0.0f);                      /* no event pair around a render unless asked for (1.1) */
    CHECK(fr_ctx_set_option(ctx, "timing", 1) == FR_OK);
    CHECK(fr_render(ctx, &p, W, H, &out) == FR_OK && fr_ctx_last_kernel_ms(ctx) > 0.0f);

    /* a part of a sharded frame equals the rows of the whole frame */
    fr_shard sh = {1u, 2u, 8u};
    static int32_t part[W * H];
    fr_output po = {NULL, NULL, part, FR_MEM_HOST, FR_LAYOUT_PACKED};
    CHECK(fr_render_shard(ctx, &p, W, H, &sh, &po) == FR_OK);
    const uint32_t rows = fr_shard_rows(&sh, H);
    for (uint32_t r = 0; r < rows; ++r)
        CHECK(memcmp(part + (size_t)r * W, it + (size_t)fr_shard_global_row(&sh, H, r) * W, W * sizeof(int32_t)) == 0);

    static uint8_t rgb8[W * H * 3];
    CHECK(fr_export_rgb8(ctx, rgba, W, H, rgb8, FR_MEM_HOST, 1) == FR_OK);
    CHECK(fr_write_png(png, W, H, 8, rgb8, NULL, 0, 0) == FR_OK);
    CHECK(fr_render_frame_png(ctx, &p, W, H, png) == FR_OK);

    p.fractal_type = FR_FRACTAL_PHOENIX;
    CHECK(fr_render(ctx, &p, W, H, &out) == FR_ERR_UNSUPPORTED);
    fr_ctx_destroy(ctx);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 4 && strcmp(argv[1], "host") == 0) { const int r = host_part(argv[2], argv[3]); if (!r) puts("host ok"); return r; }
    if (argc >= 3 && strcmp(argv[1], "gpu") == 0) { const int r = gpu_part(argv[2]); if (!r) puts("gpu ok"); return r; }
    fprintf(stderr, "usage: client host <franim> <out.png> | client gpu <out.png>\n");
    return 2;
}
