/*
 * node_client.c -- a plain C11 caller of the multi-GPU entry points (fr_node_*) through include/fractalrenderer_amd.h only.
 *   node_client lanes            n = 2, 4, 8 "devices" that are all ordinal 0: the band / strip arithmetic and the in-place
 *                                stores of every part into ONE set of whole-frame planes, bitwise against fr_render
 *   node_client rccl             the RCCL leg as far as one card can drive it: plugin load, one-rank communicator, a grouped
 *                                ncclSend / ncclRecv pair on a stream (fr_node_rccl_selftest, internal header)
 *   node_client node <n>         n DISTINCT devices 0..n-1 (needs an n-GPU box): RCCL gather (both payloads) and in-place
 *                                peer stores, every root, bitwise against fr_render on device 0
 * Exit code 0 = every check passed; otherwise the failing line is printed.
 */
#include "fractalrenderer_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int fr_node_rccl_selftest(int device, size_t bytes, int* rccl_version);   /* fractalrenderer_amd/csrc/fr_tuning.h */

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "%s:%d: check failed: %s (last error: %s)\n", __FILE__, __LINE__, #cond, fr_last_error()); return 1; } } while (0)

enum { W = 520, H = 300 };                         /* ragged: 65 sub-tile columns, 37.5 sub-tile rows */

typedef struct planes { float* rgba; double* nu; int32_t* it; } planes;

static int alloc_planes(planes* p)
{
    p->rgba = (float*)malloc((size_t)W * H * 16);
    p->nu = (double*)malloc((size_t)W * H * 8);
    p->it = (int32_t*)malloc((size_t)W * H * 4);
    return p->rgba && p->nu && p->it ? 0 : 1;
}

static int same(const planes* a, const planes* b, int with_nu, int with_it)
{
    if (memcmp(a->rgba, b->rgba, (size_t)W * H * 16) != 0) return 0;
    if (with_nu && memcmp(a->nu, b->nu, (size_t)W * H * 8) != 0) return 0;
    if (with_it && memcmp(a->it, b->it, (size_t)W * H * 4) != 0) return 0;
    return 1;
}

static void views(fr_params v[3])
{
    for (int k = 0; k < 3; ++k) fr_params_default(&v[k]);
    v[0].max_iterations = 1024;                                                   /* C2's view: tile pass + lane pool */
    v[1].max_iterations = 200; v[1].center_x = -0.743643887037151; v[1].center_y = 0.13182590420533; v[1].zoom = 0.008;
    v[2].fractal_type = FR_FRACTAL_JULIA; v[2].precision = FR_PRECISION_F32; v[2].max_iterations = 2048;   /* C3's */
    v[2].center_x = 0.0; v[2].julia_c_real = -0.8; v[2].julia_c_imag = 0.156;
}

/* every view, every root, strips and bands, host planes: the node's frame against the one context's */
static int compare_node(fr_node* node, fr_ctx* ctx, int n, int gather, int with_planes)
{
    planes want, got;
    CHECK(alloc_planes(&want) == 0 && alloc_planes(&got) == 0);
    fr_params v[3];
    views(v);
    CHECK(fr_node_set_option(node, "gather", gather) == FR_OK);
    for (int k = 0; k < 3; ++k) {
        fr_output wo = {want.rgba, want.nu, want.it, FR_MEM_HOST, FR_LAYOUT_PACKED};
        if (v[k].precision == FR_PRECISION_F32) wo.nu = NULL;                  /* (float nu plane: not compared here) */
        CHECK(fr_render(ctx, &v[k], W, H, &wo) == FR_OK);
        for (int layout = 0; layout < 2; ++layout) {
            CHECK(fr_node_set_option(node, "layout", layout) == FR_OK);
            for (int root = 0; root < n; root += (n > 2 ? n - 1 : 1)) {
                memset(got.rgba, 0, (size_t)W * H * 16); memset(got.nu, 0, (size_t)W * H * 8); memset(got.it, 0, (size_t)W * H * 4);
                fr_output go = {got.rgba, with_planes && wo.nu ? got.nu : NULL, with_planes ? got.it : NULL, FR_MEM_HOST, FR_LAYOUT_PACKED};
                CHECK(fr_node_render(node, &v[k], W, H, root, &go) == FR_OK);
                CHECK(fr_node_last_gather(node) == gather);
                CHECK(same(&want, &got, go.nu != NULL, go.iter != NULL));
            }
        }
    }
    free(want.rgba); free(want.nu); free(want.it); free(got.rgba); free(got.nu); free(got.it);
    return 0;
}

static int lanes_part(void)
{
    fr_ctx* ctx = NULL;
    CHECK(fr_ctx_create(0, &ctx) == FR_OK);
    for (int n = 2; n <= 8; n *= 2) {
        int devs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        fr_node* node = NULL;
        CHECK(fr_node_create(devs, n, &node) == FR_OK && fr_node_device_count(node) == n);
        CHECK(compare_node(node, ctx, n, FR_GATHER_PEER, 1) == 0);
        /* one ordinal listed twice cannot be two RCCL ranks: refused, with a message, and the node stays usable */
        CHECK(fr_node_set_option(node, "gather", FR_GATHER_RCCL) == FR_OK);
        fr_params p;
        fr_params_default(&p);
        static float small[64 * 64 * 4];
        fr_output so = {small, NULL, NULL, FR_MEM_HOST, FR_LAYOUT_PACKED};
        CHECK(fr_node_render(node, &p, 64, 64, 0, &so) == FR_ERR_UNSUPPORTED && strlen(fr_last_error()) > 0);
        CHECK(fr_node_set_option(node, "gather", FR_GATHER_AUTO) == FR_OK);
        CHECK(fr_node_render(node, &p, 64, 64, n - 1, &so) == FR_OK && fr_node_last_gather(node) == FR_GATHER_PEER);
        CHECK(fr_node_render(node, &p, 64, 64, n, &so) == FR_ERR_INVALID_ARG);
        /* async + wait; an option of the contexts reaches every part */
        CHECK(fr_node_set_option(node, "periodicity", -1) == FR_OK);
        CHECK(fr_node_render_async(node, &p, 64, 64, 0, &so) == FR_OK);
        CHECK(fr_node_render_async(node, &p, 64, 64, 0, &so) == FR_ERR_INVALID_ARG);     /* not waited for yet */
        CHECK(fr_node_wait(node) == FR_OK && fr_node_wait(node) == FR_OK);
        CHECK(fr_node_last_kernel_ms(node, n - 1) > 0.0f);
        fr_node_destroy(node);
    }
    fr_ctx_destroy(ctx);
    return 0;
}

static int rccl_part(void)
{
    int version = 0;
    CHECK(fr_node_rccl_selftest(0, (size_t)1 << 20, &version) == FR_OK);
    CHECK(version > 0);
    printf("rccl version %d\n", version);
    return 0;
}

static int node_part(int n)
{
    int devs[16];
    CHECK(n >= 2 && n <= 16);
    for (int k = 0; k < n; ++k) devs[k] = k;
    fr_ctx* ctx = NULL;
    CHECK(fr_ctx_create(0, &ctx) == FR_OK);
    fr_node* node = NULL;
    CHECK(fr_node_create(devs, n, &node) == FR_OK);
    CHECK(compare_node(node, ctx, n, FR_GATHER_RCCL, 0) == 0);      /* rgba only: the smooth-count payload + recolour */
    CHECK(compare_node(node, ctx, n, FR_GATHER_RCCL, 1) == 0);      /* every plane shipped */
    CHECK(compare_node(node, ctx, n, FR_GATHER_PEER, 1) == 0);
    fr_node_destroy(node);
    fr_ctx_destroy(ctx);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 2 && strcmp(argv[1], "lanes") == 0) { const int r = lanes_part(); if (!r) puts("lanes ok"); return r; }
    if (argc >= 2 && strcmp(argv[1], "rccl") == 0) { const int r = rccl_part(); if (!r) puts("rccl ok"); return r; }
    if (argc >= 3 && strcmp(argv[1], "node") == 0) { const int r = node_part(atoi(argv[2])); if (!r) puts("node ok"); return r; }
    fprintf(stderr, "usage: node_client lanes | rccl | node <n>\n");
    return 2;
}
