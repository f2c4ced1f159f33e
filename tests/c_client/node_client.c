/*
 * node_client.c -- a plain C11 caller of the multi-GPU entry points (fr_node_*) through include/fractalrenderer_amd.h only.
 *   node_client lanes            n = 2, 4, 8 "devices" that are all ordinal 0: the band / strip arithmetic and the in-place
 *                                stores of every part into ONE set of whole-frame planes, bitwise against fr_render
 *   node_client rccl             the RCCL leg as far as one card can drive it: plugin load, one-rank communicator, a grouped
 *                                ncclSend / ncclRecv pair on a stream (fr_node_rccl_selftest, internal header)
 *   node_client seq              a 16-frame sequence (C2- and C3-like views, zooming) on devices = {0,0,0,0}: frames in flight
 *                                (slots x lanes = 1x1, 2x2, 4x2, 8x4), rotating roots, tickets waited for in a scrambled
 *                                order, every frame bitwise against fr_render
 *   node_client failsafe         the two-phase gather and its failure paths on one card: a part that fails before its render
 *                                (every gather), a whole frame through the RCCL calls of a one-rank communicator
 *                                ("rccl_loopback"), a part that fails between the barrier and its sends (communicators
 *                                aborted, the wait returns an error, the node carries on)
 *   node_client anim <folder>    fr_node_render_animation (AnimationRenderer::start_render over a node): a 3-keyframe animation built
 *                                with fr_anim_add_keyframe, every 10th frame on devices = {0,0}, each PNG byte for byte the one
 *                                fr_render_frame_png writes; the frame callback, in order
 *   node_client node <n>         n DISTINCT devices 0..n-1 (needs an n-GPU box): RCCL gather (both payloads) and in-place
 *                                peer stores, every root, bitwise against fr_render on device 0; then a pipelined sequence
 *                                with rotating roots through both gathers
 * Exit code 0 = every check passed; otherwise the failing line is printed.
 */
#include "fractalrenderer_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int fr_node_rccl_selftest(int device, size_t bytes, int* rccl_version);   /* fractalrenderer_amd/csrc/fr_tuning.h */
int fr_node_set_tuning(fr_node* node, const char* name, int64_t value);
int fr_node_rccl_usable(const fr_node* node);

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
        CHECK(fr_node_last_kernel_ms(node, n - 1) < 0.0f);                                /* no event pair unless asked for (1.1) */
        CHECK(fr_node_set_option(node, "timing", 1) == FR_OK);
        CHECK(fr_node_render_async(node, &p, 64, 64, 0, &so) == FR_OK);
        CHECK(fr_node_render_async(node, &p, 64, 64, 0, &so) == FR_OK);                  /* a second frame in flight (1.1) */
        CHECK(fr_node_in_flight(node) == 2);
        CHECK(fr_node_set_option(node, "periodicity", 0) == FR_ERR_INVALID_ARG);         /* not while frames are in flight */
        CHECK(fr_node_wait(node) == FR_OK && fr_node_wait(node) == FR_OK && fr_node_in_flight(node) == 0);
        CHECK(fr_node_last_kernel_ms(node, n - 1) > 0.0f);
        fr_node_destroy(node);
    }
    fr_ctx_destroy(ctx);
    return 0;
}

/* ---- a sequence with frames in flight --------------------------------------------------------------------------------- */
enum { SEQ_FRAMES = 16 };

static void seq_view(int f, fr_params* p)
{
    fr_params v[3];
    views(v);
    *p = v[f % 2 ? 2 : 0];                                        /* C2's and C3's views alternate ... */
    p->zoom = 3.0 * (1.0 - 0.04 * f);                             /* ... and zoom in: no two frames are alike */
    if (f % 2 == 0) p->center_x = -0.5 - 0.01 * f;
}

static int run_sequence(fr_node* node, fr_ctx* ctx, int n, int slots, int lanes, int gather, int rotate)
{
    static float* want[SEQ_FRAMES];
    static float* got[SEQ_FRAMES];
    uint64_t ticket[SEQ_FRAMES];
    const size_t bytes = (size_t)W * H * 16;
    CHECK(fr_node_set_option(node, "slots", slots) == FR_OK && fr_node_set_option(node, "lanes", lanes) == FR_OK);
    CHECK(fr_node_set_option(node, "gather", gather) == FR_OK && fr_node_set_option(node, "layout", 0) == FR_OK);
    for (int f = 0; f < SEQ_FRAMES; ++f) {
        if (!want[f]) {
            want[f] = (float*)malloc(bytes); got[f] = (float*)malloc(bytes);
            CHECK(want[f] && got[f]);
            fr_params p;
            seq_view(f, &p);
            fr_output wo = {want[f], NULL, NULL, FR_MEM_HOST, FR_LAYOUT_PACKED};
            CHECK(fr_render(ctx, &p, W, H, &wo) == FR_OK);
        }
        memset(got[f], 0, bytes);
    }
    for (int f = 0; f < SEQ_FRAMES; ++f) {
        fr_params p;
        seq_view(f, &p);
        fr_output go = {got[f], NULL, NULL, FR_MEM_HOST, FR_LAYOUT_PACKED};
        CHECK(fr_node_submit(node, &p, W, H, rotate ? FR_ROOT_ROTATE : f % n, &go, &ticket[f]) == FR_OK);
        CHECK(ticket[f] != 0 && (f == 0 || ticket[f] == ticket[f - 1] + 1));
        CHECK(fr_node_in_flight(node) >= 1 && fr_node_in_flight(node) <= slots);
    }
    /* any order: 11 is coprime to 16 */
    for (int i = 0; i < SEQ_FRAMES; ++i) {
        const int f = (i * 11 + 5) % SEQ_FRAMES;
        CHECK(fr_node_wait_frame(node, ticket[f]) == FR_OK);
        CHECK(memcmp(want[f], got[f], bytes) == 0);
    }
    CHECK(fr_node_in_flight(node) == 0 && fr_node_wait(node) == FR_OK);
    CHECK(fr_node_wait_frame(node, ticket[SEQ_FRAMES - 1] + 1) == FR_ERR_INVALID_ARG);      /* never handed out */
    CHECK(fr_node_wait_frame(node, ticket[3]) == FR_OK);                                     /* a verdict can be asked for again */
    return 0;
}

static int seq_part(void)
{
    fr_ctx* ctx = NULL;
    CHECK(fr_ctx_create(0, &ctx) == FR_OK);
    int devs[4] = {0, 0, 0, 0};
    static const int shape[4][2] = {{1, 1}, {2, 2}, {4, 2}, {8, 4}};
    for (int n = 1; n <= 4; n += 3) {
        fr_node* node = NULL;
        CHECK(fr_node_create(devs, n, &node) == FR_OK);
        for (int k = 0; k < 4; ++k) CHECK(run_sequence(node, ctx, n, shape[k][0], shape[k][1], FR_GATHER_AUTO, k % 2) == 0);
        CHECK(fr_node_last_gather(node) == FR_GATHER_PEER);
        fr_node_destroy(node);
    }
    /* destroyed with frames still in flight: drains by itself */
    fr_node* node = NULL;
    CHECK(fr_node_create(devs, 4, &node) == FR_OK);
    static float plane[W * H * 4];
    fr_params p;
    seq_view(0, &p);
    fr_output o = {plane, NULL, NULL, FR_MEM_HOST, FR_LAYOUT_PACKED};
    CHECK(fr_node_render_async(node, &p, W, H, 1, &o) == FR_OK && fr_node_render_async(node, &p, W, H, 2, &o) == FR_OK);
    fr_node_destroy(node);
    fr_ctx_destroy(ctx);
    return 0;
}

/* ---- failure paths of the gather, as far as one card can drive them ---------------------------------------------------- */
static int failsafe_part(void)
{
    fr_ctx* ctx = NULL;
    CHECK(fr_ctx_create(0, &ctx) == FR_OK);
    planes want, got;
    CHECK(alloc_planes(&want) == 0 && alloc_planes(&got) == 0);
    fr_params v[3];
    views(v);
    fr_output wo = {want.rgba, want.nu, want.it, FR_MEM_HOST, FR_LAYOUT_PACKED};
    fr_output go = {got.rgba, got.nu, got.it, FR_MEM_HOST, FR_LAYOUT_PACKED};
    CHECK(fr_render(ctx, &v[0], W, H, &wo) == FR_OK);

    /* (1) a part fails before its render is enqueued: that frame reports it (with the part named), the frames around it
     * are untouched, the node stays usable */
    int devs[4] = {0, 0, 0, 0};
    fr_node* node = NULL;
    CHECK(fr_node_create(devs, 4, &node) == FR_OK);
    uint64_t t1 = 0, t2 = 0, t3 = 0;
    CHECK(fr_node_set_option(node, "slots", 4) == FR_OK);
    CHECK(fr_node_submit(node, &v[0], W, H, 0, &go, &t1) == FR_OK && fr_node_wait_frame(node, t1) == FR_OK);
    CHECK(same(&want, &got, 1, 1));
    CHECK(fr_node_set_tuning(node, "fail_part_phase1", 3) == FR_OK);
    CHECK(fr_node_submit(node, &v[0], W, H, 1, &go, &t2) == FR_OK);
    CHECK(fr_node_submit(node, &v[0], W, H, 2, &go, &t3) == FR_OK);
    CHECK(fr_node_wait_frame(node, t3) == FR_OK);
    CHECK(fr_node_wait_frame(node, t2) == FR_ERR_INTERNAL && strstr(fr_last_error(), "part 2") != NULL);
    CHECK(fr_node_wait(node) == FR_OK);                                   /* ... and has been told once */
    CHECK(fr_node_set_tuning(node, "fail_part_phase1", 1) == FR_OK);
    CHECK(fr_node_render_async(node, &v[0], W, H, 0, &go) == FR_OK && fr_node_wait(node) == FR_ERR_INTERNAL);
    memset(got.rgba, 0, (size_t)W * H * 16);
    CHECK(fr_node_render(node, &v[0], W, H, 3, &go) == FR_OK && same(&want, &got, 1, 1));
    fr_node_destroy(node);

    /* (2) a whole frame through the RCCL calls: one part, a one-rank communicator, the part's strips sent to itself and
     * received in place; both payloads, every view */
    int one[1] = {0};
    CHECK(fr_node_create(one, 1, &node) == FR_OK);
    CHECK(fr_node_set_tuning(node, "rccl_loopback", 1) == FR_OK && fr_node_set_option(node, "gather", FR_GATHER_RCCL) == FR_OK);
    for (int k = 0; k < 3; ++k) {
        fr_output w2 = wo, g2 = go;
        if (v[k].precision == FR_PRECISION_F32) { w2.nu = NULL; g2.nu = NULL; }
        CHECK(fr_render(ctx, &v[k], W, H, &w2) == FR_OK);
        for (int pass = 0; pass < 3; ++pass) {      /* colour only (smooth-count payload + recolour), every plane, strips of 8 rows */
            CHECK(fr_node_set_option(node, "rows_per_strip", pass == 2 ? 8 : 0) == FR_OK);
            memset(got.rgba, 0, (size_t)W * H * 16); memset(got.nu, 0, (size_t)W * H * 8); memset(got.it, 0, (size_t)W * H * 4);
            fr_output g3 = g2;
            if (pass == 0) { g3.nu = NULL; g3.iter = NULL; }
            uint64_t ta = 0, tb = 0;                 /* two frames in flight through the one communicator */
            CHECK(fr_node_submit(node, &v[k], W, H, 0, &g3, &ta) == FR_OK);
            CHECK(fr_node_last_gather(node) == FR_GATHER_RCCL);
            CHECK(fr_node_submit(node, &v[k], W, H, 0, &g3, &tb) == FR_OK);
            CHECK(fr_node_wait_frame(node, tb) == FR_OK && fr_node_wait_frame(node, ta) == FR_OK);
            CHECK(same(&want, &got, g3.nu != NULL, g3.iter != NULL));
        }
    }
    /* (3) the part fails between the barrier and its sends, its receives posted: the communicators are aborted, the wait
     * returns the error instead of hanging, and the node carries on with the in-place gather */
    CHECK(fr_render(ctx, &v[0], W, H, &wo) == FR_OK);
    CHECK(fr_node_rccl_usable(node) == 1);
    CHECK(fr_node_set_tuning(node, "fail_part_before_send", 1) == FR_OK);
    CHECK(fr_node_set_tuning(node, "rccl_timeout_ms", 5000) == FR_OK);
    CHECK(fr_node_render(node, &v[0], W, H, 0, &go) != FR_OK && strstr(fr_last_error(), "injected") != NULL);
    CHECK(fr_node_rccl_usable(node) == 0);
    memset(got.rgba, 0, (size_t)W * H * 16);
    CHECK(fr_node_render(node, &v[0], W, H, 0, &go) == FR_OK && fr_node_last_gather(node) == FR_GATHER_PEER);
    CHECK(same(&want, &got, 1, 1));
    fr_node_destroy(node);
    fr_ctx_destroy(ctx);
    return 0;
}

/* ---- the reference's animation loop, from C ------------------------------------------------------------------------------ */
static int seen_frames[64], n_seen = 0;
static int on_frame(int32_t frame, int32_t total, void* user) { (void)total; (void)user; if (n_seen < 64) seen_frames[n_seen++] = frame; return 0; }

static int same_file(const char* a, const char* b)
{
    FILE* fa = fopen(a, "rb"); FILE* fb = fopen(b, "rb");
    if (!fa || !fb) { if (fa) fclose(fa); if (fb) fclose(fb); return 0; }
    int same = 1, ca, cb;
    do { ca = fgetc(fa); cb = fgetc(fb); if (ca != cb) same = 0; } while (same && ca != EOF);
    fclose(fa); fclose(fb);
    return same;
}

static int anim_part(const char* folder)
{
    fr_anim* anim = NULL;
    CHECK(fr_anim_create(&anim) == FR_OK);
    fr_params k;
    fr_params_default(&k);
    k.max_iterations = 300;
    CHECK(fr_anim_add_keyframe(anim, 0.0f, &k, FR_INTERP_LINEAR) == FR_OK);
    k.center_x = -0.743643887037151; k.center_y = 0.13182590420533; k.zoom = 0.05; k.max_iterations = 900; k.palette_mode = 3;
    CHECK(fr_anim_add_keyframe(anim, 1.0f, &k, FR_INTERP_EASE_IN_OUT) == FR_OK);
    k.zoom = 0.002; k.max_iterations = 1500;
    CHECK(fr_anim_add_keyframe(anim, 2.0f, &k, FR_INTERP_EXPONENTIAL) == FR_OK);
    fr_params base;
    fr_params_default(&base);
    base.precision = FR_PRECISION_F64;
    int devs[2] = {0, 0};
    fr_node* node = NULL;
    CHECK(fr_node_create(devs, 2, &node) == FR_OK);
    fr_anim_render_options o;
    memset(&o, 0, sizeof o);
    o.width = 240; o.height = 160; o.frame_step = 10; o.on_frame_complete = on_frame;
    int32_t wrote = 0;
    const int32_t total = fr_anim_frame_count(anim);
    CHECK(total > 20);
    CHECK(fr_node_render_animation(node, anim, &base, &o, folder, &wrote) == FR_OK);
    CHECK(wrote == (total + 9) / 10 && n_seen == wrote);
    fr_ctx* ctx = NULL;
    CHECK(fr_ctx_create(0, &ctx) == FR_OK);
    for (int i = 0; i < wrote; ++i) {
        CHECK(seen_frames[i] == 10 * i);
        fr_params p;
        CHECK(fr_anim_state_at(anim, fr_anim_frame_time(anim, 10 * i), &base, &p) == FR_OK);
        char got[4096], want[4096];
        CHECK(fr_frame_path(folder, 10 * i, got, sizeof got) == FR_OK);
        snprintf(want, sizeof want, "%s/reference_%d.png", folder, i);
        CHECK(fr_render_frame_png(ctx, &p, 240, 160, want) == FR_OK);
        CHECK(same_file(got, want));
    }
    fr_ctx_destroy(ctx);
    fr_node_destroy(node);
    fr_anim_free(anim);
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
    CHECK(fr_node_set_option(node, "payload", 0) == FR_OK);
    CHECK(run_sequence(node, ctx, n, 2, 2, FR_GATHER_RCCL, 1) == 0 && fr_node_last_gather(node) == FR_GATHER_RCCL);
    CHECK(run_sequence(node, ctx, n, 4, 2, FR_GATHER_PEER, 1) == 0);
    CHECK(run_sequence(node, ctx, n, 4, 4, FR_GATHER_AUTO, 0) == 0);
    fr_node_destroy(node);
    fr_ctx_destroy(ctx);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 2 && strcmp(argv[1], "lanes") == 0) { const int r = lanes_part(); if (!r) puts("lanes ok"); return r; }
    if (argc >= 2 && strcmp(argv[1], "seq") == 0) { const int r = seq_part(); if (!r) puts("seq ok"); return r; }
    if (argc >= 2 && strcmp(argv[1], "failsafe") == 0) { const int r = failsafe_part(); if (!r) puts("failsafe ok"); return r; }
    if (argc >= 3 && strcmp(argv[1], "anim") == 0) { const int r = anim_part(argv[2]); if (!r) puts("anim ok"); return r; }
    if (argc >= 2 && strcmp(argv[1], "rccl") == 0) { const int r = rccl_part(); if (!r) puts("rccl ok"); return r; }
    if (argc >= 3 && strcmp(argv[1], "node") == 0) { const int r = node_part(atoi(argv[2])); if (!r) puts("node ok"); return r; }
    fprintf(stderr, "usage: node_client lanes | seq | failsafe | anim <folder> | rccl | node <n>\n");
    return 2;
}
