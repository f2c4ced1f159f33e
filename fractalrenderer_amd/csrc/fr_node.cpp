/*
 * fr_node.cpp -- one frame over the GPUs of a node, behind the C ABI (fr_node_* in include/fractalrenderer_amd.h).
 *
 * BASELINE.json's north star keeps the host in C and tiles the image "across the 8 GPUs of one node as disjoint row bands
 * with a final RCCL gather over xGMI".  The reference has nothing to restate here -- it renders on the one GPU it picked
 * (src/vk_engine.cpp:608) -- so this is the MI355X-side design of the same render(viewport, max_iter, out_buffer) surface
 * (src/animation_renderer.h:41-48) for a caller that owns several devices:
 *
 *   - ONE process; per device one render context (fr_ctx: its own stream, scratch and control block) and one host WORKER
 *     THREAD bound to that device.  A part of a C2 frame is ~0.1 ms of GPU work behind three launches; enqueued from one
 *     thread, eight devices' worth of launches would take longer than the kernels run.
 *   - the path shards with no exchange during compute: part k renders strips k, k + n, ... of the frame (fr_shard).
 *   - the gather is the only transfer, and there are two forms of it (fr_gather): the kernels' stores go straight into
 *     the root's planes through the peer mapping (FR_LAYOUT_FRAME: every part addresses whole-frame planes), or the
 *     parts ship their strips with grouped ncclSend / ncclRecv (fr_rccl_plugin.cpp), received in place: a strip is one
 *     contiguous byte range both in the part's packed buffer and in the frame.
 *
 * The arithmetic of every part is fr_render_shard_async's, so the frame is byte-identical to fr_render's.
 */
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "fr_internal.h"
#include "fr_tuning.h"

namespace {

constexpr int kMaxParts = 16;

struct RcclApi {
    void* handle = nullptr;
    int (*init)(const int*, int, void**, char*, size_t) = nullptr;
    void (*destroy)(void**, int) = nullptr;
    int (*group_start)(void) = nullptr;
    int (*group_end)(char*, size_t) = nullptr;
    int (*send)(void*, const void*, size_t, int, void*, char*, size_t) = nullptr;
    int (*recv)(void*, void*, size_t, int, void*, char*, size_t) = nullptr;
    int (*version)(void) = nullptr;
};

/* one host thread per device: runs the jobs handed to it, in order */
struct Worker {
    std::thread thread;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int status = FR_OK;
    char err[512] = {0};

    void loop()
    {
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit) return;
                j = std::move(job);
                has_job = false;
            }
            const int st = j();
            {
                std::lock_guard<std::mutex> lk(m);
                status = st;
                if (st != FR_OK) snprintf(err, sizeof err, "%s", fr_last_error());   /* the thread-local message of THIS thread */
                done = true;
            }
            cv.notify_all();
        }
    }
    void post(std::function<int()> j)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return status;
    }
};

struct Planes { float* rgba; void* nu; int32_t* iter; };

}  // namespace

struct fr_node {
    int n = 0;
    int devices[kMaxParts] = {0};
    bool distinct = false;                /* no ordinal occurs twice */
    bool peer_ok = false;                 /* every pair of different ordinals can map each other's memory */
    fr_ctx* ctx[kMaxParts] = {nullptr};
    Worker* workers[kMaxParts] = {nullptr};
    /* options */
    int gather = FR_GATHER_AUTO, layout = 0, payload = 0;
    uint32_t rows_per_strip = 0;
    /* RCCL leg */
    RcclApi rccl;
    void* comms[kMaxParts] = {nullptr};
    /* per-part packed staging on the part's own device (RCCL gather), grow-only */
    void* stage[kMaxParts] = {nullptr};
    size_t stage_bytes[kMaxParts] = {0};
    /* per-device whole-frame staging: FR_MEM_HOST outputs and the nu frame of the nu payload, grow-only */
    void* frame_buf[kMaxParts] = {nullptr};
    size_t frame_bytes[kMaxParts] = {0};
    /* the render in flight */
    bool in_flight = false;
    int last_gather = -1;
    int root = 0;
    bool host_out = false;
    fr_output user_out = {nullptr, nullptr, nullptr, 0, 0};
    Planes dev_frame = {nullptr, nullptr, nullptr};
    size_t npx = 0, nu_elt = 0;
};

namespace {

#define NODE_HIP_TRY(expr)                                                             \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fr_set_error(FR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

/* run fn(k) on worker k for every part, wait for all; the first failure's status and message are the call's */
int run_all(fr_node* nd, const std::function<int(int)>& fn)
{
    for (int k = 0; k < nd->n; ++k) nd->workers[k]->post([&fn, k] { return fn(k); });
    int st = FR_OK;
    for (int k = 0; k < nd->n; ++k) {
        const int s = nd->workers[k]->wait();
        if (s != FR_OK && st == FR_OK) {
            st = s;
            fr_set_error(s, "part %d (device %d): %s", k, nd->devices[k], nd->workers[k]->err);
        }
    }
    return st;
}

int grow(void** buf, size_t* have, size_t need)
{
    if (need <= *have) return FR_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *have = 0; }
    NODE_HIP_TRY(hipMalloc(buf, need));
    *have = need;
    return FR_OK;
}

/* libfractalrenderer_amd_rccl.so from the directory this library was loaded from */
int load_rccl(fr_node* nd)
{
    if (nd->rccl.handle) return FR_OK;
    Dl_info info;
    char path[4096];
    if (dladdr((void*)&fr_node_create, &info) && info.dli_fname && strrchr(info.dli_fname, '/')) {
        const size_t dir = (size_t)(strrchr(info.dli_fname, '/') - info.dli_fname) + 1;
        snprintf(path, sizeof path, "%.*slibfractalrenderer_amd_rccl.so", (int)dir, info.dli_fname);
    } else {
        snprintf(path, sizeof path, "libfractalrenderer_amd_rccl.so");
    }
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fr_set_error(FR_ERR_UNSUPPORTED, "the RCCL gather needs %s: %s", path, dlerror());
    RcclApi& r = nd->rccl;
    r.init = (decltype(r.init))dlsym(h, "fr_rccl_init");
    r.destroy = (decltype(r.destroy))dlsym(h, "fr_rccl_destroy");
    r.group_start = (decltype(r.group_start))dlsym(h, "fr_rccl_group_start");
    r.group_end = (decltype(r.group_end))dlsym(h, "fr_rccl_group_end");
    r.send = (decltype(r.send))dlsym(h, "fr_rccl_send");
    r.recv = (decltype(r.recv))dlsym(h, "fr_rccl_recv");
    r.version = (decltype(r.version))dlsym(h, "fr_rccl_version");
    if (!r.init || !r.destroy || !r.group_start || !r.group_end || !r.send || !r.recv || !r.version) {
        dlclose(h);
        return fr_set_error(FR_ERR_UNSUPPORTED, "%s lacks an entry point", path);
    }
    r.handle = h;
    return FR_OK;
}

int ensure_comms(fr_node* nd)
{
    if (nd->comms[0]) return FR_OK;
    if (!nd->distinct)
        return fr_set_error(FR_ERR_UNSUPPORTED, "the RCCL gather needs distinct devices (one communicator rank per device); "
                                                "this node lists a device twice: use the in-place gather (\"gather\" = 1)");
    int st = load_rccl(nd);
    if (st != FR_OK) return st;
    char err[256] = {0};
    if (nd->rccl.init(nd->devices, nd->n, nd->comms, err, sizeof err) != 0)
        return fr_set_error(FR_ERR_HIP, "%s", err);
    return FR_OK;
}

/* strips: 32 rows (whole 8x8 sub-tile rows: the lean tile kernel applies) dealt round-robin; bands: one contiguous
 * band per part, a whole number of sub-tile rows high, the last one short */
uint32_t strip_rows(const fr_node* nd, uint32_t H)
{
    const uint32_t n = (uint32_t)nd->n;
    if (nd->layout == 1) {
        const uint32_t band = (H + n - 1) / n;
        return (band + 7u) / 8u * 8u;
    }
    if (nd->rows_per_strip) return nd->rows_per_strip;
    uint32_t R = 32;
    while (R > 8 && (uint64_t)R * n > H) R >>= 1;          /* small frames: every part still gets rows */
    return R;
}

}  // namespace

extern "C" int fr_node_create(const int* devices, int n, fr_node** out)
{
    if (!out) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > kMaxParts)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_create: 1..%d devices", kMaxParts);
    fr_node* nd = new (std::nothrow) fr_node();
    if (!nd) return fr_set_error(FR_ERR_NOMEM, "out of host memory");
    nd->n = n;
    nd->distinct = true;
    for (int k = 0; k < n; ++k) {
        nd->devices[k] = devices[k];
        for (int j = 0; j < k; ++j) if (devices[j] == devices[k]) nd->distinct = false;
    }
    for (int k = 0; k < n; ++k) {
        const int st = fr_ctx_create(devices[k], &nd->ctx[k]);
        if (st != FR_OK) { fr_node_destroy(nd); return st; }
    }
    /* peer mappings between every pair of different ordinals (any device may be a frame's root) */
    nd->peer_ok = true;
    for (int k = 0; k < n && nd->peer_ok; ++k) {
        for (int j = 0; j < n; ++j) {
            if (devices[j] == devices[k]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[k], devices[j]) != hipSuccess || !can) { nd->peer_ok = false; break; }
            if (hipSetDevice(devices[k]) != hipSuccess) { nd->peer_ok = false; break; }
            const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (e != hipSuccess) { nd->peer_ok = false; break; }
        }
    }
    for (int k = 0; k < n; ++k) {
        nd->workers[k] = new (std::nothrow) Worker();
        if (!nd->workers[k]) { fr_node_destroy(nd); return fr_set_error(FR_ERR_NOMEM, "out of host memory"); }
        Worker* w = nd->workers[k];
        const int dev = devices[k];
        w->thread = std::thread([w, dev] { (void)hipSetDevice(dev); w->loop(); });
    }
    *out = nd;
    return FR_OK;
}

extern "C" void fr_node_destroy(fr_node* nd)
{
    if (!nd) return;
    if (nd->in_flight) (void)fr_node_wait(nd);
    for (int k = 0; k < nd->n; ++k) {
        Worker* w = nd->workers[k];
        if (!w) continue;
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
        w->cv.notify_all();
        if (w->thread.joinable()) w->thread.join();
        delete w;
    }
    if (nd->rccl.handle) {
        if (nd->comms[0]) nd->rccl.destroy(nd->comms, nd->n);
        dlclose(nd->rccl.handle);
    }
    for (int k = 0; k < nd->n; ++k) {
        if (nd->stage[k] || nd->frame_buf[k]) {
            (void)hipSetDevice(nd->devices[k]);
            if (nd->stage[k]) (void)hipFree(nd->stage[k]);
            if (nd->frame_buf[k]) (void)hipFree(nd->frame_buf[k]);
        }
        if (nd->ctx[k]) fr_ctx_destroy(nd->ctx[k]);
    }
    delete nd;
}

extern "C" int fr_node_device_count(const fr_node* nd) { return nd ? nd->n : fr_set_error(FR_ERR_INVALID_ARG, "node is NULL"); }

extern "C" int fr_node_last_gather(const fr_node* nd) { return nd ? nd->last_gather : fr_set_error(FR_ERR_INVALID_ARG, "node is NULL"); }

extern "C" float fr_node_last_kernel_ms(fr_node* nd, int part)
{
    if (!nd || part < 0 || part >= nd->n) return -1.0f;
    return fr_ctx_last_kernel_ms(nd->ctx[part]);
}

extern "C" int fr_node_set_option(fr_node* nd, const char* name, int64_t value)
{
    if (!nd || !name) return fr_set_error(FR_ERR_INVALID_ARG, "node/name is NULL");
    if (nd->in_flight) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_set_option: a render is in flight (fr_node_wait first)");
    if (!strcmp(name, "gather")) {
        if (value < 0 || value > 2) return fr_set_error(FR_ERR_INVALID_ARG, "gather must be 0 (automatic), 1 (in-place peer stores) or 2 (RCCL)");
        nd->gather = (int)value;
    } else if (!strcmp(name, "layout")) {
        if (value < 0 || value > 1) return fr_set_error(FR_ERR_INVALID_ARG, "layout must be 0 (interleaved strips) or 1 (contiguous bands)");
        nd->layout = (int)value;
    } else if (!strcmp(name, "rows_per_strip")) {
        if (value < 0 || value > (1 << 20)) return fr_set_error(FR_ERR_INVALID_ARG, "rows_per_strip out of range");
        nd->rows_per_strip = (uint32_t)value;
    } else if (!strcmp(name, "payload")) {
        if (value < 0 || value > 1) return fr_set_error(FR_ERR_INVALID_ARG, "payload must be 0 (automatic) or 1 (the planes asked for)");
        nd->payload = (int)value;
    } else {
        for (int k = 0; k < nd->n; ++k) {
            const int st = fr_ctx_set_option(nd->ctx[k], name, value);
            if (st != FR_OK) return st;
        }
    }
    return FR_OK;
}

extern "C" int fr_node_render_async(fr_node* nd, const fr_params* p, uint32_t W, uint32_t H, int root, const fr_output* out)
{
    if (!nd || !p || !out) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_render: NULL argument");
    if (nd->in_flight) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_render: the previous render has not been waited for (fr_node_wait)");
    if (root < 0 || root >= nd->n) return fr_set_error(FR_ERR_INVALID_ARG, "root %d outside [0, %d)", root, nd->n);
    int st = fr_params_validate(p, W, H);
    if (st != FR_OK) return st;
    if (!out->rgba && !out->nu && !out->iter) return fr_set_error(FR_ERR_INVALID_ARG, "fr_output has no plane to write");
    if (out->memory != FR_MEM_DEVICE && out->memory != FR_MEM_HOST)
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown fr_output.memory %d", out->memory);
    if (out->layout != FR_LAYOUT_PACKED)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_render assembles whole frames: fr_output.layout must be 0");

    int gather = nd->gather;
    if (gather == FR_GATHER_AUTO) gather = nd->distinct && nd->n > 1 ? FR_GATHER_RCCL : FR_GATHER_PEER;
    if (gather == FR_GATHER_PEER && !nd->peer_ok)
        return fr_set_error(FR_ERR_UNSUPPORTED, "in-place gather: the devices of this node cannot map each other's memory");
    if (gather == FR_GATHER_RCCL && nd->n > 1) {
        st = ensure_comms(nd);
        if (st != FR_OK) return st;
    }

    const uint32_t R = strip_rows(nd, H);
    const size_t npx = (size_t)W * H;
    const size_t nu_elt = (p->precision == FR_PRECISION_F64 && p->fractal_type != FR_FRACTAL_DEEP_ZOOM) ? 8 : 4;
    const bool host_out = out->memory == FR_MEM_HOST;
    /* nu payload (RCCL gather): parts render and ship only the smooth-count plane, the root recolours */
    const bool nu_payload = gather == FR_GATHER_RCCL && nd->n > 1 && nd->payload == 0 && out->rgba && !out->iter &&
                            fr_colorize_supported(p) == 1;
    const bool need_nu_frame = nu_payload && !out->nu;

    /* whole-frame planes on the root device: the caller's, or (host outputs / the nu frame of the nu payload) the node's */
    Planes frame = {out->rgba, out->nu, out->iter};
    if (host_out || need_nu_frame) {
        const size_t off_nu = host_out && out->rgba ? npx * 16 : 0;
        const size_t off_iter = off_nu + ((host_out && out->nu) || need_nu_frame ? npx * 8 : 0);
        const size_t need = off_iter + (host_out && out->iter ? npx * 4 : 0);
        NODE_HIP_TRY(hipSetDevice(nd->devices[root]));
        (void)fr_ctx_synchronize(nd->ctx[root]);                  /* growing frees the old buffer */
        st = grow(&nd->frame_buf[root], &nd->frame_bytes[root], need);
        if (st != FR_OK) return st;
        char* base = (char*)nd->frame_buf[root];
        if (host_out) {
            frame.rgba = out->rgba ? (float*)base : nullptr;
            frame.nu = out->nu || need_nu_frame ? (void*)(base + off_nu) : nullptr;
            frame.iter = out->iter ? (int32_t*)(base + off_iter) : nullptr;
        } else {
            frame.nu = (void*)(base + off_nu);
        }
    }

    nd->root = root; nd->host_out = host_out; nd->user_out = *out; nd->dev_frame = frame; nd->npx = npx; nd->nu_elt = nu_elt;
    nd->last_gather = gather;

    const fr_params params = *p;
    const int n = nd->n;
    st = run_all(nd, [&](int k) -> int {
        const fr_shard sh = {(uint32_t)k, (uint32_t)n, R};
        const uint32_t rows = fr_shard_rows(&sh, H);
        fr_ctx* c = nd->ctx[k];
        if (gather == FR_GATHER_PEER || n == 1) {
            if (rows == 0) return FR_OK;
            const fr_output o = {frame.rgba, frame.nu, frame.iter, FR_MEM_DEVICE, FR_LAYOUT_FRAME};
            return fr_render_shard_async(c, &params, W, H, &sh, &o, nullptr);
        }
        /* ---- RCCL gather ---- */
        const bool ship_rgba = frame.rgba && !nu_payload, ship_nu = frame.nu != nullptr, ship_iter = frame.iter != nullptr;
        void* stream = fr_ctx_stream_handle(c);
        char err[256] = {0};
        if (k == root) {
            if (rows) {
                const fr_output o = {ship_rgba ? frame.rgba : nullptr, frame.nu, frame.iter, FR_MEM_DEVICE, FR_LAYOUT_FRAME};
                const int s = fr_render_shard_async(c, &params, W, H, &sh, &o, nullptr);
                if (s != FR_OK) return s;
            }
            if (nd->rccl.group_start() != 0) return fr_set_error(FR_ERR_HIP, "ncclGroupStart failed");
            int bad = 0;
            for (int src = 0; src < n && !bad; ++src) {
                if (src == root) continue;
                const fr_shard ss = {(uint32_t)src, (uint32_t)n, R};
                const uint32_t srows = fr_shard_rows(&ss, H);
                for (uint32_t lr = 0; lr < srows && !bad; lr += R) {
                    const uint32_t g = fr_shard_global_row(&ss, H, lr);
                    const size_t nr = (size_t)(lr + R <= srows ? R : srows - lr) * W, at = (size_t)g * W;
                    if (ship_rgba) bad |= nd->rccl.recv(nd->comms[k], (char*)frame.rgba + at * 16, nr * 16, src, stream, err, sizeof err);
                    if (ship_nu) bad |= nd->rccl.recv(nd->comms[k], (char*)frame.nu + at * nu_elt, nr * nu_elt, src, stream, err, sizeof err);
                    if (ship_iter) bad |= nd->rccl.recv(nd->comms[k], (char*)frame.iter + at * 4, nr * 4, src, stream, err, sizeof err);
                }
            }
            char gerr[256] = {0};
            if (nd->rccl.group_end(gerr, sizeof gerr) != 0 || bad)
                return fr_set_error(FR_ERR_HIP, "%s", bad ? err : gerr);
            if (nu_payload)           /* the assembled smooth-count frame -> colour, behind the receives on the same stream */
                return fr_colorize_async(c, &params, (uint64_t)npx, frame.nu, frame.rgba, stream);
            return FR_OK;
        }
        if (rows == 0) return FR_OK;
        /* packed planes of this part on its own device */
        const size_t pr = (size_t)rows * W;
        const size_t off_nu = ship_rgba ? pr * 16 : 0, off_iter = off_nu + (ship_nu ? pr * 8 : 0), need = off_iter + (ship_iter ? pr * 4 : 0);
        (void)fr_ctx_synchronize(c);
        int s = grow(&nd->stage[k], &nd->stage_bytes[k], need);
        if (s != FR_OK) return s;
        char* base = (char*)nd->stage[k];
        const fr_output o = {ship_rgba ? (float*)base : nullptr, ship_nu ? (void*)(base + off_nu) : nullptr,
                             ship_iter ? (int32_t*)(base + off_iter) : nullptr, FR_MEM_DEVICE, FR_LAYOUT_PACKED};
        s = fr_render_shard_async(c, &params, W, H, &sh, &o, nullptr);
        if (s != FR_OK) return s;
        if (nd->rccl.group_start() != 0) return fr_set_error(FR_ERR_HIP, "ncclGroupStart failed");
        int bad = 0;
        for (uint32_t lr = 0; lr < rows && !bad; lr += R) {
            const size_t nr = (size_t)(lr + R <= rows ? R : rows - lr) * W, at = (size_t)lr * W;
            if (ship_rgba) bad |= nd->rccl.send(nd->comms[k], base + at * 16, nr * 16, root, stream, err, sizeof err);
            if (ship_nu) bad |= nd->rccl.send(nd->comms[k], base + off_nu + at * nu_elt, nr * nu_elt, root, stream, err, sizeof err);
            if (ship_iter) bad |= nd->rccl.send(nd->comms[k], base + off_iter + at * 4, nr * 4, root, stream, err, sizeof err);
        }
        char gerr[256] = {0};
        if (nd->rccl.group_end(gerr, sizeof gerr) != 0 || bad)
            return fr_set_error(FR_ERR_HIP, "%s", bad ? err : gerr);
        return FR_OK;
    });
    nd->in_flight = true;                 /* even after a failure: some parts may have enqueued work that must drain */
    if (st != FR_OK) {
        char keep[512];
        snprintf(keep, sizeof keep, "%s", fr_last_error());
        (void)fr_node_wait(nd);
        return fr_set_error(st, "%s", keep);
    }
    return FR_OK;
}

extern "C" int fr_node_wait(fr_node* nd)
{
    if (!nd) return fr_set_error(FR_ERR_INVALID_ARG, "node is NULL");
    if (!nd->in_flight) return FR_OK;
    nd->in_flight = false;
    int st = run_all(nd, [&](int k) -> int { return fr_ctx_synchronize(nd->ctx[k]); });
    if (st != FR_OK) return st;
    if (nd->host_out) {
        NODE_HIP_TRY(hipSetDevice(nd->devices[nd->root]));
        const fr_output& u = nd->user_out;
        if (u.rgba) NODE_HIP_TRY(hipMemcpy(u.rgba, nd->dev_frame.rgba, nd->npx * 16, hipMemcpyDeviceToHost));
        if (u.nu) NODE_HIP_TRY(hipMemcpy(u.nu, nd->dev_frame.nu, nd->npx * nd->nu_elt, hipMemcpyDeviceToHost));
        if (u.iter) NODE_HIP_TRY(hipMemcpy(u.iter, nd->dev_frame.iter, nd->npx * 4, hipMemcpyDeviceToHost));
    }
    return FR_OK;
}

/* Internal (fr_tuning.h): drives the RCCL leg on ONE device -- loads the plugin, creates a one-rank communicator on
 * `device`, sends `bytes` bytes to itself through a grouped ncclSend / ncclRecv pair on a stream and compares.  What a
 * one-GPU box can exercise of the RCCL path: library load, communicator life cycle, the grouped point-to-point calls and
 * their stream ordering.  Returns FR_OK, or the failing step's status. */
extern "C" int fr_node_rccl_selftest(int device, size_t bytes, int* rccl_version)
{
    fr_node nd;
    nd.n = 1; nd.devices[0] = device; nd.distinct = true;
    int st = load_rccl(&nd);
    if (st != FR_OK) return st;
    if (rccl_version) *rccl_version = nd.rccl.version();
    NODE_HIP_TRY(hipSetDevice(device));
    char err[256] = {0};
    if (nd.rccl.init(nd.devices, 1, nd.comms, err, sizeof err) != 0) { dlclose(nd.rccl.handle); return fr_set_error(FR_ERR_HIP, "%s", err); }
    unsigned char *src = nullptr, *dst = nullptr, *host = (unsigned char*)malloc(2 * bytes);
    hipStream_t s = nullptr;
    st = FR_OK;
    auto fail = [&](const char* what, hipError_t e) { st = fr_set_error(FR_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e)); };
    hipError_t e;
    if (!host) st = fr_set_error(FR_ERR_NOMEM, "out of host memory");
    else if ((e = hipMalloc((void**)&src, bytes)) != hipSuccess) fail("hipMalloc", e);
    else if ((e = hipMalloc((void**)&dst, bytes)) != hipSuccess) fail("hipMalloc", e);
    else if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) fail("hipStreamCreate", e);
    if (st == FR_OK) {
        for (size_t i = 0; i < bytes; ++i) host[i] = (unsigned char)(i * 131u + 7u);
        if ((e = hipMemcpyAsync(src, host, bytes, hipMemcpyHostToDevice, s)) != hipSuccess) fail("hipMemcpyAsync", e);
        else if ((e = hipMemsetAsync(dst, 0, bytes, s)) != hipSuccess) fail("hipMemsetAsync", e);
    }
    if (st == FR_OK) {
        int bad = nd.rccl.group_start();
        bad |= nd.rccl.send(nd.comms[0], src, bytes, 0, (void*)s, err, sizeof err);
        bad |= nd.rccl.recv(nd.comms[0], dst, bytes, 0, (void*)s, err, sizeof err);
        char gerr[256] = {0};
        if (nd.rccl.group_end(gerr, sizeof gerr) != 0 || bad) st = fr_set_error(FR_ERR_HIP, "%s", bad ? err : gerr);
    }
    if (st == FR_OK) {
        if ((e = hipMemcpyAsync(host + bytes, dst, bytes, hipMemcpyDeviceToHost, s)) != hipSuccess) fail("hipMemcpyAsync", e);
        else if ((e = hipStreamSynchronize(s)) != hipSuccess) fail("hipStreamSynchronize", e);
        else if (memcmp(host, host + bytes, bytes) != 0) st = fr_set_error(FR_ERR_INTERNAL, "RCCL self send/recv returned different bytes");
    }
    if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    free(host);
    nd.rccl.destroy(nd.comms, 1);
    dlclose(nd.rccl.handle);
    nd.rccl.handle = nullptr;
    return st;
}

extern "C" int fr_node_render(fr_node* nd, const fr_params* p, uint32_t W, uint32_t H, int root, const fr_output* out)
{
    const int st = fr_node_render_async(nd, p, W, H, root, out);
    if (st != FR_OK) return st;
    return fr_node_wait(nd);
}
