/*
 * fr_node.cpp -- frames over the GPUs of a node, behind the C ABI (fr_node_* in include/fractalrenderer_amd.h).
 *
 * BASELINE.json's north star keeps the host in C and tiles the image "across the 8 GPUs of one node as disjoint row bands
 * with a final RCCL gather over xGMI".  The reference has nothing to restate here -- it renders on the one GPU it picked
 * (src/vk_engine.cpp:608) -- so this is the MI355X-side design of the same render(viewport, max_iter, out_buffer) surface
 * (src/animation_renderer.h:41-48) for a caller that owns several devices and, like the reference's caller
 * (AnimationRenderer::start_render, src/animation_renderer.cpp:75-127), renders a SEQUENCE of frames:
 *
 *   - ONE process; per device (part) one host WORKER THREAD bound to it, with a FIFO of frame jobs.  A part of a C2 frame is
 *     ~0.1 ms of GPU work behind three launches; enqueued from one thread, eight devices' worth of launches would take
 *     longer than the kernels run.  fr_node_submit only validates, takes a frame slot and posts the job to every worker.
 *   - the path shards with no exchange during compute: part k renders strips k, k + n, ... of the frame (fr_shard).
 *   - FRAMES IN FLIGHT: a ring of `slots` frame slots (per-frame state, staging buffers, events) and up to 4 RENDER LANES
 *     per part (a lane = one fr_ctx = its own stream, scratch and control block; frame t runs on lane t % lanes), so that
 *     frame t + 1's ramp-up fills frame t's drain on every device -- what bench.py --pipelined measures on one context
 *     pair, and what a 1/8 share of a frame needs even more (its fixed costs are 1.8x its size).
 *   - the gather is the only transfer, and there are two forms of it (fr_gather): the kernels' stores go straight into
 *     the root's planes through the peer mapping (FR_LAYOUT_FRAME: every part addresses whole-frame planes), or the
 *     parts ship their strips with grouped ncclSend / ncclRecv (fr_rccl_plugin.cpp) on a COMM STREAM per part, ordered
 *     behind the part's render by an event, received in place: a strip is one contiguous byte range both in the part's
 *     packed buffer and in the frame.  The root is per frame: a caller rotates it so that every link carries traffic.
 *   - the RCCL gather is TWO-PHASE: every part first grows its staging, enqueues its render and REPORTS; only when all
 *     parts reported success do the workers enter their ncclGroupStart ... End, so sends and receives always match.  A
 *     failure after that point (or a gather that does not finish within "rccl_timeout_ms") aborts every communicator
 *     (ncclCommAbort) -- no stream is left waiting for a transfer that cannot come -- and the node goes on with the
 *     in-place gather where the devices can map each other.
 *
 * The arithmetic of every part is fr_render_shard_async's, so every frame is byte-identical to fr_render's.
 * Not re-entrant: one caller thread at a time drives a node (as one fr_ctx); distinct nodes are independent.
 */
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "fr_internal.h"
#include "fr_tuning.h"

namespace {

constexpr int kMaxParts = 16;
constexpr int kMaxSlots = 8;
constexpr int kMaxLanes = 4;
constexpr int kResults = 64;             /* verdicts of the most recent tickets, for fr_node_wait_frame after the slot moved on */

struct RcclApi {
    void* handle = nullptr;
    int (*init)(const int*, int, void**, char*, size_t) = nullptr;
    void (*destroy)(void**, int) = nullptr;
    void (*abort)(void**, int) = nullptr;
    int (*group_start)(void) = nullptr;
    int (*group_end)(char*, size_t) = nullptr;
    int (*send)(void*, const void*, size_t, int, void*, char*, size_t) = nullptr;
    int (*recv)(void*, void*, size_t, int, void*, char*, size_t) = nullptr;
    int (*version)(void) = nullptr;
};

/* one host thread per part, bound to the part's device: runs the jobs handed to it, in order */
struct Worker {
    std::thread thread;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    bool quit = false;

    void loop()
    {
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return !jobs.empty() || quit; });
                if (jobs.empty()) return;                        /* quit, and nothing left to run */
                j = std::move(jobs.front());
                jobs.pop_front();
            }
            j();
        }
    }
    void post(std::function<void()> j)
    {
        { std::lock_guard<std::mutex> lk(m); jobs.push_back(std::move(j)); }
        cv.notify_one();
    }
};

struct Planes { float* rgba; void* nu; int32_t* iter; };

/* one frame in flight: what fr_node_submit decided, and what the workers report */
struct Frame {
    bool busy = false;                    /* submitted and not completed yet (caller thread only) */
    uint64_t ticket = 0;
    int slot = 0, lane = 0, root = 0, gather = FR_GATHER_PEER;
    bool rccl = false, loopback = false, nu_payload = false, host_out = false;
    fr_params params;
    uint32_t W = 0, H = 0, R = 0;
    size_t npx = 0, nu_elt = 0;
    fr_output user_out = {nullptr, nullptr, nullptr, 0, 0};
    Planes dev = {nullptr, nullptr, nullptr};   /* whole-frame planes on the root device */
    /* worker side */
    std::mutex m;
    std::condition_variable cv;
    int phase1_left = 0, jobs_left = 0;
    int status = FR_OK;                   /* first failure of any part */
    char err[512] = {0};
    bool done_recorded[kMaxParts] = {false};

    void fail(int k, int device, int st, const char* msg)
    {
        std::lock_guard<std::mutex> lk(m);
        if (status == FR_OK) {
            status = st;
            snprintf(err, sizeof err, "part %d (device %d): %s", k, device, msg);
        }
    }
};

struct Result { uint64_t ticket = 0; int status = FR_OK; bool reported = true; char err[512] = {0}; };

}  // namespace

struct fr_node {
    int n = 0;
    int devices[kMaxParts] = {0};
    bool distinct = false;                /* no ordinal occurs twice */
    bool peer_ok = false;                 /* every pair of different ordinals can map each other's memory */
    fr_ctx* ctx[kMaxParts][kMaxLanes] = {{nullptr}};
    Worker* workers[kMaxParts] = {nullptr};
    /* options */
    int gather = FR_GATHER_AUTO, layout = 0, payload = 0, slots = 2, lanes = 2;
    uint32_t rows_per_strip = 0;
    std::vector<std::pair<std::string, int64_t>> ctx_options, ctx_tunings;   /* replayed on contexts created later */
    std::mutex ctx_mu;                    /* ... which the workers do, while the caller may be setting options: never both
                                             (options are refused while frames are in flight), the lock is for the record */
    /* internal knobs (fr_tuning.h) */
    int fail_phase1 = 0, fail_before_send = 0;   /* part + 1, one shot */
    int rccl_loopback = 0;
    int rccl_timeout_ms = 30000;
    /* RCCL leg */
    RcclApi rccl;
    void* comms[kMaxParts] = {nullptr};
    std::timed_mutex comm_mu[kMaxParts];  /* held by part k's worker while it uses comms[k] */
    std::mutex abort_mu;
    std::atomic<bool> rccl_broken{false};
    hipStream_t comm_stream[kMaxParts] = {nullptr};
    /* per slot and part */
    hipEvent_t ev_rendered[kMaxSlots][kMaxParts] = {{nullptr}};
    hipEvent_t ev_done[kMaxSlots][kMaxParts] = {{nullptr}};
    void* stage[kMaxSlots][kMaxParts] = {{nullptr}};        /* packed staging on the part's own device (RCCL gather), grow-only */
    size_t stage_bytes[kMaxSlots][kMaxParts] = {{0}};
    void* frame_buf[kMaxSlots][kMaxParts] = {{nullptr}};    /* whole-frame staging on a root's device: FR_MEM_HOST outputs and the
                                                               nu frame of the nu payload, grow-only */
    size_t frame_bytes[kMaxSlots][kMaxParts] = {{0}};
    /* frames */
    Frame frames[kMaxSlots];
    Result results[kResults];
    uint64_t next_ticket = 1;
    int last_gather = -1, last_lane = 0;
};

namespace {

#define NODE_HIP_TRY(expr)                                                             \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fr_set_error(FR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

/* the entry points switch devices (peer mappings, staging on a root, copies back): the caller's current device is its own */
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

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
    r.abort = (decltype(r.abort))dlsym(h, "fr_rccl_abort");
    r.group_start = (decltype(r.group_start))dlsym(h, "fr_rccl_group_start");
    r.group_end = (decltype(r.group_end))dlsym(h, "fr_rccl_group_end");
    r.send = (decltype(r.send))dlsym(h, "fr_rccl_send");
    r.recv = (decltype(r.recv))dlsym(h, "fr_rccl_recv");
    r.version = (decltype(r.version))dlsym(h, "fr_rccl_version");
    if (!r.init || !r.destroy || !r.abort || !r.group_start || !r.group_end || !r.send || !r.recv || !r.version) {
        dlclose(h);
        return fr_set_error(FR_ERR_UNSUPPORTED, "%s lacks an entry point", path);
    }
    r.handle = h;
    return FR_OK;
}

int ensure_comms(fr_node* nd)
{
    if (nd->rccl_broken.load())
        return fr_set_error(FR_ERR_UNSUPPORTED, "the RCCL leg of this node was aborted after a failed gather");
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

/* Every communicator of the node, at once and only once.  Called by the worker whose gather failed after the two-phase
 * barrier (its own comm_mu released), or by the caller's thread when a gather ran out of time.  A worker that is INSIDE
 * its group calls holds comm_mu[j]: if it does not come out within 100 ms it is blocked in there (connecting to the
 * failed peer) and the abort is what lets it out -- ncclCommAbort may be called while another thread is blocked in a
 * call on that communicator; a worker that is merely enqueuing finishes in microseconds and the lock is had. */
void abort_rccl(fr_node* nd)
{
    std::lock_guard<std::mutex> g(nd->abort_mu);
    if (nd->rccl_broken.exchange(true)) return;
    if (!nd->rccl.handle) return;
    for (int j = 0; j < nd->n; ++j) {
        if (!nd->comms[j]) continue;
        const bool locked = nd->comm_mu[j].try_lock_for(std::chrono::milliseconds(100));
        void* c = nd->comms[j];
        nd->comms[j] = nullptr;
        nd->rccl.abort(&c, 1);
        if (locked) nd->comm_mu[j].unlock();
    }
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

int apply_ctx_settings(fr_node* nd, fr_ctx* c)
{
    for (const auto& o : nd->ctx_options) {
        const int st = fr_ctx_set_option(c, o.first.c_str(), o.second);
        if (st != FR_OK) return st;
    }
    for (const auto& o : nd->ctx_tunings) {
        const int st = fr_ctx_set_tuning(c, o.first.c_str(), o.second);
        if (st != FR_OK) return st;
    }
    return FR_OK;
}

/* worker k: the render context of lane `lane` (lane 0 exists since fr_node_create; the others on first use) */
int ensure_ctx(fr_node* nd, int k, int lane, fr_ctx** out)
{
    if (!nd->ctx[k][lane]) {
        fr_ctx* c = nullptr;
        int st = fr_ctx_create(nd->devices[k], &c);
        if (st != FR_OK) return st;
        {
            std::lock_guard<std::mutex> lk(nd->ctx_mu);
            st = apply_ctx_settings(nd, c);
        }
        if (st != FR_OK) { fr_ctx_destroy(c); return st; }
        nd->ctx[k][lane] = c;
    }
    *out = nd->ctx[k][lane];
    return FR_OK;
}

int ensure_events(fr_node* nd, int slot, int k)
{
    if (!nd->ev_done[slot][k]) NODE_HIP_TRY(hipEventCreateWithFlags(&nd->ev_done[slot][k], hipEventDisableTiming));
    if (!nd->ev_rendered[slot][k]) NODE_HIP_TRY(hipEventCreateWithFlags(&nd->ev_rendered[slot][k], hipEventDisableTiming));
    return FR_OK;
}

void remember_setting(std::vector<std::pair<std::string, int64_t>>& list, const char* name, int64_t value)
{
    for (auto& o : list)
        if (o.first == name) { o.second = value; return; }
    list.emplace_back(name, value);
}

/* ---- what worker k does for frame F --------------------------------------------------------------------------------- */

/* phase 1: everything that can fail for reasons of this part alone -- context, staging, the render's launches */
int part_phase1(fr_node* nd, Frame* F, int k, hipStream_t* last_stream)
{
    if (nd->fail_phase1 == k + 1) {
        nd->fail_phase1 = 0;
        return fr_set_error(FR_ERR_INTERNAL, "injected failure (fr_node_set_tuning \"fail_part_phase1\")");
    }
    fr_ctx* c = nullptr;
    int st = ensure_ctx(nd, k, F->lane, &c);
    if (st != FR_OK) return st;
    NODE_HIP_TRY(hipSetDevice(nd->devices[k]));
    st = ensure_events(nd, F->slot, k);
    if (st != FR_OK) return st;
    hipStream_t lane_stream = (hipStream_t)fr_ctx_stream_handle(c);
    *last_stream = lane_stream;
    const int n = nd->n;
    const fr_shard sh = {(uint32_t)k, (uint32_t)n, F->R};
    const uint32_t rows = fr_shard_rows(&sh, F->H);
    const Planes& fr = F->dev;
    if (!F->rccl) {
        if (rows == 0) return FR_OK;
        const fr_output o = {fr.rgba, fr.nu, fr.iter, FR_MEM_DEVICE, FR_LAYOUT_FRAME};
        return fr_render_shard_async(c, &F->params, F->W, F->H, &sh, &o, nullptr);
    }
    /* ---- RCCL gather: the root renders its own strips in place, every other part into packed staging of its own ---- */
    const bool ship_rgba = fr.rgba && !F->nu_payload, ship_nu = fr.nu != nullptr, ship_iter = fr.iter != nullptr;
    const bool source = k != F->root || F->loopback;
    if (rows) {
        if (!source) {
            const fr_output o = {ship_rgba ? fr.rgba : nullptr, fr.nu, fr.iter, FR_MEM_DEVICE, FR_LAYOUT_FRAME};
            st = fr_render_shard_async(c, &F->params, F->W, F->H, &sh, &o, nullptr);
        } else {
            const size_t pr = (size_t)rows * F->W;
            const size_t off_nu = ship_rgba ? pr * 16 : 0, off_iter = off_nu + (ship_nu ? pr * 8 : 0), need = off_iter + (ship_iter ? pr * 4 : 0);
            /* the slot's previous frame was completed before the slot was handed out: nobody reads the old buffer */
            st = grow(&nd->stage[F->slot][k], &nd->stage_bytes[F->slot][k], need);
            if (st != FR_OK) return st;
            char* base = (char*)nd->stage[F->slot][k];
            const fr_output o = {ship_rgba ? (float*)base : nullptr, ship_nu ? (void*)(base + off_nu) : nullptr,
                                 ship_iter ? (int32_t*)(base + off_iter) : nullptr, FR_MEM_DEVICE, FR_LAYOUT_PACKED};
            st = fr_render_shard_async(c, &F->params, F->W, F->H, &sh, &o, nullptr);
        }
        if (st != FR_OK) return st;
    }
    if (!nd->comm_stream[k]) NODE_HIP_TRY(hipStreamCreateWithFlags(&nd->comm_stream[k], hipStreamNonBlocking));
    NODE_HIP_TRY(hipEventRecord(nd->ev_rendered[F->slot][k], lane_stream));
    return FR_OK;
}

/* phase 2 (RCCL gather, entered only when EVERY part's phase 1 succeeded): the part's sends, the root's receives, on the
 * part's comm stream behind its render */
int part_phase2(fr_node* nd, Frame* F, int k, hipStream_t* last_stream)
{
    const int n = nd->n, root = F->root;
    const uint32_t R = F->R, W = F->W, H = F->H;
    const Planes& fr = F->dev;
    const bool ship_rgba = fr.rgba && !F->nu_payload, ship_nu = fr.nu != nullptr, ship_iter = fr.iter != nullptr;
    const bool source = k != root || F->loopback;
    const size_t nu_elt = F->nu_elt;
    hipStream_t comm = nd->comm_stream[k];
    NODE_HIP_TRY(hipStreamWaitEvent(comm, nd->ev_rendered[F->slot][k], 0));
    *last_stream = comm;

    std::unique_lock<std::timed_mutex> lk(nd->comm_mu[k]);
    void* mine = nd->comms[k];
    if (nd->rccl_broken.load() || !mine)
        return fr_set_error(FR_ERR_HIP, "the RCCL leg of this node was aborted after a failed gather");
    char err[256] = {0}, gerr[256] = {0};
    int bad = 0;
    bool injected = false;
    if (nd->rccl.group_start() != 0) return fr_set_error(FR_ERR_HIP, "ncclGroupStart failed");
    if (k == root) {
        for (int src = 0; src < n && !bad; ++src) {
            if (src == root && !F->loopback) continue;
            const fr_shard ss = {(uint32_t)src, (uint32_t)n, R};
            const uint32_t srows = fr_shard_rows(&ss, H);
            for (uint32_t lr = 0; lr < srows && !bad; lr += R) {
                const uint32_t g = fr_shard_global_row(&ss, H, lr);
                const size_t nr = (size_t)(lr + R <= srows ? R : srows - lr) * W, at = (size_t)g * W;
                if (ship_rgba) bad |= nd->rccl.recv(mine, (char*)fr.rgba + at * 16, nr * 16, src, comm, err, sizeof err);
                if (ship_nu) bad |= nd->rccl.recv(mine, (char*)fr.nu + at * nu_elt, nr * nu_elt, src, comm, err, sizeof err);
                if (ship_iter) bad |= nd->rccl.recv(mine, (char*)fr.iter + at * 4, nr * 4, src, comm, err, sizeof err);
            }
        }
    }
    if (source) {
        if (nd->fail_before_send == k + 1) {          /* fr_tuning.h: this part "dies" between the barrier and its sends */
            nd->fail_before_send = 0;
            injected = true;
        } else {
            const fr_shard sh = {(uint32_t)k, (uint32_t)n, R};
            const uint32_t rows = fr_shard_rows(&sh, H);
            const size_t pr = (size_t)rows * W;
            const size_t off_nu = ship_rgba ? pr * 16 : 0, off_iter = off_nu + (ship_nu ? pr * 8 : 0);
            const char* base = (const char*)nd->stage[F->slot][k];
            for (uint32_t lr = 0; lr < rows && !bad; lr += R) {
                const size_t nr = (size_t)(lr + R <= rows ? R : rows - lr) * W, at = (size_t)lr * W;
                if (ship_rgba) bad |= nd->rccl.send(mine, base + at * 16, nr * 16, root, comm, err, sizeof err);
                if (ship_nu) bad |= nd->rccl.send(mine, base + off_nu + at * nu_elt, nr * nu_elt, root, comm, err, sizeof err);
                if (ship_iter) bad |= nd->rccl.send(mine, base + off_iter + at * 4, nr * 4, root, comm, err, sizeof err);
            }
        }
    }
    const int ge = nd->rccl.group_end(gerr, sizeof gerr);
    lk.unlock();
    if (injected) return fr_set_error(FR_ERR_INTERNAL, "injected failure (fr_node_set_tuning \"fail_part_before_send\")%s%s",
                                      ge ? "; " : "", ge ? gerr : "");
    if (ge != 0 || bad) return fr_set_error(FR_ERR_HIP, "%s", bad ? err : gerr);
    if (k == root && F->nu_payload) {   /* the assembled smooth-count frame -> colour, behind the receives on the same stream */
        fr_ctx* c = nd->ctx[k][F->lane];
        return fr_colorize_async(c, &F->params, (uint64_t)F->npx, fr.nu, fr.rgba, comm);
    }
    return FR_OK;
}

void part_job(fr_node* nd, Frame* F, int k)
{
    hipStream_t last = nullptr;
    int st = part_phase1(nd, F, k, &last);
    if (st != FR_OK) F->fail(k, nd->devices[k], st, fr_last_error());
    if (F->rccl) {
        /* the two-phase barrier: nobody posts a send or a receive before every part has its render enqueued */
        bool go;
        {
            std::unique_lock<std::mutex> lk(F->m);
            if (--F->phase1_left == 0) F->cv.notify_all();
            F->cv.wait(lk, [&] { return F->phase1_left == 0; });
            go = F->status == FR_OK;
        }
        if (go) {
            st = part_phase2(nd, F, k, &last);
            if (st != FR_OK) {
                F->fail(k, nd->devices[k], st, fr_last_error());
                abort_rccl(nd);          /* peers may have posted their side of a transfer that will not happen */
            }
        }
    }
    bool recorded = false;
    if (last && nd->ev_done[F->slot][k]) {
        const hipError_t e = hipEventRecord(nd->ev_done[F->slot][k], last);
        if (e == hipSuccess) recorded = true;
        else F->fail(k, nd->devices[k], FR_ERR_HIP, hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lk(F->m);
        F->done_recorded[k] = recorded;
        if (--F->jobs_left == 0) F->cv.notify_all();
    }
}

/* caller thread: wait for frame F (every part's enqueue, then every part's device work), copy host outputs back, file the
 * verdict under its ticket */
int complete_frame(fr_node* nd, Frame* F)
{
    {
        std::unique_lock<std::mutex> lk(F->m);
        F->cv.wait(lk, [&] { return F->jobs_left == 0; });
    }
    int st = F->status;
    char err[512];
    snprintf(err, sizeof err, "%s", F->err);
    auto fail = [&](int s, const char* msg) { if (st == FR_OK) { st = s; snprintf(err, sizeof err, "%s", msg); } };
    for (int k = 0; k < nd->n; ++k) {
        if (!F->done_recorded[k]) continue;
        hipEvent_t ev = nd->ev_done[F->slot][k];
        if (F->rccl && nd->rccl_timeout_ms > 0) {
            /* a gather that does not finish: abort the communicators rather than wait for ever */
            const auto t0 = std::chrono::steady_clock::now();
            while (hipEventQuery(ev) == hipErrorNotReady) {
                const auto waited = std::chrono::steady_clock::now() - t0;
                if (waited > std::chrono::milliseconds(nd->rccl_timeout_ms)) {
                    fail(FR_ERR_HIP, "the RCCL gather did not finish within \"rccl_timeout_ms\": communicators aborted");
                    abort_rccl(nd);
                    break;
                }
                if (waited > std::chrono::milliseconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(100));
                else std::this_thread::yield();
            }
            (void)hipGetLastError();
        }
        const hipError_t e = hipEventSynchronize(ev);
        if (e != hipSuccess) fail(FR_ERR_HIP, hipGetErrorString(e));
    }
    for (int k = 0; k < nd->n; ++k) {
        fr_ctx* c = nd->ctx[k][F->lane];
        if (c && fr_ctx_check(c) != FR_OK) fail(FR_ERR_INTERNAL, fr_last_error());
    }
    if (st == FR_OK && F->host_out) {
        const fr_output& u = F->user_out;
        hipError_t e = hipSetDevice(nd->devices[F->root]);
        if (e == hipSuccess && u.rgba) e = hipMemcpy(u.rgba, F->dev.rgba, F->npx * 16, hipMemcpyDeviceToHost);
        if (e == hipSuccess && u.nu) e = hipMemcpy(u.nu, F->dev.nu, F->npx * F->nu_elt, hipMemcpyDeviceToHost);
        if (e == hipSuccess && u.iter) e = hipMemcpy(u.iter, F->dev.iter, F->npx * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) fail(FR_ERR_HIP, hipGetErrorString(e));
    }
    Result& r = nd->results[F->ticket % kResults];
    r.ticket = F->ticket;
    r.status = st;
    r.reported = false;
    snprintf(r.err, sizeof r.err, "%s", err);
    F->busy = false;
    return st;
}

int frames_in_flight(const fr_node* nd)
{
    int c = 0;
    for (int s = 0; s < kMaxSlots; ++s) c += nd->frames[s].busy ? 1 : 0;
    return c;
}

}  // namespace

extern "C" int fr_node_create(const int* devices, int n, fr_node** out)
{
    if (!out) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > kMaxParts)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_create: 1..%d devices", kMaxParts);
    DeviceGuard guard;
    fr_node* nd = new (std::nothrow) fr_node();
    if (!nd) return fr_set_error(FR_ERR_NOMEM, "out of host memory");
    nd->n = n;
    nd->distinct = true;
    for (int k = 0; k < n; ++k) {
        nd->devices[k] = devices[k];
        for (int j = 0; j < k; ++j) if (devices[j] == devices[k]) nd->distinct = false;
    }
    for (int k = 0; k < n; ++k) {
        const int st = fr_ctx_create(devices[k], &nd->ctx[k][0]);
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
    DeviceGuard guard;
    (void)fr_node_wait(nd);
    for (int k = 0; k < nd->n; ++k) {
        Worker* w = nd->workers[k];
        if (!w) continue;
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
        w->cv.notify_all();
        if (w->thread.joinable()) w->thread.join();
        delete w;
    }
    if (nd->rccl.handle) {
        const int ncomm = nd->rccl_loopback && nd->n == 1 ? 1 : nd->n;
        if (!nd->rccl_broken.load()) nd->rccl.destroy(nd->comms, ncomm);
        dlclose(nd->rccl.handle);
    }
    for (int k = 0; k < nd->n; ++k) {
        (void)hipSetDevice(nd->devices[k]);
        if (nd->comm_stream[k]) { (void)hipStreamSynchronize(nd->comm_stream[k]); (void)hipStreamDestroy(nd->comm_stream[k]); }
        for (int s = 0; s < kMaxSlots; ++s) {
            if (nd->ev_done[s][k]) (void)hipEventDestroy(nd->ev_done[s][k]);
            if (nd->ev_rendered[s][k]) (void)hipEventDestroy(nd->ev_rendered[s][k]);
        }
        for (int l = 0; l < kMaxLanes; ++l)
            if (nd->ctx[k][l]) fr_ctx_destroy(nd->ctx[k][l]);        /* synchronises the device before it frees */
        for (int s = 0; s < kMaxSlots; ++s) {
            if (nd->stage[s][k]) (void)hipFree(nd->stage[s][k]);
            if (nd->frame_buf[s][k]) (void)hipFree(nd->frame_buf[s][k]);
        }
    }
    delete nd;
}

extern "C" int fr_node_device_count(const fr_node* nd) { return nd ? nd->n : fr_set_error(FR_ERR_INVALID_ARG, "node is NULL"); }

extern "C" int fr_node_last_gather(const fr_node* nd) { return nd ? nd->last_gather : fr_set_error(FR_ERR_INVALID_ARG, "node is NULL"); }

extern "C" int fr_node_in_flight(const fr_node* nd) { return nd ? frames_in_flight(nd) : fr_set_error(FR_ERR_INVALID_ARG, "node is NULL"); }

extern "C" float fr_node_last_kernel_ms(fr_node* nd, int part)
{
    if (!nd || part < 0 || part >= nd->n || !nd->ctx[part][nd->last_lane]) return -1.0f;
    DeviceGuard guard;
    return fr_ctx_last_kernel_ms(nd->ctx[part][nd->last_lane]);
}

extern "C" int fr_node_set_option(fr_node* nd, const char* name, int64_t value)
{
    if (!nd || !name) return fr_set_error(FR_ERR_INVALID_ARG, "node/name is NULL");
    if (frames_in_flight(nd)) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_set_option: frames are in flight (fr_node_wait first)");
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
    } else if (!strcmp(name, "slots")) {
        if (value < 0 || value > kMaxSlots) return fr_set_error(FR_ERR_INVALID_ARG, "slots must be 0 (automatic: 2) or 1..%d", kMaxSlots);
        nd->slots = value ? (int)value : 2;
    } else if (!strcmp(name, "lanes")) {
        if (value < 0 || value > kMaxLanes) return fr_set_error(FR_ERR_INVALID_ARG, "lanes must be 0 (automatic: 2) or 1..%d", kMaxLanes);
        nd->lanes = value ? (int)value : 2;
    } else {
        std::lock_guard<std::mutex> lk(nd->ctx_mu);
        for (int k = 0; k < nd->n; ++k)
            for (int l = 0; l < kMaxLanes; ++l)
                if (nd->ctx[k][l]) {
                    const int st = fr_ctx_set_option(nd->ctx[k][l], name, value);
                    if (st != FR_OK) return st;
                }
        remember_setting(nd->ctx_options, name, value);
    }
    return FR_OK;
}

/* Internal (fr_tuning.h): fault injection and the one-card RCCL loopback; every other name is an fr_ctx_set_tuning name,
 * applied to all render contexts of the node. */
extern "C" int fr_node_set_tuning(fr_node* nd, const char* name, int64_t value)
{
    if (!nd || !name) return fr_set_error(FR_ERR_INVALID_ARG, "node/name is NULL");
    if (frames_in_flight(nd)) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_set_tuning: frames are in flight (fr_node_wait first)");
    if (!strcmp(name, "fail_part_phase1")) nd->fail_phase1 = (int)value;
    else if (!strcmp(name, "fail_part_before_send")) nd->fail_before_send = (int)value;
    else if (!strcmp(name, "rccl_loopback")) {
        if (value && nd->n != 1) return fr_set_error(FR_ERR_INVALID_ARG, "rccl_loopback drives the RCCL gather through a one-rank communicator: nodes of ONE part only");
        nd->rccl_loopback = value ? 1 : 0;
    } else if (!strcmp(name, "rccl_timeout_ms")) nd->rccl_timeout_ms = (int)value;
    else {
        std::lock_guard<std::mutex> lk(nd->ctx_mu);
        for (int k = 0; k < nd->n; ++k)
            for (int l = 0; l < kMaxLanes; ++l)
                if (nd->ctx[k][l]) {
                    const int st = fr_ctx_set_tuning(nd->ctx[k][l], name, value);
                    if (st != FR_OK) return st;
                }
        remember_setting(nd->ctx_tunings, name, value);
    }
    return FR_OK;
}

extern "C" int fr_node_rccl_usable(const fr_node* nd) { return nd && !nd->rccl_broken.load() ? 1 : 0; }

extern "C" int fr_node_submit(fr_node* nd, const fr_params* p, uint32_t W, uint32_t H, int root, const fr_output* out,
                              uint64_t* ticket)
{
    if (ticket) *ticket = 0;
    if (!nd || !p || !out) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_submit: NULL argument");
    int st = fr_params_validate(p, W, H);
    if (st != FR_OK) return st;
    if (!out->rgba && !out->nu && !out->iter) return fr_set_error(FR_ERR_INVALID_ARG, "fr_output has no plane to write");
    if (out->memory != FR_MEM_DEVICE && out->memory != FR_MEM_HOST)
        return fr_set_error(FR_ERR_INVALID_ARG, "unknown fr_output.memory %d", out->memory);
    if (out->layout != FR_LAYOUT_PACKED)
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_render assembles whole frames: fr_output.layout must be 0");
    const uint64_t t = nd->next_ticket;
    const bool host_out = out->memory == FR_MEM_HOST;
    if (root == FR_ROOT_ROTATE) {
        /* the planes of a device output live on ONE device, which only the caller knows */
        bool one_device = true;
        for (int k = 1; k < nd->n; ++k) one_device = one_device && nd->devices[k] == nd->devices[0];
        if (!host_out && !one_device)
            return fr_set_error(FR_ERR_INVALID_ARG, "FR_ROOT_ROTATE needs FR_MEM_HOST planes (device planes live on the device the caller names as root)");
        root = (int)((t - 1) % (uint64_t)nd->n);
    }
    if (root < 0 || root >= nd->n) return fr_set_error(FR_ERR_INVALID_ARG, "root %d outside [0, %d)", root, nd->n);
    DeviceGuard guard;

    /* ---- which gather ---- */
    const bool loopback = nd->rccl_loopback && nd->n == 1 && nd->gather == FR_GATHER_RCCL;
    int gather = nd->gather;
    if (gather == FR_GATHER_AUTO)          /* the in-place gather wherever the devices can map each other: the one the one-card
                                              tests and the soak compare bitwise; RCCL where they cannot */
        gather = (nd->n == 1 || !nd->distinct || nd->peer_ok) ? FR_GATHER_PEER : FR_GATHER_RCCL;
    if (gather == FR_GATHER_RCCL && (nd->n > 1 || loopback)) {
        st = ensure_comms(nd);
        if (st != FR_OK) {
            /* no plugin, no communicators, or the leg was aborted: the in-place gather serves where it can -- except for a node
             * that lists a device twice and asked for RCCL: that is a caller's mistake, not an environment's */
            if (!nd->distinct || !nd->peer_ok) return st;
            gather = FR_GATHER_PEER;
        }
    }
    if (gather == FR_GATHER_PEER && !nd->peer_ok)
        return fr_set_error(FR_ERR_UNSUPPORTED, "in-place gather: the devices of this node cannot map each other's memory");
    const bool rccl = gather == FR_GATHER_RCCL && (nd->n > 1 || loopback);

    /* ---- a slot: the one ticket t maps to; its previous frame is completed first (its verdict is kept for its ticket) ---- */
    const int slot = (int)(t % (uint64_t)nd->slots);
    Frame* F = &nd->frames[slot];
    if (F->busy) (void)complete_frame(nd, F);

    const uint32_t R = strip_rows(nd, H);
    const size_t npx = (size_t)W * H;
    const size_t nu_elt = (p->precision == FR_PRECISION_F64 && p->fractal_type != FR_FRACTAL_DEEP_ZOOM) ? 8 : 4;
    /* nu payload (RCCL gather): parts render and ship only the smooth-count plane, the root recolours */
    const bool nu_payload = rccl && nd->payload == 0 && out->rgba && !out->iter && fr_colorize_supported(p) == 1;
    const bool need_nu_frame = nu_payload && !out->nu;

    /* whole-frame planes on the root device: the caller's, or (host outputs / the nu frame of the nu payload) the slot's */
    Planes frame = {out->rgba, out->nu, out->iter};
    if (host_out || need_nu_frame) {
        const size_t off_nu = host_out && out->rgba ? npx * 16 : 0;
        const size_t off_iter = off_nu + ((host_out && out->nu) || need_nu_frame ? npx * 8 : 0);
        const size_t need = off_iter + (host_out && out->iter ? npx * 4 : 0);
        NODE_HIP_TRY(hipSetDevice(nd->devices[root]));
        st = grow(&nd->frame_buf[slot][root], &nd->frame_bytes[slot][root], need);
        if (st != FR_OK) return st;
        char* base = (char*)nd->frame_buf[slot][root];
        if (host_out) {
            frame.rgba = out->rgba ? (float*)base : nullptr;
            frame.nu = out->nu || need_nu_frame ? (void*)(base + off_nu) : nullptr;
            frame.iter = out->iter ? (int32_t*)(base + off_iter) : nullptr;
        } else {
            frame.nu = (void*)(base + off_nu);
        }
    }

    F->busy = true;
    F->ticket = t;
    F->slot = slot;
    F->lane = (int)(t % (uint64_t)nd->lanes);
    F->root = root;
    F->gather = gather;
    F->rccl = rccl;
    F->loopback = loopback;
    F->nu_payload = nu_payload;
    F->host_out = host_out;
    F->params = *p;
    F->W = W; F->H = H; F->R = R;
    F->npx = npx; F->nu_elt = nu_elt;
    F->user_out = *out;
    F->dev = frame;
    F->phase1_left = nd->n;
    F->jobs_left = nd->n;
    F->status = FR_OK;
    F->err[0] = 0;
    for (int k = 0; k < nd->n; ++k) F->done_recorded[k] = false;
    nd->last_gather = gather;
    nd->last_lane = F->lane;
    ++nd->next_ticket;
    for (int k = 0; k < nd->n; ++k) nd->workers[k]->post([nd, F, k] { part_job(nd, F, k); });
    if (ticket) *ticket = t;
    return FR_OK;
}

static int report(fr_node* nd, uint64_t ticket)
{
    Result& r = nd->results[ticket % kResults];
    if (r.ticket != ticket)
        return fr_set_error(FR_ERR_INVALID_ARG, "ticket %llu is older than the %d verdicts a node keeps", (unsigned long long)ticket, kResults);
    r.reported = true;
    return r.status == FR_OK ? FR_OK : fr_set_error(r.status, "frame %llu: %s", (unsigned long long)ticket, r.err);
}

extern "C" int fr_node_wait_frame(fr_node* nd, uint64_t ticket)
{
    if (!nd) return fr_set_error(FR_ERR_INVALID_ARG, "node is NULL");
    if (ticket == 0 || ticket >= nd->next_ticket) return fr_set_error(FR_ERR_INVALID_ARG, "ticket %llu was never handed out", (unsigned long long)ticket);
    DeviceGuard guard;
    for (int s = 0; s < kMaxSlots; ++s) {
        Frame* F = &nd->frames[s];
        if (F->busy && F->ticket == ticket) (void)complete_frame(nd, F);
    }
    return report(nd, ticket);
}

extern "C" int fr_node_wait(fr_node* nd)
{
    if (!nd) return fr_set_error(FR_ERR_INVALID_ARG, "node is NULL");
    DeviceGuard guard;
    /* oldest first */
    for (;;) {
        Frame* oldest = nullptr;
        for (int s = 0; s < kMaxSlots; ++s) {
            Frame* F = &nd->frames[s];
            if (F->busy && (!oldest || F->ticket < oldest->ticket)) oldest = F;
        }
        if (!oldest) break;
        (void)complete_frame(nd, oldest);
    }
    /* the first failure nobody has been told about yet; every verdict on file counts as delivered afterwards */
    int st = FR_OK;
    uint64_t first = 0;
    for (int i = 0; i < kResults; ++i) {
        Result& r = nd->results[i];
        if (r.ticket && !r.reported) {
            r.reported = true;
            if (r.status != FR_OK && (first == 0 || r.ticket < first)) { first = r.ticket; st = r.status; }
        }
    }
    if (st != FR_OK) {
        const Result& r = nd->results[first % kResults];
        return fr_set_error(st, "frame %llu: %s", (unsigned long long)first, r.err);
    }
    return FR_OK;
}

extern "C" int fr_node_render_async(fr_node* nd, const fr_params* p, uint32_t W, uint32_t H, int root, const fr_output* out)
{
    return fr_node_submit(nd, p, W, H, root, out, nullptr);
}

extern "C" int fr_node_render(fr_node* nd, const fr_params* p, uint32_t W, uint32_t H, int root, const fr_output* out)
{
    uint64_t t = 0;
    const int st = fr_node_submit(nd, p, W, H, root, out, &t);
    if (st != FR_OK) return st;
    return fr_node_wait_frame(nd, t);
}

/* ---- AnimationRenderer::start_render over the GPUs of a node ------------------------------------------------------------
 * src/animation_renderer.cpp:26-152 is a loop: time = frame / float(fps) (:80), state = interpolate(time) (:83),
 * "<folder>/frame_%06d.png" (:86-88), render_frame -> the RenderFrameCallback (:216), on_frame_complete (:123-125),
 * cancel_requested (:76).  Here every frame is cut over the node's parts (fr_node_submit, roots rotating, up to "slots"
 * frames in flight), and what the reference's callback body does after its dispatch (src/vk_engine.cpp:1266-1381: readback,
 * half -> float, second ACES + gamma, u8, flip, PNG) runs where the frame was assembled: fr_export_rgb8 on the root's device,
 * 3 bytes per pixel come back, and a writer thread deflates (fr_write_png, band-parallel) while the devices render the next
 * frames.  The files are byte-identical to fr_render_frame_png's. */
namespace {

int make_dirs(const char* path)                           /* std::filesystem::create_directories, :58-69 */
{
    char buf[4096];
    const size_t n = strlen(path);
    if (n == 0 || n >= sizeof buf) return fr_set_error(FR_ERR_INVALID_ARG, "output folder: empty or too long");
    memcpy(buf, path, n + 1);
    for (size_t i = 1; i <= n; ++i) {
        if (buf[i] != '/' && buf[i] != 0) continue;
        const char keep = buf[i];
        buf[i] = 0;
        if (mkdir(buf, 0777) != 0 && errno != EEXIST) return fr_set_error(FR_ERR_IO, "Failed to create output directory %s: %s", buf, strerror(errno));
        buf[i] = keep;
    }
    struct stat sb;
    if (stat(path, &sb) != 0 || !S_ISDIR(sb.st_mode)) return fr_set_error(FR_ERR_IO, "Failed to create output directory %s", path);
    return FR_OK;
}

struct AnimJob { int32_t frame; uint8_t* rgb8; };

struct AnimWriter {                                       /* one thread: PNG files, in submission order */
    std::thread thread;
    std::mutex m;
    std::condition_variable cv;
    std::deque<AnimJob> jobs;
    std::deque<uint8_t*> free_bufs;
    std::deque<int32_t> written;                          /* frames whose files are complete, for the caller's callbacks */
    bool quit = false;
    int status = FR_OK;
    char err[512] = {0};
    const char* folder = nullptr;
    int raw_fd = 0;                                       /* > 0: packed RGB24 frames to this descriptor instead of PNG files */
    uint32_t W = 0, H = 0;

    void loop()
    {
        for (;;) {
            AnimJob j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return !jobs.empty() || quit; });
                if (jobs.empty()) return;
                j = jobs.front();
                jobs.pop_front();
            }
            char path[4096];
            int st;
            if (raw_fd > 0) st = fr_write_raw_rgb24(raw_fd, j.rgb8, W, H);
            else {
                st = fr_frame_path(folder, j.frame, path, sizeof path);
                if (st == FR_OK) st = fr_write_png(path, W, H, 8, j.rgb8, nullptr, 0, 0);
            }
            {
                std::lock_guard<std::mutex> lk(m);
                if (st != FR_OK && status == FR_OK) { status = st; snprintf(err, sizeof err, "frame %d: %s", j.frame, fr_last_error()); }
                if (st == FR_OK) written.push_back(j.frame);
                free_bufs.push_back(j.rgb8);
            }
            cv.notify_all();
        }
    }
};

}  // namespace

extern "C" int fr_node_render_animation(fr_node* nd, const fr_anim* anim, const fr_params* base, const fr_anim_render_options* opt,
                                        const char* output_folder, int32_t* frames_written)
{
    if (frames_written) *frames_written = 0;
    if (!nd || !anim || !base) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_render_animation: NULL argument");
    if (!output_folder && !(opt && opt->raw_fd > 0))
        return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_render_animation: no output folder and no raw_fd");
    if (frames_in_flight(nd)) return fr_set_error(FR_ERR_INVALID_ARG, "fr_node_render_animation: frames are in flight (fr_node_wait first)");
    fr_anim_info info;
    int st = fr_anim_get_info(anim, &info);
    if (st != FR_OK) return st;
    if (info.keyframe_count < 2) return fr_set_error(FR_ERR_INVALID_ARG, "Need at least 2 keyframes to render");       /* :35-42 */
    fr_anim_render_options none;
    memset(&none, 0, sizeof none);
    const fr_anim_render_options o = opt ? *opt : none;
    const int32_t total = fr_anim_frame_count(anim);                                                                   /* :48 */
    const uint32_t W = (uint32_t)(o.width > 0 ? o.width : info.export_width), H = (uint32_t)(o.height > 0 ? o.height : info.export_height);
    const int32_t step = o.frame_step > 1 ? o.frame_step : 1;
    const int32_t first = o.first_frame > 0 ? o.first_frame : 0;
    int32_t last = total;                                 /* exclusive */
    if (o.frame_count > 0 && (int64_t)first + (int64_t)o.frame_count * step < (int64_t)total) last = first + o.frame_count * step;
    {   /* the first frame's parameters must be renderable before anything is created */
        fr_params p0;
        st = fr_anim_state_at(anim, fr_anim_frame_time(anim, first < total ? first : 0), base, &p0);
        if (st == FR_OK) st = fr_params_validate(&p0, W, H);
        if (st != FR_OK) return st;
    }
    if (o.raw_fd <= 0) {
        st = make_dirs(output_folder);
        if (st != FR_OK) return st;
    }
    if (first >= last) return FR_OK;

    DeviceGuard guard;
    const int n = nd->n, S = nd->slots;
    const size_t npx = (size_t)W * H;
    /* per plane set (as many as frame slots) and root: the frame's colour plane and its RGB8 on the root's device, on first use */
    float* d_rgba[kMaxSlots][kMaxParts] = {{nullptr}};
    uint8_t* d_rgb8[kMaxSlots][kMaxParts] = {{nullptr}};
    hipStream_t xs[kMaxParts] = {nullptr};                /* export + copy-back stream per root */
    std::vector<uint8_t*> host_bufs;
    AnimWriter wr;
    wr.folder = output_folder; wr.raw_fd = o.raw_fd > 0 ? o.raw_fd : 0; wr.W = W; wr.H = H;
    struct Pending { uint64_t ticket; int32_t frame; int root, set; };
    std::deque<Pending> pending;
    int rc = FR_OK;
    char keep[512] = {0};
    int32_t done = 0;
    bool cancelled = false;
    auto fail = [&](int s) { if (rc == FR_OK) { rc = s; snprintf(keep, sizeof keep, "%s", fr_last_error()); } };
    auto hip_fail = [&](const char* what, hipError_t e) { if (rc == FR_OK) { rc = FR_ERR_HIP; snprintf(keep, sizeof keep, "%s failed: %s", what, hipGetErrorString(e)); } };

    for (int k = 0; k < S + 1 && rc == FR_OK; ++k) {      /* pinned RGB8 frames: one being filled, S at the writer */
        uint8_t* b = nullptr;
        const hipError_t e = hipHostMalloc((void**)&b, npx * 3, hipHostMallocDefault);
        if (e != hipSuccess) { hip_fail("hipHostMalloc", e); break; }
        host_bufs.push_back(b);
        wr.free_bufs.push_back(b);
    }
    if (rc == FR_OK) wr.thread = std::thread([&wr] { wr.loop(); });

    /* the caller's callbacks, on the caller's thread, for frames whose files are complete */
    auto deliver = [&]() {
        for (;;) {
            int32_t f;
            {
                std::lock_guard<std::mutex> lk(wr.m);
                if (wr.status != FR_OK && rc == FR_OK) { rc = wr.status; snprintf(keep, sizeof keep, "%s", wr.err); }
                if (wr.written.empty()) return;
                f = wr.written.front();
                wr.written.pop_front();
            }
            ++done;
            if (o.on_frame_complete && o.on_frame_complete(f, total, o.user) != 0) cancelled = true;      /* :123-125, :76 */
        }
    };
    /* the oldest frame in flight: wait, export where it lives, copy 3 B/pixel back, hand it to the writer */
    auto finish_oldest = [&]() {
        const Pending f = pending.front();
        pending.pop_front();
        int s = fr_node_wait_frame(nd, f.ticket);
        if (s != FR_OK) { fail(s); return; }
        if (rc != FR_OK) return;
        uint8_t* hb = nullptr;
        {
            std::unique_lock<std::mutex> lk(wr.m);
            wr.cv.wait(lk, [&] { return !wr.free_bufs.empty(); });
            hb = wr.free_bufs.front();
            wr.free_bufs.pop_front();
        }
        hipError_t e = hipSetDevice(nd->devices[f.root]);
        if (e == hipSuccess) {
            s = fr_export_rgb8_async(nd->ctx[f.root][0], d_rgba[f.set][f.root], W, H, d_rgb8[f.set][f.root], 1, (void*)xs[f.root]);
            if (s != FR_OK) fail(s);
            else {
                e = hipMemcpyAsync(hb, d_rgb8[f.set][f.root], npx * 3, hipMemcpyDeviceToHost, xs[f.root]);
                if (e == hipSuccess) e = hipStreamSynchronize(xs[f.root]);
            }
        }
        if (e != hipSuccess) hip_fail("readback", e);
        {
            std::lock_guard<std::mutex> lk(wr.m);
            if (rc == FR_OK) wr.jobs.push_back({f.frame, hb});
            else wr.free_bufs.push_back(hb);
        }
        wr.cv.notify_all();
    };

    int submitted = 0;
    for (int32_t frame = first; frame < last && rc == FR_OK && !cancelled; frame += step) {
        const float time = fr_anim_frame_time(anim, frame);                                                          /* :80 */
        fr_params p;
        st = fr_anim_state_at(anim, time, base, &p);                                                                  /* :83 */
        if (st != FR_OK) { fail(st); break; }
        if (o.fractal_type_override) p.fractal_type = o.fractal_type_override - 1;
        if (o.max_iterations_override > 0) p.max_iterations = o.max_iterations_override;
        if (p.fractal_type != FR_FRACTAL_DEEP_ZOOM) p.flags |= FR_FLAG_POST_CHAIN;     /* what the rgba16f storage image holds */
        const int set = submitted % S, root = submitted % n;
        while ((int)pending.size() >= S && rc == FR_OK) finish_oldest();             /* that plane set's previous frame goes out first */
        if (rc != FR_OK) break;
        if (!d_rgba[set][root]) {
            hipError_t e = hipSetDevice(nd->devices[root]);
            if (e == hipSuccess) e = hipMalloc((void**)&d_rgba[set][root], npx * 16);
            if (e == hipSuccess) e = hipMalloc((void**)&d_rgb8[set][root], npx * 3);
            if (e == hipSuccess && !xs[root]) e = hipStreamCreateWithFlags(&xs[root], hipStreamNonBlocking);
            if (e != hipSuccess) { hip_fail("hipMalloc", e); break; }
        }
        const fr_output out = {d_rgba[set][root], nullptr, nullptr, FR_MEM_DEVICE, FR_LAYOUT_PACKED};
        uint64_t t = 0;
        st = fr_node_submit(nd, &p, W, H, root, &out, &t);
        if (st != FR_OK) { fail(st); break; }
        pending.push_back({t, frame, root, set});
        ++submitted;
        deliver();
    }
    while (!pending.empty()) {
        if (rc == FR_OK && !cancelled) finish_oldest();
        else { (void)fr_node_wait_frame(nd, pending.front().ticket); pending.pop_front(); }
    }
    if (wr.thread.joinable()) {
        { std::lock_guard<std::mutex> lk(wr.m); wr.quit = true; }
        wr.cv.notify_all();
        wr.thread.join();
    }
    deliver();
    for (int s2 = 0; s2 < kMaxSlots; ++s2)
        for (int k = 0; k < n; ++k)
            if (d_rgba[s2][k] || d_rgb8[s2][k]) {
                (void)hipSetDevice(nd->devices[k]);
                if (d_rgba[s2][k]) (void)hipFree(d_rgba[s2][k]);
                if (d_rgb8[s2][k]) (void)hipFree(d_rgb8[s2][k]);
            }
    for (int k = 0; k < n; ++k)
        if (xs[k]) { (void)hipSetDevice(nd->devices[k]); (void)hipStreamDestroy(xs[k]); }
    for (uint8_t* b : host_bufs) (void)hipHostFree(b);
    if (frames_written) *frames_written = done;
    return rc == FR_OK ? FR_OK : fr_set_error(rc, "%s", keep);
}

/* Internal (fr_tuning.h): drives the RCCL leg on ONE device -- loads the plugin, creates a one-rank communicator on
 * `device`, sends `bytes` bytes to itself through a grouped ncclSend / ncclRecv pair on a stream and compares.  What a
 * one-GPU box can exercise of the RCCL path below a frame: library load, communicator life cycle, the grouped
 * point-to-point calls and their stream ordering (a whole frame through the same calls: "rccl_loopback").  Returns
 * FR_OK, or the failing step's status. */
extern "C" int fr_node_rccl_selftest(int device, size_t bytes, int* rccl_version)
{
    DeviceGuard guard;
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

/* Internal (fr_tuning.h): which librccl / libamdhip64 files this process has mapped (one line per file, '\n' separated),
 * for the record of which runtime the plugin bound to next to PyTorch's bundled copies. */
extern "C" int fr_node_mapped_runtimes(char* out, size_t cap)
{
    if (!out || cap == 0) return fr_set_error(FR_ERR_INVALID_ARG, "out is NULL");
    out[0] = 0;
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return fr_set_error(FR_ERR_IO, "cannot read /proc/self/maps");
    char line[4352];
    size_t used = 0;
    while (fgets(line, sizeof line, f)) {
        const char* path = strchr(line, '/');
        if (!path) continue;
        if (!strstr(path, "librccl") && !strstr(path, "libamdhip64") && !strstr(path, "libfractalrenderer_amd")) continue;
        const size_t len = strlen(path);                         /* fgets kept the '\n': entries are whole lines */
        if (len == 0 || path[len - 1] != '\n') continue;
        bool seen = false;                                        /* once per file */
        for (const char* q = out; (q = strstr(q, path)) != nullptr; ++q)
            if (q == out || q[-1] == '\n') { seen = true; break; }
        if (seen || used + len + 1 > cap) continue;
        memcpy(out + used, path, len);
        used += len;
        out[used] = 0;
    }
    fclose(f);
    return FR_OK;
}
