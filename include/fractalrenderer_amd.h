/*
 * fractalrenderer_amd.h -- C ABI of the MI355X-native escape-time renderer.
 *
 * This is the drop-in boundary for the ONE hot path of franklynch/FractalRenderer:
 * the per-pixel z <- z^2 + c iteration with smooth colouring (Mandelbrot/Julia).
 * Every entry point below replaces the reference interface cited next to it.
 * Path shorthand: src/ = FractalRenderer/src/, shaders/ = FractalRenderer/shaders/.
 *
 * Conventions
 *   - plain C11, no C++ or torch types; all pointers + sizes.
 *   - every function returns an fr_status (0 = FR_OK, < 0 = error) unless noted;
 *     nothing aborts (the reference's VK_CHECK aborts, src/vk/vk_types.h:143-150).
 *     fr_last_error() returns a thread-local message for the last failure.
 *   - there is NO CPU fallback: a render call without a usable HIP device fails
 *     with FR_ERR_NO_DEVICE.
 *   - image layout: row-major, row 0 = smallest pixel y (as gl_GlobalInvocationID.y,
 *     shaders/mandelbrot.comp:215; no vertical flip), RGBA f32, alpha = 1.
 */
#ifndef FRACTALRENDERER_AMD_H
#define FRACTALRENDERER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_VERSION_MAJOR 1
#define FR_VERSION_MINOR 1   /* 1.1: frames in flight on a node (fr_node_submit, fr_node_wait_frame, "slots", "lanes") */

typedef enum fr_status {
    FR_OK               =  0,
    FR_ERR_INVALID_ARG  = -1,   /* NULL pointer, w/h == 0, max_iter outside [1, 2^24], zoom 0/non-finite ... */
    FR_ERR_NO_DEVICE    = -2,   /* no HIP device / device ordinal out of range */
    FR_ERR_HIP          = -3,   /* a HIP runtime call failed (message has the hipError string) */
    FR_ERR_UNSUPPORTED  = -4,   /* fractal type outside the hot path (Mandelbrot, JuliaSet, Deep_Zoom are in) */
    FR_ERR_IO           = -5,   /* .franim file could not be read / written */
    FR_ERR_PARSE        = -6,   /* .franim JSON malformed or a required key is missing */
    FR_ERR_NOMEM        = -7,
    FR_ERR_INTERNAL     = -8    /* the library caught itself out (a survivor stream overflowed): the frame is incomplete */
} fr_status;

/* FractalType, src/fractal_state.h:6-14 (same numeric values).  Mandelbrot and JuliaSet are the
 * hot path; BurningShip (shaders/burning_ship.comp) is the same loop with z = abs(z) before the
 * square; Deep_Zoom is the reference's perturbation shader (shaders/test_deep_zoom.comp), restated
 * with its fp32 float-float arithmetic; Mandelbulb and Phoenix return FR_ERR_UNSUPPORTED. */
typedef enum fr_fractal_type {
    FR_FRACTAL_MANDELBROT   = 0,
    FR_FRACTAL_JULIA        = 1,
    FR_FRACTAL_BURNING_SHIP = 2,
    FR_FRACTAL_MANDELBULB   = 3,
    FR_FRACTAL_PHOENIX      = 4,
    FR_FRACTAL_DEEP_ZOOM    = 5
} fr_fractal_type;

/* Arithmetic type of the iteration.  The reference computes in fp32 only
 * (shaders/mandelbrot.comp:147-170) after narrowing its double viewport
 * (src/compute_effect_manager.h:85-90); FR_PRECISION_F64 keeps the doubles. */
typedef enum fr_precision {
    FR_PRECISION_F32 = 0,
    FR_PRECISION_F64 = 1
} fr_precision;

/* fr_params.flags */
#define FR_FLAG_POST_CHAIN   0x1u  /* apply enhance_color -> aces_tonemap -> pow(1/2.2)
                                      (shaders/mandelbrot.comp:233-235, shaders/julia.comp:330-337);
                                      default is the LINEAR colour, before that chain */

/*
 * fr_params -- the hot-path fields of FractalState (src/fractal_state.h:16-91),
 * i.e. the union of what ComputeEffect::update_from_state packs for Mandelbrot
 * (src/compute_effect_manager.h:84-113) and Julia (:115-140).  centre/zoom stay
 * double: the fp64 kernels need what the reference throws away at :86-88.
 * fr_params_default() fills the FractalState member initialisers.
 */
typedef struct fr_params {
    int32_t fractal_type;          /* fr_fractal_type                                   */
    int32_t precision;             /* fr_precision                                      */
    double  center_x;              /* src/fractal_state.h:18   default -0.5             */
    double  center_y;              /*                   :19    default  0.0             */
    double  zoom;                  /*                   :20    default  3.0  (view height in c-plane units) */
    int32_t max_iterations;        /*                   :21    default  256             */
    float   bailout;               /*                   :36    default  4.0; test is |z|^2 > bailout^2 */
    double  julia_c_real;          /*                   :29    default -0.7f  (float in the reference)  */
    double  julia_c_imag;          /*                   :30    default  0.27015f        */
    int32_t antialiasing_samples;  /*                   :37    default  1  (n -> n x n samples)         */
    int32_t palette_mode;          /*                   :40    default  0               */
    float   color_offset;          /*                   :41    default  0               */
    float   color_scale;           /*                   :42    default  1               */
    int32_t interior_style;        /*                   :47    default  0               */
    int32_t orbit_trap_enabled;    /*                   :48    default  0 (bool)        */
    float   orbit_trap_radius;     /*                   :49    default  0.5             */
    int32_t stripe_enabled;        /*                   :50    default  0 (bool)        */
    float   stripe_density;        /*                   :51    default  10              */
    float   color_brightness;      /*                   :77    default  1               */
    float   color_saturation;      /*                   :78    default  1               */
    float   color_contrast;        /*                   :79    default  1               */
    uint32_t flags;                /* FR_FLAG_*                                         */
    int32_t use_perturbation;      /*                   :86    default  0 (bool).  Deep_Zoom only: 1 = the fp64
                                      reference orbit at the centre is computed (as prepare_deep_zoom_rendering
                                      does every frame, src/vk_engine.cpp:215-251) and the shader perturbs
                                      around it; 0 = empty orbit, plain fp32 iteration                  */
} fr_params;

/* ---- parameters ------------------------------------------------------------------ */

/* FractalState{} member initialisers, src/fractal_state.h:18-51,77-79;
 * fractal_type = Mandelbrot, precision = FR_PRECISION_F64, flags = 0. */
int fr_params_default(fr_params* p);

/* FractalState::reset(), src/fractal_state.h:135-153: centre (-0.5,0), zoom 1.5 (NOT 3.0),
 * max_iterations 256, brightness/saturation/contrast 1; other fields untouched. */
int fr_params_reset(fr_params* p);

/* Validation the reference does not do (it patches or aborts, src/compute_effect_manager.h:335-345).
 * width,height > 0; width*height < 2^31; max_iterations in [1, 2^24] (ints travel as floats
 * through the push constants, exact to 2^24, shaders/mandelbrot.comp:18); zoom finite and != 0;
 * centre finite; bailout finite > 0; antialiasing_samples in [0, 16]; fractal/precision known. */
int fr_params_validate(const fr_params* p, uint32_t width, uint32_t height);

/* ComputeEffect::update_from_state, src/compute_effect_manager.h:84-113 (Mandelbrot) and
 * :115-140 (Julia): the 80-byte ComputePushConstants block (:11-17) as 20 floats,
 * bit-for-bit (every field static_cast<float>). */
int fr_pack_push_constants(const fr_params* p, float out[20]);

/* ---- context ---------------------------------------------------------------------- */

typedef struct fr_ctx fr_ctx;   /* owns the device ordinal, a stream, timing events and the
                                   tile-queue words; not re-entrant, distinct contexts may
                                   run concurrently (reference: single thread, src/vk_engine.cpp) */

/* Replaces ComputeEffectManager's constructor + init_pipelines
 * (src/compute_effect_manager.cpp:19-60,120-140): binds a HIP device; kernels are
 * precompiled for gfx950 in this library, there is nothing to load at run time. */
int  fr_ctx_create(int device_ordinal, fr_ctx** out);
void fr_ctx_destroy(fr_ctx* ctx);

/* ---- render ----------------------------------------------------------------------- */

typedef enum fr_memory {
    FR_MEM_DEVICE = 0,    /* pointers are HIP device pointers on ctx's device (no copies)   */
    FR_MEM_HOST   = 1     /* pointers are host memory; the library stages through device
                             scratch it owns and copies back (PCIe-inclusive path)          */
} fr_memory;

/* Where a part of a row-strip sharding (fr_shard) stores its rows. */
typedef enum fr_layout {
    FR_LAYOUT_PACKED = 0, /* the planes hold only this part's rows, packed densely in strip order (fr_shard_rows() rows) */
    FR_LAYOUT_FRAME  = 1  /* the planes are WHOLE-FRAME planes (height rows) and the part stores its rows in place: the parts
                             of a frame may then share one set of planes -- on one device, or on a peer device whose memory
                             is mapped (hipDeviceEnablePeerAccess): the gather of a multi-GPU frame fused into the kernels'
                             stores (fr_node_render).  FR_MEM_DEVICE only. */
} fr_layout;

/* Output planes of one render.  rgba is required unless nu or iter is given. */
typedef struct fr_output {
    float*   rgba;     /* rows*W*4 f32, linear colour (or post-chained with FR_FLAG_POST_CHAIN) */
    void*    nu;       /* rows*W smooth iteration count of sample (0,0): double for
                          FR_PRECISION_F64, float for FR_PRECISION_F32; max_iterations
                          for interior pixels.  NULL to skip                                */
    int32_t* iter;     /* rows*W index i of the escaping update (shaders/mandelbrot.comp:157-170),
                          max_iterations for interior.  NULL to skip                        */
    int32_t  memory;   /* fr_memory                                                         */
    int32_t  layout;   /* fr_layout (0 = packed part rows; occupies what was padding before 1.0) */
} fr_output;

/* Row-strip sharding of one frame over the GPUs of a node: the frame's rows are cut
 * into strips of rows_per_strip rows, dealt round-robin: strip s belongs to part
 * (s % nparts).  Part `part` renders only its strips, packed densely in strip order,
 * into its fr_output (fr_shard_rows() rows).  {0,1,0} = the whole frame. */
typedef struct fr_shard {
    uint32_t part;
    uint32_t nparts;
    uint32_t rows_per_strip;   /* 0 with nparts == 1 means "whole frame" */
} fr_shard;

/* number of rows part `part` owns / first-row table helpers (host-side arithmetic only) */
uint32_t fr_shard_rows(const fr_shard* s, uint32_t height);
/* global row index of local row `local_row` of this part; UINT32_MAX if out of range */
uint32_t fr_shard_global_row(const fr_shard* s, uint32_t height, uint32_t local_row);

/*
 * fr_render -- the reference's render(viewport, max_iter, out_buffer) surface:
 *   AnimationRenderer::RenderFrameCallback  bool(const FractalState&, uint32_t width,
 *   uint32_t height, const std::string& path)  src/animation_renderer.h:41-48, whose body
 *   VulkanEngine::render_animation_frame (src/vk_engine.cpp:1181-1418) reaches
 *   ComputeEffectManager::dispatch (src/compute_effect_manager.h:435-468) ->
 *   vkCmdDispatch(ceil(W/16), ceil(H/16), 1) of shaders/mandelbrot.comp / julia.comp.
 * Synchronous: returns after the planes are complete (reference: immediate_submit waits
 * on its fence, src/vk_engine.cpp:2319-2321).
 */
int fr_render(fr_ctx* ctx, const fr_params* p, uint32_t width, uint32_t height,
              const fr_output* out);

/* Same, restricted to one part of a row-strip sharding (multi-GPU row bands). */
int fr_render_shard(fr_ctx* ctx, const fr_params* p, uint32_t width, uint32_t height,
                    const fr_shard* shard, const fr_output* out);

/* Asynchronous form for frame pipelining: enqueues on `hip_stream` (a hipStream_t passed
 * as void*; NULL = the context's own stream) and returns without waiting.  Device memory only.
 * STEADY STATE it is launch-only (kernel launches -- a small one that clears the control block and writes the frame's
 * coordinate tables, then the render's -- + two event records: no allocation, no host synchronisation, capturable
 * into a hipGraph).  What is not steady state:
 *   - the first render of a geometry LARGER than any before it on this context grows the context's
 *     survivor-stream scratch and coordinate tables (hipFree + hipMalloc), and the first render of a new (W, H, fractal,
 *     precision) checks on the host that the divide-free viewport map is exact for every column and row
 *     (a loop over W + H numerators, cached for 8 geometries).  fr_ctx_reserve() does both ahead of time;
 *   - Deep_Zoom with use_perturbation recomputes the fp64 reference orbit on the host for every frame, as
 *     the reference does (src/vk_engine.cpp:215-251), and waits for `hip_stream` before reusing its pinned
 *     upload buffer: never launch-only.
 * A failure the device reports later (FR_ERR_INTERNAL) surfaces at the next call on the context or
 * through fr_ctx_check(). */
int fr_render_shard_async(fr_ctx* ctx, const fr_params* p, uint32_t width, uint32_t height,
                          const fr_shard* shard, const fr_output* out, void* hip_stream);

/* Pre-sizes everything a render of (params, width, height, shard) allocates or caches on first use, so
 * that later fr_render_shard_async calls of this or any smaller geometry are launch-only (see there).
 * Synchronises the context's own stream.  New design: the reference allocates its image and staging
 * buffer per call (src/vk_engine.cpp:1197-1221,1268-1272). */
int fr_ctx_reserve(fr_ctx* ctx, const fr_params* p, uint32_t width, uint32_t height, const fr_shard* shard);

/* FR_OK, or FR_ERR_INTERNAL if a render that has completed on this context since the last call lost
 * pixels (cleared by the call).  The synchronous entry points check by themselves; a caller of the
 * _async forms calls this after it has synchronised its stream. */
int fr_ctx_check(fr_ctx* ctx);

/* Device time of the most recent render's kernels on this context, from a HIP event pair recorded around its launches on
 * the launch stream (blocks until they are done).  Needs the option "timing" = 1, set BEFORE that render: the pair is not
 * recorded by default since 1.1 (two timed event records cost a frame ~4.7 us -- 0.6 % of a 4096^2 / 1024 frame, 10 % of a
 * 1080p / 256 one).  < 0 if nothing was rendered with it yet. */
float fr_ctx_last_kernel_ms(fr_ctx* ctx);

/* Options by name; value 0 restores the automatic choice (made per launch from the frame geometry).  None of them can
 * change a pixel (tests/test_gpu_parity.py::test_tuning_variants_are_bit_identical).
 *   "periodicity"   -1 = off, 0 = automatic (ON), 1 = on, N > 1 = on with a first snapshot window of N iterations.
 *                   Cycle closing: the kernels keep, per lane, the orbit state at the wave's last snapshot; a lane whose
 *                   state returns to it is on a cycle, can never escape, and is retired as interior at once instead of
 *                   being iterated to max_iter.  Exact, not a heuristic: the update is a deterministic function of (z, c),
 *                   so every plane stays byte-identical (tests/test_gpu_parity.py::test_periodicity_never_changes_a_pixel);
 *                   what changes is the number of iterations executed -- the reference's shaders iterate every interior
 *                   sample to max_iter (C2: 1.4x fewer, a filled Julia set 2.8x).  Automatic (0): a context whose lane pools
 *                   retired less than an eighth of their records by closed cycles -- a Julia dust -- renders its next 14
 *                   frames of the same kind without looking, then looks once more (no synchronisation: the verdict of a
 *                   frame travels with the next frame's first launch); 1 / N: always look.
 *                   Not where a sample's final z is read (stripe shading), in the Burning Ship's effects variants or Deep_Zoom.  bench.py's headline switches it OFF so that its roofline
 *                   is quoted on the reference's iteration count.
 *   "staging"       0 = automatic, 1 = single pass (every sample runs to max_iter in the tile kernel), 3 = tile pass for
 *                   the first b0 iterations + ONE lane-pool pass over the compacted survivors, whatever max_iter is.
 *                   Automatic: 3 where it applies (not with the Burning Ship's trap / stripe effects; a supersampled frame's
 *                   sample grid is rendered as a frame of its own and takes the same choice) and pays off -- a Julia set from
 *                   max_iterations 256, fp64 from 512 (384 on frames above 2^23 pixels), fp32 from 768 (512); frames of up
 *                   to 2^19 pixels: fp64 from 1024 (1536 up to 2^18 pixels), fp32 from 1536, a Julia set of up to 2^20 pixels
 *                   from 512 -- the second launch and the
 *                   pool's ramp cost a small frame more than they save (profiles/r04_staging_crossover.txt,
 *                   r04_small_frame_staging.txt).
 *   "shards"        8 or 64: shards of the work queue (each has ONE head word that its waves update with returning
 *                   atomics, ~15 ns apart).  Automatic: 64 (8 per XCD) for launches whose waves stop at their home shards
 *                   on grids of >= 256 workgroups over frames of >= 4096 sub-tiles, else 8.
 *   "tile_kernel"   1 = the general tile kernel; 0 = automatic: the LEAN tile kernel (coordinate tables written by its own
 *                   first workgroups -- by a small launch in front of it on capturing streams --, two 8x8 sub-tiles per wave
 *                   and trip) for every one-sample render (the Burning Ship's trap / stripe effects excepted) whose row
 *                   strips, if sharded, are whole sub-tile rows.
 *   "timing"        1 = record a HIP event pair around every render (fr_ctx_last_kernel_ms, fr_node_last_kernel_ms); 0 = off, the
 *                   default since 1.1.
 *   "diag_buffer"   device pointer to 4 x uint64 per wave (t_start, t_end in 100 MHz ticks, items processed, dequeues);
 *                   0 disables.  "diag_stride" = uint64 words between the tile pass's and the lane pool's regions.
 * Changed in 1.0 (INTEGRATION.md lists the breaks): "periodicity" 0 means automatic = ON since 0.2 (it meant off in
 * 0.1; -1 is off); "staging" 2 / 4 and the options "pool", "stage_ratio" are accepted and ignored (the schedules they
 * selected are gone); the queue / stream tuning names moved to the internal fr_ctx_set_tuning
 * (fractalrenderer_amd/csrc/fr_tuning.h). */
int fr_ctx_set_option(fr_ctx* ctx, const char* name, int64_t value);

/* Workgroups per launch of the most recent render on this context (bits 0-15) and its number of
 * stages (bits 16-..: 1 = single pass). */
int fr_ctx_last_grid(fr_ctx* ctx);

/* Number of compute units of the context's device (hipDeviceProp_t.multiProcessorCount). */
int fr_ctx_compute_units(fr_ctx* ctx);

/* Waits for everything this context has enqueued on its OWN stream (fr_render_shard_async with a NULL stream, the
 * exports, fr_colorize_async with a NULL stream) and returns fr_ctx_check()'s verdict. */
int fr_ctx_synchronize(fr_ctx* ctx);

/* ---- frames over the GPUs of a node (BASELINE.json north_star: "tiled across the 8 GPUs of one node as disjoint row
 * bands with a final RCCL gather over xGMI") -----------------------------------------------------------------------------
 * New design, no reference counterpart: the reference renders on the one GPU it picked (src/vk_engine.cpp:608).  What it
 * replaces is the same RenderFrameCallback / dispatch surface as fr_render (src/animation_renderer.h:41-48,
 * src/compute_effect_manager.h:435-468), for a caller that owns several devices and, like the reference's caller
 * (AnimationRenderer::start_render, src/animation_renderer.cpp:75-127), renders a sequence of frames: ONE process, one
 * host worker thread per device, up to 4 render contexts ("lanes") per device, a ring of frame slots.
 *
 * fr_node_create: devices[0..n-1] are HIP device ordinals (n <= 16).  They may repeat: n parts on ONE device are n
 *   concurrent renders on that device -- and how the band arithmetic and the in-place stores are tested on one card.
 *   The caller's current device is left as it was, by this and every other fr_node_* call.
 * A frame: its rows are cut into strips (fr_shard: strips of R rows dealt round-robin, part k on devices[k]; "layout" = 1
 *   makes them n contiguous bands), every part renders its rows concurrently, and the frame is assembled in `out`, whose
 *   planes live on devices[root] (FR_MEM_DEVICE) or in host memory (FR_MEM_HOST: staged through a frame buffer on
 *   devices[root], copied back when the frame is waited for).  No collective on the compute path; the one exchange is the
 *   gather, by
 *     FR_GATHER_PEER  the kernels of every part store straight into the root's planes (FR_LAYOUT_FRAME), mapped into the
 *                     other devices' address spaces with hipDeviceEnablePeerAccess: the gather is fused into the stores,
 *                     nothing is staged and nothing waits for a transfer phase (last sentence);
 *     FR_GATHER_RCCL  every part renders into a buffer on its own device and ships its strips to the root with grouped
 *                     ncclSend / ncclRecv (RCCL, point to point over xGMI's full mesh) on a communication stream of its own,
 *                     ordered behind its render by an event, received in place.  For the plain colourings the parts render
 *                     and ship only the smooth-count plane (8 B/pixel instead of 16) and the root recolours the assembled
 *                     frame (fr_colorize_async), bit-identically ("payload").  Needs distinct devices (one communicator
 *                     rank per device); librccl is loaded, through libfractalrenderer_amd_rccl.so, by the first frame that
 *                     uses it.  Fail-safe: every part enqueues its render and reports before ANY part posts a send or a
 *                     receive, so the two sides always match; a failure after that point, or a gather that does not finish
 *                     in 30 s, aborts every communicator (ncclCommAbort) instead of leaving a stream waiting, the frame
 *                     reports the error, and the node carries on with FR_GATHER_PEER where the devices can map each other.
 *   Planes are byte-identical to fr_render's, whatever n, root, layout, gather, slots and lanes.
 * fr_node_render: one frame, synchronously (= fr_node_submit + fr_node_wait_frame).
 * fr_node_submit: validates, takes a frame slot, hands the frame to the workers and returns its ticket (1, 2, ...) without
 *   waiting for anything: up to "slots" frames are IN FLIGHT at once, frame t's parts running on lane t % "lanes" of every
 *   device, so that frame t + 1's ramp-up fills frame t's drain (a 1/8 share of a frame rendered alone costs 1.8x its size).
 *   A submit that finds its slot (ticket % slots) still holding an earlier frame first completes that frame; its verdict is
 *   kept for its ticket (the last 64 are).  `root` is per frame: a caller rotates it (frame f to root f % n) so that every
 *   link carries gather traffic; FR_ROOT_ROTATE does that for FR_MEM_HOST planes.  `out`'s planes must stay valid until the
 *   frame has been waited for.  fr_node_render_async is fr_node_submit without the ticket.
 * fr_node_wait_frame(ticket): waits for that frame alone (its parts' device work; FR_MEM_HOST: + the copy back), in any
 *   order, and returns its verdict.  fr_node_wait: waits for every frame in flight, oldest first, and returns the first
 *   failure nobody has been told about yet.
 * One caller thread at a time drives a node (as one fr_ctx); distinct nodes are independent.
 * fr_node_set_option (refused while frames are in flight): "gather" (fr_gather; 0 = automatic: in-place stores wherever
 *   the devices can map each other -- the gather every one-card test compares bitwise --, RCCL where they cannot; an
 *   explicit 2 whose plugin or communicators cannot be had falls back to 1 if possible: fr_node_last_gather tells),
 *   "layout" (0 = interleaved strips, balanced within a frame; 1 = contiguous bands), "rows_per_strip" (0 = automatic:
 *   32, a whole number of sub-tile rows), "payload" (0 = automatic: the smooth-count plane where fr_colorize_supported,
 *   1 = always the planes asked for), "slots" (1..8 frames in flight, 0 = automatic: 2), "lanes" (1..4 render contexts
 *   per part, created on first use, 0 = automatic: 2), and every fr_ctx_set_option name, applied to all contexts. */
typedef struct fr_node fr_node;
typedef struct fr_anim fr_anim;   /* (.franim animations, below) */
typedef enum fr_gather { FR_GATHER_AUTO = 0, FR_GATHER_PEER = 1, FR_GATHER_RCCL = 2 } fr_gather;
#define FR_ROOT_ROTATE (-1)           /* fr_node_submit / fr_node_render: root = (ticket - 1) % n; FR_MEM_HOST planes */

int  fr_node_create(const int* devices, int n, fr_node** out);
void fr_node_destroy(fr_node* node);
int  fr_node_device_count(const fr_node* node);
int  fr_node_set_option(fr_node* node, const char* name, int64_t value);
int  fr_node_render(fr_node* node, const fr_params* p, uint32_t width, uint32_t height, int root, const fr_output* out);
int  fr_node_submit(fr_node* node, const fr_params* p, uint32_t width, uint32_t height, int root, const fr_output* out,
                    uint64_t* ticket);
int  fr_node_wait_frame(fr_node* node, uint64_t ticket);
int  fr_node_render_async(fr_node* node, const fr_params* p, uint32_t width, uint32_t height, int root, const fr_output* out);
int  fr_node_wait(fr_node* node);
/* frames submitted and not waited for (or completed by a later submit) yet */
int  fr_node_in_flight(const fr_node* node);
/* the gather the most recently submitted frame uses (fr_gather, never AUTO), or < 0 before the first one */
int  fr_node_last_gather(const fr_node* node);
/* device time of part `part`'s kernels of the most recently submitted frame (fr_ctx_last_kernel_ms of its context: needs the
 * option "timing" = 1 on the node) */
float fr_node_last_kernel_ms(fr_node* node, int part);

/* AnimationRenderer::start_render (src/animation_renderer.cpp:26-152) over the GPUs of a node: for frame = first, first +
 * step, ...: time = frame / float(target_fps) (:80), state = interpolate(time) (:83; fr_anim_state_at with `base`, whose
 * fractal_type / precision / flags are not animated), the frame rendered over the node's parts with the shader's post chain
 * (fr_node_submit, roots rotating, up to "slots" frames in flight), and the RenderFrameCallback's tail
 * (src/vk_engine.cpp:1266-1381: readback, second ACES + gamma, u8, flip, PNG) done where the frame was assembled -- the
 * 8-bit export on the root's device, 3 bytes per pixel back, "<folder>/frame_%06d.png" (:86-88) written by a thread of
 * its own while the devices render the next frames.  Files are byte-identical to fr_render_frame_png's.  Fewer than 2
 * keyframes: FR_ERR_INVALID_ARG ("Need at least 2 keyframes to render", :35-42).  on_frame_complete (:123-125) is called on
 * the CALLER's thread for every frame whose file is complete, in order; returning non-zero requests cancellation
 * (cancel_requested, :76): no further frame is started, the call returns FR_OK with *frames_written short of the plan. */
typedef int (*fr_frame_callback)(int32_t frame, int32_t total_frames, void* user);
typedef struct fr_anim_render_options {
    int32_t width, height;                /* 0: the animation's export_width / export_height                        */
    int32_t first_frame, frame_count;     /* frame_count 0: to the end (total = int(duration * target_fps), :48)     */
    int32_t frame_step;                   /* every n-th frame (a sub-sampled sweep); 0 / 1: every frame              */
    int32_t max_iterations_override;      /* 0: the keyframes'                                                       */
    int32_t fractal_type_override;        /* 0: base's; else fr_fractal_type + 1 (the engine's current_fractal_type) */
    fr_frame_callback on_frame_complete;  /* may be NULL                                                             */
    void* user;
    int32_t raw_fd;                       /* > 0: no PNG files -- every frame goes to this file descriptor as packed RGB24,
                                           * top row first, in frame order (fr_write_raw_rgb24: the stdin of `ffmpeg -f rawvideo
                                           * -pix_fmt rgb24 -s WxH -framerate F -i -`); the bytes are the PNG files' pixels.
                                           * output_folder may then be NULL.  0: PNG files (the reference's behaviour)    */
    int32_t reserved;                     /* 0 */
} fr_anim_render_options;
int  fr_node_render_animation(fr_node* node, const fr_anim* anim, const fr_params* base, const fr_anim_render_options* options,
                              const char* output_folder, int32_t* frames_written);

/* ---- recolour from the smooth-count plane (multi-GPU exchange payload) ------------------------
 * New design, no reference counterpart (the reference is single-GPU): for the plain colourings the
 * shaders' colour is a function of the smooth count alone (shaders/mandelbrot.comp:179-190,
 * shaders/julia.comp:243-248, shaders/burning_ship.comp:296-299), so the ranks of a row-band sharded
 * frame can ship the 8-byte (fp64) / 4-byte (fp32) nu plane over xGMI instead of the 16-byte colour
 * and the destination recolours it, bit-identically to what the render kernels write.
 *
 * fr_colorize_supported: 1 when that holds for `params` (Mandelbrot / JuliaSet / BurningShip, no
 * trap / stripe / interior-style variant, antialiasing_samples <= 1, bailout large enough that
 * nu == max_iterations identifies exactly the interior samples, and for FR_PRECISION_F32 max_iterations
 * <= 2^22 so that a float nu just below max_iterations does not round up to it), else 0.
 * fr_colorize_async: nu (n_pixels doubles for FR_PRECISION_F64, floats for F32) -> rgba (n_pixels x
 * RGBA f32), both device pointers, enqueued on hip_stream (NULL: the context's stream), no host sync;
 * FR_FLAG_POST_CHAIN applies the post chain as fr_render does.  FR_ERR_UNSUPPORTED when not supported. */
int fr_colorize_supported(const fr_params* params);
int fr_colorize_async(fr_ctx* ctx, const fr_params* params, uint64_t n_pixels, const void* nu, float* rgba,
                      void* hip_stream);

/* ---- 8-bit export (reference a9) ------------------------------------------------------ */

/* The CPU loop of VulkanEngine::render_animation_frame, src/vk_engine.cpp:1344-1371, on the
 * GPU: per channel ACES tonemap -> pow(1/2.2) -> (uint8)(v*255) with a vertical flip
 * (:1359), RGBA f32 in -> packed RGB8 out (rows*W*3).  through_half != 0 first rounds each
 * channel to fp16 as the reference's rgba16f storage image does (shaders/mandelbrot.comp:5). */
int fr_export_rgb8(fr_ctx* ctx, const float* rgba, uint32_t width, uint32_t height,
                   uint8_t* rgb8, int32_t memory, int32_t through_half);

/* 16-bit export of VulkanEngine::export_print_quality, src/vk_engine.cpp:2054-2073: per channel
 * clamp(v, 0, 1) -> (uint16)(v*65535) with a vertical flip and NO second tonemap (unlike the 8-bit path),
 * RGBA f32 in -> packed RGB16 (host-endian uint16, rows*W*3) out. */
int fr_export_rgb16(fr_ctx* ctx, const float* rgba, uint32_t width, uint32_t height,
                    uint16_t* rgb16, int32_t memory, int32_t through_half);

/* Stream ordering of the two exports above (and of fr_colorize_async with a NULL stream): they run on the
 * context's own stream, ordered behind the most recent render of THIS context whatever stream that was
 * enqueued on; a plane produced by anything else (another context, torch) must be complete, or the caller
 * uses the _async forms below on the producing stream.  The _async forms: device memory only, enqueued on
 * `hip_stream` (NULL = the context's stream), no host synchronisation. */
int fr_export_rgb8_async(fr_ctx* ctx, const float* rgba, uint32_t width, uint32_t height, uint8_t* rgb8,
                         int32_t through_half, void* hip_stream);
int fr_export_rgb16_async(fr_ctx* ctx, const float* rgba, uint32_t width, uint32_t height, uint16_t* rgb16,
                          int32_t through_half, void* hip_stream);

/* ---- frame output (reference f3) ---------------------------------------------------------- */

typedef struct fr_png_text { const char* key; const char* text; } fr_png_text;   /* one tEXt chunk */

/* PNG writer (colour type RGB, no interlace, filter None, zlib deflate).  bit_depth 8: the file
 * stbi_write_png produces pixel-wise (src/vk_engine.cpp:1374-1381); bit_depth 16: host-endian uint16
 * samples written big-endian, compression level 9 (src/vk_engine.cpp:2114-2208).  print_metadata != 0
 * adds what the print export adds: gAMA 1/2.2, sRGB perceptual, pHYs 300 dpi, tIME; `texts` become
 * uncompressed tEXt chunks.  The deflate is band-parallel (FR_PNG_THREADS, default: all cores); for 8-bit files the
 * environment variable FR_PNG_LEVEL = 1..9 picks the zlib level (default 6: an animation export is bound by it). */
int fr_write_png(const char* path, uint32_t width, uint32_t height, int32_t bit_depth, const void* rgb,
                 const fr_png_text* texts, int32_t ntexts, int32_t print_metadata);

/* One packed RGB24 frame to a file descriptor (the stdin pipe of an encoder such as
 * `ffmpeg -f rawvideo -pix_fmt rgb24 -s WxH -i -`, src/video_encoder.cpp:195-224), retrying short writes.  A reader that
 * has gone away is FR_ERR_IO: SIGPIPE is blocked in the calling thread for the duration of the call. */
int fr_write_raw_rgb24(int fd, const uint8_t* rgb8, uint32_t width, uint32_t height);

/* "<folder>/frame_%06d.png", src/animation_renderer.cpp:86-88 */
int fr_frame_path(const char* folder, int32_t frame, char* out, size_t cap);

/* The body of the RenderFrameCallback, bool(const FractalState&, width, height, path)
 * (src/animation_renderer.h:41-48 -> VulkanEngine::render_animation_frame, src/vk_engine.cpp:1181-1418),
 * end to end on the GPU: render with the shader's post chain (what the rgba16f storage image holds),
 * round to fp16, second ACES + gamma, u8, vertical flip (:1344-1371), copy 3 B/pixel back, write the PNG. */
int fr_render_frame_png(fr_ctx* ctx, const fr_params* p, uint32_t width, uint32_t height, const char* path);

/* ---- .franim animations --------------------------------------------------------------- */

typedef struct fr_anim fr_anim;

/* InterpolationType, src/animation_system.h:8-14 */
typedef enum fr_interp {
    FR_INTERP_LINEAR = 0, FR_INTERP_EASE_IN_OUT = 1, FR_INTERP_EASE_IN = 2,
    FR_INTERP_EASE_OUT = 3, FR_INTERP_EXPONENTIAL = 4
} fr_interp;

/* Animation metadata, src/animation_system.h:24-35 */
typedef struct fr_anim_info {
    float   duration;
    int32_t loop;
    int32_t target_fps;
    int32_t export_width;
    int32_t export_height;
    int32_t keyframe_count;
} fr_anim_info;

/* One Keyframe (src/animation_system.h:16-22) restricted to the fields
 * AnimationSystem::load_from_file reads (src/animation_system.cpp:291-301). */
typedef struct fr_keyframe {
    float   time;
    int32_t interp_type;
    fr_params state;
} fr_keyframe;

/* AnimationSystem::load_from_file, src/animation_system.cpp:275-313: parses the .franim
 * JSON.  Required keys exactly as the reference reads them: top level name, description,
 * duration, loop, target_fps, export_width, export_height, keyframes[]; per keyframe time,
 * interp_type, center_x, center_y, zoom, max_iterations, palette_mode, color_offset,
 * color_scale.  The extra keys the reference's writer emits (:246-255) are accepted and
 * loaded when present.  Keyframes keep file order (the reference does not sort on load). */
int  fr_anim_load(const char* path, fr_anim** out);
int  fr_anim_parse(const char* json, size_t len, fr_anim** out);
void fr_anim_free(fr_anim* a);

/* AnimationSystem::save_to_file, src/animation_system.cpp:221-273 (same 19 keyframe keys). */
int  fr_anim_save(const fr_anim* a, const char* path);

int  fr_anim_get_info(const fr_anim* a, fr_anim_info* info);
int  fr_anim_get_keyframe(const fr_anim* a, int32_t index, fr_keyframe* out);
/* name / description strings (owned by the animation) */
const char* fr_anim_name(const fr_anim* a);
const char* fr_anim_description(const fr_anim* a);

/* Empty animation (AnimationSystem ctor, src/animation_system.cpp:7-10: duration 10) and
 * AnimationSystem::add_keyframe (:12-23: append, stable-sort by time, extend duration to
 * time+1 when time > duration). */
int  fr_anim_create(fr_anim** out);
int  fr_anim_add_keyframe(fr_anim* a, float time, const fr_params* state, int32_t interp_type);

/* AnimationSystem::interpolate(time), src/animation_system.cpp:82-181 (+ find_keyframe_pair
 * :183-197, easing :199-212).  `base` is the live FractalState returned when there are no
 * keyframes (:83); fields the reference leaves at FractalState defaults because `result` is
 * default-constructed (:125) come out as fr_params_default() values.  fractal_type/precision/
 * flags are not animated: they are copied from `base`. */
int  fr_anim_state_at(const fr_anim* a, float time, const fr_params* base, fr_params* out);

/* AnimationRenderer::start_render frame arithmetic, src/animation_renderer.cpp:48,80:
 * total_frames = int(duration * target_fps); time(frame) = frame / float(target_fps). */
int32_t fr_anim_frame_count(const fr_anim* a);
float   fr_anim_frame_time(const fr_anim* a, int32_t frame);

/* ---- deep-zoom reference orbit (reference a8) ------------------------------------------- */

/* DeepZoomManager::compute_reference_orbit fp64 loop, src/deep_zoom_system.cpp:378-424:
 * single point, host side in the reference too.  out_xy has room for max_iter (re,im)
 * pairs; *out_len receives the trimmed length. */
int fr_reference_orbit(double cx, double cy, int32_t max_iter, double* out_xy, int32_t* out_len);

/* ---- deep-zoom zoom paths ---------------------------------------------------------------------
 * DeepZoomManager's zoom-path animation, src/deep_zoom_system.cpp:454-556 (host side in the reference too): a list of
 * ZoomKeyframes (src/deep_zoom_system.h:84-89; centre and zoom are "ArbitraryFloat"s that hold a double) walked by
 * update_animation(delta_time): centre linear, zoom in log space, the view snapped onto a keyframe when its duration
 * has passed.  *orbit_dirty is set where the reference recomputes its reference orbit (:505); here the next
 * fr_render of a Deep_Zoom frame does that anyway. */
typedef struct fr_zoom_keyframe {
    double center_x, center_y, zoom;
    float  duration;               /* seconds from the previous keyframe to this one */
} fr_zoom_keyframe;
typedef struct fr_zoom_path fr_zoom_path;

int  fr_zoom_path_create(fr_zoom_path** out);
void fr_zoom_path_free(fr_zoom_path* z);
/* playZoomPath, :454-460 (n == 0: nothing to play) */
int  fr_zoom_path_play(fr_zoom_path* z, const fr_zoom_keyframe* path, int32_t n);
/* zoomTo, :462-485: from `current`'s view (a start keyframe of duration 0) to the target in `duration` seconds */
int  fr_zoom_path_zoom_to(fr_zoom_path* z, const fr_params* current, double target_x, double target_y, double target_zoom,
                          float duration);
/* update_animation, :487-531: advances by delta_time and writes centre / zoom into *state; *animating and *progress
 * mirror DeepZoomState::zoom_animating / zoom_progress (src/deep_zoom_system.h:115-116).  Output pointers may be NULL. */
int  fr_zoom_path_update(fr_zoom_path* z, float delta_time, fr_params* state, int32_t* animating, float* progress,
                         int32_t* orbit_dirty);
/* DeepZoomPresets, :575-601: 0 Seahorse (1e-6, 5 s), 1 Elephant (1e-8, 7 s), 2 Mini Mandelbrot (1e-10, 10 s) */
int  fr_zoom_preset(int32_t which, fr_zoom_keyframe* out);

/* ---- misc ---------------------------------------------------------------------------------- */

const char* fr_last_error(void);
const char* fr_status_string(int status);
void        fr_version(int* major, int* minor);

#ifdef __cplusplus
}
#endif
#endif /* FRACTALRENDERER_AMD_H */
