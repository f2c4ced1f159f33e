/*
 * fr_tuning.h -- internal: tuning knobs of the persistent queues and the survivor stream.
 *
 * NOT part of the drop-in boundary (include/fractalrenderer_amd.h): a caller of the library needs none of these.
 * They exist for tests/ (which prove that no queue geometry, refill threshold or region layout can change a pixel),
 * tools/ and A/B measurements; names and meanings may change with any build.
 *
 * fr_ctx_set_tuning(ctx, name, value); value 0 restores the automatic choice (made per launch from the frame geometry).
 *   "workgroups_per_cu"  workgroups of 256 threads launched per compute unit (tile pass)
 *   "run_max", "run_min" longest / shortest run of sub-tiles one dequeue may claim
 *   "shift_bias"         signed change of log2 of the guided-run divisor
 *   "stage_first"        iteration budget b0 of the tile pass (default ~max_iter/28 within [32, 192], or chosen from the
 *                        frame's coarse escape-count sample); an explicit value also stages frames below the automatic
 *                        max_iter threshold when max_iter >= 2 b0
 *   "pool_refill_at"     lane pool: finished lanes wait until this many are idle (default 24)
 *   "stream_run_max", "stream_run_min"  run of survivor blocks one dequeue of the lane pool may claim
 *   "stream_workgroups_per_cu"          workgroups per compute unit of the lane-pool pass
 *   "pool_items_per_wg"  lane-pool grid cap: at most one workgroup per this many sub-tiles of the frame (default 32)
 *   "probes", "stream_probes"  queue shards a wave tries before it exits, tile pass / lane-pool pass
 *                        (1..15; 0 = automatic: 1 (2 with 64 shards) for a staged or short-orbit tile pass, 4 for the lane
 *                        pool, all shards otherwise and on grids of fewer workgroups than shards)
 *   "regions"            8 or 64: regions of the survivor stream (default: as many as the tile queue has shards)
 *   "stream_rotate"      2 = survivor-stream writers rotate over the regions (equal regions), 1 = one region per XCD,
 *                        0 = automatic (= 2)
 *   "tile_pixels"        1 / 2 sub-tiles per trip of the lean tile kernel (0 = 2)
 *   "prepare"            1 = control block + coordinate tables in a launch of their own (prepare_kernel) in front of the lean
 *                        tile pass; 0 = automatic: by the first workgroups of the tile pass itself (lean_prologue), the
 *                        separate launch only on capturing streams
 *   "tile_exit"          lean tile pass, staged: a trip leaves before b0 once alive x (left + cost) < slots x left; the value
 *                        is that cost in updates (0 = automatic, 1 = off)
 *   "tile_exit_from"     ... and not before this many updates (0 = automatic)
 *   "subtile_shape"      3: 8x8 pixel sub-tiles per wave, 4: 16x4, 6: 64x1 (general tile kernel)
 *   "ssaa"               SSAA: 0 = automatic (staged wherever the lean kernels apply; whole frames above 2^29 samples in bands, row-strip
 *                        shards above it by the sample loop),
 *                        1 = the sample loop of the general tile kernel, 2 = staged wherever it applies
 *   "stripes"            the Mandelbrot shader's effects (stripe shading, orbit trap, trap-coloured interior): 0 = automatic (lean tile
 *                        pass + lane pool in their code-3 instantiations), 1 = the effects variant of the general tile kernel (tests
 *                        compare the two bitwise)
 *   "ssaa_band_samples"  staged SSAA of a whole frame: sample grids larger than this go through the scratch in bands of whole
 *                        sub-tile rows (0 = automatic: 2^29 samples; tests set it small to band small frames)
 *   "debug_prologue_epoch" tests only: sets the context's prologue epoch (28 bits), to walk it across its wrap
 *   "debug_region_blocks" caps the capacity of a survivor-stream region so that the overflow report (FR_ERR_INTERNAL)
 *                        can be exercised; 0 = the real capacity (1.5x the worst case)
 */
#ifndef FR_TUNING_H
#define FR_TUNING_H

#include "fractalrenderer_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

int fr_ctx_set_tuning(fr_ctx* ctx, const char* name, int64_t value);

/* 1 / 0: the lane pool of the context's most recent render looked / did not look for cycles (automatic "periodicity":
 * a context whose pools closed nothing skips the looking for a while, fr_device.hip pool_wants_cycle_closing); -1: that
 * render had no lane pool */
int fr_ctx_last_pool_closing(const fr_ctx* ctx);

/* RCCL leg of fr_node on ONE device (fr_node.cpp): plugin load, one-rank communicator, a grouped ncclSend / ncclRecv of
 * `bytes` bytes to itself on a stream, compared on the host.  *rccl_version receives ncclGetVersion(). */
int fr_node_rccl_selftest(int device, size_t bytes, int* rccl_version);

/* fr_node: fault injection, the one-card RCCL loopback, and fr_ctx_set_tuning names for every render context of the node.
 *   "fail_part_phase1"       k + 1: part k's next frame fails before its render is enqueued (one shot).  With the RCCL gather
 *                            the two-phase barrier then keeps EVERY part out of its ncclGroupStart ... End.
 *   "fail_part_before_send"  k + 1: part k's next RCCL frame fails after the barrier, its receives (if it is the root) posted
 *                            and its sends not: the case that used to leave the root's stream waiting for ever.  The node
 *                            answers with ncclCommAbort on every communicator.
 *   "rccl_loopback"          1, nodes of ONE part: "gather" = 2 sends that part's strips through a one-rank communicator to
 *                            itself (packed staging -> grouped ncclSend / ncclRecv -> in-place receives -> recolour of the
 *                            smooth-count payload): the whole RCCL frame path on one card.
 *   "rccl_timeout_ms"        how long a frame's wait lets an RCCL gather run before it aborts the communicators (30000) */
int fr_node_set_tuning(fr_node* node, const char* name, int64_t value);
/* 0 once the RCCL leg of the node was aborted */
int fr_node_rccl_usable(const fr_node* node);
/* the librccl / libamdhip64 / libfractalrenderer_amd files this process has mapped, one per line (from /proc/self/maps):
 * which runtime the RCCL plugin bound to next to PyTorch's bundled copies */
int fr_node_mapped_runtimes(char* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* FR_TUNING_H */
