/*
 * cloudsky_internal.h -- the lab bench of libcloudsky.so: measurement, tuning and test entry points.
 *
 * Same library, same conventions as cloudsky.h, NOT part of the product surface: nothing here is needed to render, and some of it selects
 * kernels that are 2x slower than the default (kept for A/B).  Users: bench.py (kernel timing, the census), tests/ (texture read-back, the
 * exhaustive sqrt check, variants / schedules / segments as parity cases), tools/ (A/B scripts, the BC7 sensitivity study).  A Godot host
 * (gdext/cloudsky_gdextension.c) and a plain C client (tests/c_abi_check.c) compile against cloudsky.h alone.
 */
#ifndef CLOUDSKY_INTERNAL_H
#define CLOUDSKY_INTERNAL_H
#include "cloudsky.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement ---------------------------------------------------------------------------------
 * Times `iters` back-to-back launches of the cloud kernel alone with HIP events on the context's stream
 * (after `warmup` untimed launches) and returns the mean per-launch milliseconds.  Also fills the
 * sample counters of one launch (the kernel's own tallies). */
int csky_time_clouds(csky_ctx* ctx, const csky_cloud_params* p, int tile_w, const csky_bands* bands, int warmup,
                     int iters, float* mean_ms, csky_cloud_stats* stats);
int csky_get_cloud_stats(csky_ctx* ctx, csky_cloud_stats* stats); /* tallies of the last stats-enabled launch */
/* Per-launch timing of the cloud kernel inside the caller's own frame loop: while enabled, every csky_render_clouds* launch is
 * bracketed by a pair of HIP events recorded on the stream the kernel is launched on.  csky_get_kernel_ms waits for the launches
 * recorded since the last call (all of them: the event pool grows on demand), returns the sum of their durations and their
 * number, and resets. */
int csky_set_kernel_timing(csky_ctx* ctx, int enabled);
/* The multi-device handle's preconditions and, while csky_multi_set_timing is on, HIP-event timings of the LAST frame enqueued: per device the
 * march of its bands and (staged form) the peer copy of those bands into the first device's frame -- the "7 concurrent P2P copies" of SURVEY 8(e).
 * csky_multi_get_stats waits for the handle's work (csky_multi_sync).  march_ms / copy_ms: -1 = not measured; copy_ms 0 in the in-place form. */
#define CSKY_MULTI_STATS_MAX 16
typedef struct csky_multi_stats {
    int32_t n_devices, staged, all_peer, groups, frames_in_flight, timing;
    int32_t device_id[CSKY_MULTI_STATS_MAX];
    int32_t peer_access[CSKY_MULTI_STATS_MAX];   /* 1 = this device can store into the first device's memory (1 for the first device) */
    float march_ms[CSKY_MULTI_STATS_MAX];
    float copy_ms[CSKY_MULTI_STATS_MAX];
} csky_multi_stats;
int csky_multi_set_timing(csky_multi* m, int enabled);
int csky_multi_get_stats(csky_multi* m, csky_multi_stats* out);
int csky_get_kernel_ms(csky_ctx* ctx, float* total_ms, int* launches);
/* Kernel variant selector for A/B measurement (csky_variant_name lists them).  -1 = the default = the fastest measured
 * (CSKY_DEFAULT_VARIANT, "compact").  Unknown ids -> CSKY_ERR_INVALID. */
#define CSKY_DEFAULT_VARIANT 3
int csky_set_variant(csky_ctx* ctx, int variant);
/* Exact height-window reject (density() provably 0 above/below the cloud body for the bound weather map): on by default;
 * 0 disables it (A/B measurement, identical results). */
int csky_set_height_window(csky_ctx* ctx, int enabled);
int csky_variant_count(void);
/* Workgroup -> XCD schedule (tuning knob, results are identical): -1 = auto (see api.cpp::clouds_dev for the launch-size policy);
 * 5 = slab rows round-robin over the XCDs; 1 = contiguous eighths; 2 = natural order (all three written on the device);
 * 7 = cost feedback: every launch records a cost per workgroup (in-cloud samples) and the next launch of the same geometry
 *     and view starts its workgroups heaviest first (the first launch runs in a static order);
 *     Only the ORDER comes from the previous launch; every sample is recomputed.
 * (0, 3, 4, 6 were azimuth-wedge / horizon-first orders of round 1; 8 / 9 the 'deadline' reorder and per-workgroup adaptive ray
 *  segments of round 2: all measured, no gain, removed -- kernels.hip keeps the numbers.) */
int csky_set_schedule(csky_ctx* ctx, int mode);
/* Ray segments: the primary march of every ray is cut into `segments` pieces marched by different wavefronts of one
 * workgroup and composited front to back (T and L are associative).  0 = auto (whole rays for large launches, 2 or 4
 * step ranges for one GPU's share of a split frame, 4 interleaved step sets for tile-sized launches such as the
 * reference's 96x96 temporal tiles), 1, 2, 4 (step ranges) or 5 (4 interleaved). */
int csky_set_segments(csky_ctx* ctx, int segments);
const char* csky_variant_name(int variant);

/* Test hook: read back what csky_set_noise built on the device.  which: 0 shape layout, 1 detail layout, 2 weather layout (csky_common.h),
 * 3 / 4 the 8-bit mip chains of the large / small volume.  out may be NULL to query the size. */
int csky_read_baked_texture(csky_ctx* ctx, int which, void* out, size_t capacity, size_t* bytes);
/* Test hook: the march's range-restricted exact square root (cloud_core.h::sqrt_shell, |p|^2 of sample positions) over an array, so that
 * a test can check it EXHAUSTIVELY against IEEE sqrtf on the range it is used on (all 30 067 floats in [3.597e13, 3.6097e13]). */
int csky_test_sqrt_shell(csky_ctx* ctx, const float* in, float* out, size_t n);
/* Measurement hook of tools/isa_profile.py: one launch of the cloud kernel over `bands` with the statistics buffer bound, then the first n
 * (<= 256) 32-bit basic-block execution counters behind the kernel's own tallies.  The counters are written only by the CENSUS build of the
 * library (the product assembly with a counter per basic block, made by that tool); the product build leaves them zero. */
int csky_census_clouds(csky_ctx* ctx, const csky_cloud_params* p, int tile_w, const csky_bands* bands, uint32_t* counts, int n);

/* The other direction, on the GPU (bc7enc.hip, one block per lane): n_images images of w x h RGBA8 texels back to back -> per image
 * ceil(h/4) x ceil(w/4) blocks of 16 bytes, row-major (a 3-D texture is its slices: one image per slice, every mip level its own call, as the
 * importer stores them).  All eight modes (6; 0-3 for opaque blocks; 4, 5 and 7 for blocks whose alpha varies); principal-axis fit + least-squares refits, smallest
 * squared error wins.  It is NOT the engine's encoder (that one cannot be reproduced): textures passed through this and csky_decode_bc7 show the
 * SIZE of what compress/mode=2 does to a frame (tools/bc7_sensitivity.py), not the reference's exact texels. */
int csky_encode_bc7(csky_ctx* ctx, const uint8_t* rgba8, int w, int h, int n_images, uint8_t* blocks_out);
/* quality 0 = csky_encode_bc7; 1 = eight instead of four partitions fitted in full per multi-subset mode and, per subset, coordinate descent on the
 * stored end points (every channel of either end -2 .. +2 steps, indices searched again, until a sweep improves nothing): the second encoder of the
 * sensitivity study's error bar (profiles/r05/bc7_sensitivity.txt).  Frozen there: a sensitivity tool, not a product feature. */
int csky_encode_bc7_quality(csky_ctx* ctx, const uint8_t* rgba8, int w, int h, int n_images, int quality, uint8_t* blocks_out);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDSKY_INTERNAL_H */
