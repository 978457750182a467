/*
 * cloudsky.h -- C ABI of libcloudsky.so: the MI355X (gfx950) HIP implementation of the compute path of
 * clayjohn/godot-volumetric-cloud-demo-v2 (cloud_sky/clouds.glsl, sky-lut.glsl, transmittance-lut.glsl).
 *
 * The reference has no FFI: the path sits behind Godot's RenderingDevice compute API driven from GDScript
 * (cloud_sky/cloud_sky.gd, sky_lut.gd, transmittance_lut.gd).  Each entry point below names the
 * reference call site (file:line, relative to the reference root) it replaces; INTEGRATION.md shows the
 * GDExtension / GDScript-side binding a maintainer would add.
 *
 * This header is the PRODUCT surface: exactly the rows of INTEGRATION.md's table.  The measurement, tuning and test entry points the benchmarks and
 * the test-suite use (kernel variants, schedules, ray segments, per-kernel timing, the instruction census, texture read-back, the BC7 encoder of
 * the sensitivity study) are exported by the same library and declared in cloudsky_internal.h; a host binds this file alone.
 *
 * Conventions: plain C types only; 0 = success, < 0 = error (csky_last_error() gives the text); nothing
 * throws or aborts across the ABI.  One context = one GPU = one caller thread (the reference marshals
 * every RenderingDevice call onto the single render thread: cloud_sky.gd:118,154).  There is NO CPU
 * fallback: without a usable HIP device csky_create fails with CSKY_ERR_NO_DEVICE.
 *
 * Process environment: when libcloudsky.so is LOADED it sets GPU_MAX_HW_QUEUES=16 for the process unless the variable is already set (the HIP
 * runtime reads it at its first call; with the default of 4 the streams of two frames in flight share hardware queues and do not overlap; a rank share with eight frames
 * in flight needs more than 8: 0.30 ms per frame with 8 queues, 0.23 with 16).
 * CSKY_NO_ENV=1 in the environment disables that; csky_set_frames_in_flight(>= 2) then leaves a warning in csky_last_warning when the
 * variable is not in effect.  Other variables read (all optional, A/B switches): CSKY_PERSISTENT, CSKY_PERSISTENT_WGS, CSKY_MULTI_STAGED.
 *
 * Images are tightly packed little-endian RGBA half floats (DATA_FORMAT_R16G16B16A16_SFLOAT,
 * cloud_sky.gd:369; sky_lut.gd:84; transmittance_lut.gd:37), row-major, row 0 = pixel y == 0.
 */
#ifndef CLOUDSKY_H
#define CLOUDSKY_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CSKY_OK 0
#define CSKY_ERR_INVALID (-1)    /* bad argument / call order                               */
#define CSKY_ERR_NO_DEVICE (-2)  /* no HIP device, or the device is not usable               */
#define CSKY_ERR_HIP (-3)        /* a HIP runtime call failed                                */
#define CSKY_ERR_IO (-4)         /* asset file problem                                       */
#define CSKY_ERR_STATE (-5)      /* e.g. clouds requested before noise / LUTs exist          */

#define CSKY_ABI_VERSION 7  /* 7: csky_multi_last_warning (csky_multi_create no longer fails without peer access: it falls back to staged copies and says so); 6: the measurement / tuning / test entry points moved to cloudsky_internal.h (same library), csky_generate_shape_noise_tuned[_device]; 5: csky_last_warning (warnings no longer sit in csky_last_error), exact fp32-coefficient texture cells, csky_render_sky_lut_rows_device, csky_interleave_bands_device, csky_encode_bc7, rings eight deep; 4: csky_submit_* / csky_collect, csky_multi_set_groups / _set_staged, csky_composite_view, csky_external_frame_* (incl. _fence / _ready / _wait); 3: csky_set_noise_mips, csky_decode_bc7, csky_load_ctex[3d]; 2: csky_multi_*, device asset builders */

typedef struct csky_ctx csky_ctx; /* opaque: owns every device allocation, the HIP stream and events */

/* Push-constant block of clouds.glsl:18-40, byte-identical (112 B), as packed by
 * cloud_sky.gd:251-289 `_fill_push_constant()`; pass that PackedFloat32Array through unchanged. */
typedef struct {
    float texture_size[2];    /*   0 */
    float update_position[2]; /*   8  pixel offset of the dispatched tile (clouds.glsl:260) */
    float cloud_pos[2];       /*  16 */
    float detailed_pos[2];    /*  24 */
    float weather_pos[2];     /*  32 */
    float pad1[2];            /*  40 */
    float ground_color[4];    /*  48 */
    float LIGHT_DIRECTION[3]; /*  64 */
    float LIGHT_ENERGY;       /*  76 */
    float LIGHT_COLOR[3];     /*  80 */
    float time;               /*  92 */
    float pad2;               /*  96 */
    float density;            /* 100 */
    float cloud_coverage;     /* 104 */
    float time_offset;        /* 108 */
} csky_cloud_params;

/* Push-constant block of sky-lut.glsl:12-18 (32 B), as packed by sky_lut.gd:123-132. */
typedef struct { float texture_size[2]; float pad2[2]; float sun_direction[3]; float pad; } csky_sky_params;
/* Push-constant block of transmittance-lut.glsl:12-15 (16 B), as packed by transmittance_lut.gd:66-70. */
typedef struct { float texture_size[2]; float pad2[2]; } csky_transmittance_params;

/* Which pixel rows of the frame one call renders: bands of `band_rows` rows; this call renders bands
 * first_band, first_band + band_stride, ... (n_bands of them) into a COMPACT output of n_bands*band_rows
 * rows.  {rows=H, 0, 1, 1} is the whole frame.  It is the multi-GPU generalisation of the reference's
 * update_position tile walk (cloud_sky.gd:156-161): rank r of N takes first_band=r, band_stride=N. */
typedef struct { int band_rows, first_band, band_stride, n_bands; } csky_bands;

typedef struct {
    uint64_t rays;            /* pixels rendered by the last csky_render_clouds* call               */
    uint64_t primary_samples; /* primary march samples evaluated (rays above the horizon x steps)    */
    uint64_t incloud_samples; /* primary samples with density > 0 (clouds.glsl:184 branch taken)     */
} csky_cloud_stats;

/* ---- lifetime ------------------------------------------------------------------------------------
 * replaces cloud_sky.gd:355-408 `_initialize_compute_code` (pipeline + texture creation) and
 * cloud_sky.gd:197-212 `cleanup` / NOTIFICATION_PREDELETE.  device_id = HIP ordinal (one ctx per GPU;
 * multi-GPU = one process per GPU).  csky_destroy(NULL) is a no-op. */
int csky_abi_version(void);
int csky_device_count(void);
int csky_create(csky_ctx** out, int device_id);
void csky_destroy(csky_ctx* ctx);
const char* csky_last_error(const csky_ctx* ctx); /* ctx may be NULL: last create/asset error of this thread */
/* Text left by the last csky_set_noise* / csky_set_frames_in_flight call that returned CSKY_OK with a caveat ("" if none).  Never mixed into
 * csky_last_error: a success does not leave text where the cause of the NEXT failure is looked for. */
const char* csky_last_warning(const csky_ctx* ctx);

/* ---- inputs --------------------------------------------------------------------------------------
 * replaces cloud_sky.gd:298-341 `_create_noise_uniform_set` (REPEAT + LINEAR sampler, three textures).
 * large_rgba8: 128^3 RGBA8, index ((z*128 + y)*128 + x)*4, z = slice index  (perlworlnoise.tga)
 * small_rgb8 : 32^3  RGB8,  index ((z*32  + y)*32  + x)*3                    (worlnoise.bmp)
 * weather_rgb8: 512x512 RGB8, row 0 = top row of the bitmap                  (weather.bmp)
 * Level-0 data only; the library builds the mip chains (2x2x2 box) and its device layouts. */
int csky_set_noise(csky_ctx* ctx, const uint8_t* large_rgba8, const uint8_t* small_rgb8, const uint8_t* weather_rgb8);
/* The same with caller-supplied mip chains: large_chain_rgba8 / small_chain_rgb8 hold ALL levels back to back (level l of an n^3 volume
 * at byte csky_mip_offset(n, l, channels), 8 levels for 128^3, 6 for 32^3).  For hosts that have the importer's own chains, e.g. read
 * from the project's .godot/imported/ files with csky_load_ctex3d (perlworlnoise.tga.import:24, worlnoise.bmp.import:24:
 * mipmaps/generate=true): the sampler of the reference sees exactly those texels, not a re-derived box filter. */
int csky_set_noise_mips(csky_ctx* ctx, const uint8_t* large_chain_rgba8, const uint8_t* small_chain_rgb8, const uint8_t* weather_rgb8);
/* The device layouts store finite differences of neighbouring texels as fp16 (exact for integers up to 2048).  Returns how
 * many coefficients of the textures bound by the last csky_set_noise* do NOT fit fp16 exactly (0 for natural noise; white noise and
 * checkerboards of extreme values exceed the range).  Such textures are marched on EXACT cells instead -- the same filter polynomial with
 * fp32 coefficients, twice the bytes per tap, always the whole-ray compact kernel (the tuning knobs of cloudsky_internal.h do not apply) --
 * so results are exact for ANY 8-bit input; csky_last_warning(ctx) says so, the count stays queryable.  The shipped textures give 0. */
int csky_noise_inexact_coeffs(csky_ctx* ctx, uint64_t* count);
/* 1 = build and march the exact fp32-coefficient cells regardless of the count above (A/B and tests: for textures that fit fp16 the two cell
 * forms filter bit-identically); 0 = only when needed (default).  Takes effect at the next csky_set_noise*. */
int csky_set_exact_cells(csky_ctx* ctx, int mode);
/* clouds.glsl:228 (128 primary steps) and clouds.glsl:186 (6 light steps) are literals in the reference;
 * this generalises them (BASELINE config 2 is 64 x 4).  light_steps in [0,6], primary_steps in [1,1024]. */
int csky_set_march(csky_ctx* ctx, int primary_steps, int light_steps);
/* Wave-level early-out threshold: a wavefront stops marching once every lane's transmittance T < eps.
 * The reference has no early-out; eps = 0 disables it (bit-for-bit reference behaviour).  Default 0. */
int csky_set_early_out(csky_ctx* ctx, float eps);

/* ---- the three kernels, host-buffer forms (block until the image is in host memory) ---------------
 * csky_render_transmittance: transmittance_lut.gd:66-77 (dispatch 32x8 groups of transmittance-lut.glsl).
 * csky_render_sky_lut      : sky_lut.gd:122-148 `render_lut` (dispatch 25x13 groups of sky-lut.glsl);
 *                            samples the context's transmittance LUT (rendering it first if needed).
 * csky_render_clouds       : cloud_sky.gd:234-248 `_render_process` (dispatch of clouds.glsl) for the
 *                            rectangle [0,tile_w) x [0,tile_h) of gl_GlobalInvocationID, offset by
 *                            params->update_position exactly like clouds.glsl:260; binds the sky LUT
 *                            rendered last (cloud_sky.gd:242).  tile = texture_size renders the whole
 *                            hemisphere in one call.
 * out may be NULL (render only; read back later with csky_read_*). */
int csky_render_transmittance(csky_ctx* ctx, const csky_transmittance_params* p, uint16_t* out_rgba16f);
int csky_render_sky_lut(csky_ctx* ctx, const csky_sky_params* p, uint16_t* out_rgba16f);
int csky_render_clouds(csky_ctx* ctx, const csky_cloud_params* p, int tile_w, int tile_h, uint16_t* out_rgba16f,
                       size_t row_pitch_bytes);

/* ---- device-buffer forms (no host copy, asynchronous on `hip_stream`) -----------------------------
 * d_out is a DEVICE pointer the caller owns (e.g. a torch tensor's data_ptr); hip_stream is a
 * hipStream_t passed as void*.  NULL = the context's own stream, which is created NON-BLOCKING: work on it is NOT ordered
 * against the HIP null stream (and a null-stream handle cannot be told apart from NULL).  A caller whose own work runs on the
 * null stream (torch's default stream) must therefore pass a real stream and order it with events on both sides, e.g.
 * side.wait_stream(default) / default.wait_stream(side): the Python host class does exactly that (cloud_sky.py::_march_stream).
 * The LUT inputs of later calls are
 * always the context's internal copies, so the chain transmittance -> sky -> clouds needs no host hop.
 * The sky LUT and the per-frame constants derived from it are rendered on an internal "prologue" stream into rings (the
 * LUT two deep: all its readers run on that stream; the per-frame constants eight deep; like the reference's texture rings, sky_lut.gd:143-146): when frames are enqueued back to back, the prologue of
 * frame k+1 overlaps the march of frame k.  The library orders prologue -> march -> reuse of a ring slot with events, so a
 * caller only has to order its own reads of d_out behind `hip_stream`; csky_render_sky_lut_device ignores `hip_stream`. */
int csky_render_sky_lut_device(csky_ctx* ctx, const csky_sky_params* p, void* hip_stream);
/* Copy the sky LUT rendered last (RGBA16F, w*h*8 bytes) into a caller-owned DEVICE buffer, asynchronously; the copy is ordered
 * behind the LUT's render and `hip_stream` is made to wait for it.  What sky_lut.gd:143-146 does by rotating texture_rd[3]: a host
 * that wants the reference's three-deep ring of LUT copies (for clouds.gdshader's sky_blend_from/to) keeps them with this. */
int csky_copy_sky_lut_device(csky_ctx* ctx, void* d_out_rgba16f, void* hip_stream);
/* One rank's part of the sky LUT when N processes split a frame (sky_lut.gd:43-52 renders the LUT once per frame, cloud_sky.gd:187:
 * on N ranks that would be N identical copies of a kernel that costs 33 us of a whole chip, 12 % of a 1/8 frame share).  Renders rows
 * first_row, first_row + row_stride, ... (0 <= first_row < row_stride) of the p->texture_size LUT, COMPACT (ceil((h - first_row) / row_stride)
 * rows of w RGBA16F texels), straight into the caller's DEVICE buffer on `hip_stream` (the context's own stream if NULL): the buffer that
 * travels to the gathering rank behind the rank's bands, where the rows are interleaved (tiling.py / bench.py).  The context keeps no LUT:
 * the (at most 12) texels its frame set-up filters (clouds.glsl:163-167) are rendered by the set-up of each following csky_render_clouds*
 * call with the same per-texel code -- WHERE it taps comes from that call's LIGHT_DIRECTION, but the sun the texels are rendered for is the one
 * recorded by THIS call (p->sun_direction), exactly as with a whole LUT (cloud_sky.gd:242 binds the LUT rendered last): after a sun change,
 * render the rows again before the next frame, or its sun colours are the old sun's.  Frames are byte-identical to those marched with a whole LUT.
 * csky_read_sky_lut / csky_copy_sky_lut_device return CSKY_ERR_STATE until the next csky_render_sky_lut*. */
int csky_render_sky_lut_rows_device(csky_ctx* ctx, const csky_sky_params* p, int first_row, int row_stride, void* d_rows_out_rgba16f,
                                    size_t capacity_bytes, void* hip_stream);
/* The gathering rank's last step when N processes split a frame (one gather to rank 0): the gather leaves every member's compact
 * bands back to back (member m at d_gathered + m * member_stride_bytes); frame band k (band_bytes each: band_rows x row bytes; total_bands of
 * them) = member k % members, local band k / members.  Asynchronous on `hip_stream`; a deliberately narrow, HBM-bound copy that runs beside the
 * marches of the following frames.  Also used for the sky-LUT rows (band = one row).  Sizes and pointers: multiples of 16 bytes. */
int csky_interleave_bands_device(csky_ctx* ctx, const void* d_gathered, size_t member_stride_bytes, int members, size_t band_bytes, int total_bands,
                                 void* d_frame, void* hip_stream);
int csky_render_clouds_device(csky_ctx* ctx, const csky_cloud_params* p, int tile_w, const csky_bands* bands,
                              void* d_out_rgba16f, size_t row_pitch_bytes, void* hip_stream);
int csky_sync(csky_ctx* ctx); /* wait for the context's own streams (work on caller streams is the caller's to wait for) */

/* ---- asynchronous host form: submit / collect (the throughput path of a host that needs the frame in HOST memory) ---------------
 * csky_render_clouds blocks for march + copy, one frame at a time.  These keep `slots` frames in flight instead: csky_submit_clouds enqueues
 * the march of tile [0,tile_w) x [0,tile_h) (binding the sky LUT rendered last, like csky_render_clouds) and the copy of the frame into a
 * PINNED host buffer of an internal ring, on that slot's own stream, and returns a ticket at once; csky_collect(ticket) blocks until that
 * frame is in host memory and returns a pointer to it (tightly packed RGBA16F, tile_w*tile_h*8 bytes), valid until `slots` further
 * submits.  Tickets count up from 0; at most `slots` may be outstanding (CSKY_ERR_STATE otherwise) and they may be collected in any order.
 * csky_poll: 1 = ready, 0 = still in flight.  With 2 or more slots the march of frame k+1 overlaps the copy and the launch tail of
 * frame k (the library sets its two-frames-in-flight launch policy, csky_set_frames_in_flight).  This is what the GDExtension's
 * submit_clouds() / collect() wrap (gdext/cloudsky_gdextension.c): the reference's frame loop (cloud_sky.gd:129-187 on frame_pre_draw,
 * rd.texture_update into the ring textures of :368-378) has a frame of slack by construction -- the cloud texture it draws with is the
 * one finished in an EARLIER update pass (:137-148). */
int csky_set_host_ring(csky_ctx* ctx, int slots);   /* 1..8, default 2 */
int csky_submit_clouds(csky_ctx* ctx, const csky_cloud_params* p, int tile_w, int tile_h, int64_t* ticket);
int csky_collect(csky_ctx* ctx, int64_t ticket, const uint16_t** frame_rgba16f, size_t* bytes);
int csky_poll(csky_ctx* ctx, int64_t ticket);

/* ---- zero-copy interop: "returns the same TextureRD" without a host hop ------------------------------------------------------------
 * The march can store its pixels straight into memory the ENGINE's texture is bound to: the engine side (Vulkan) allocates the image's
 * memory with VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT, exports the fd (vkGetMemoryFdKHR) and hands it over; these entry points import it
 * (hipImportExternalMemory / hipExternalMemoryGetMappedBuffer) and return a device pointer usable as d_out of csky_render_clouds_device
 * (row pitch = the image's VkSubresourceLayout.rowPitch for a LINEAR-tiled R16G16B16A16_SFLOAT image).  An exported VkSemaphore, imported with
 * ..._import_semaphore_fd and signalled on the march's stream by csky_external_frame_signal, orders the engine's sampling behind the march.
 * gdext/unverified/zero_copy_vulkan.c holds the Vulkan half and the Godot glue (RenderingDevice.texture_create_from_extension -> Texture2DRD); neither
 * Vulkan headers nor an engine exist in this image, so that file is compile-guarded; this half is exercised against a foreign allocator
 * (a hipMemCreate allocation exported as a dma-buf fd: tools/ext_frame_roundtrip.py, tests/test_gpu_round3.py).
 * The library takes ownership of the fds on success; on ANY failure the caller still owns them (the runtime is handed a duplicate).
 * Ordering without a semaphore: ROCm 7.2 on Linux refuses hipImportExternalSemaphore for every handle type (hipErrorNotSupported,
 * profiles/r03/external_semaphore_probe.txt; ..._import_semaphore_fd then returns CSKY_ERR_HIP and the frame stays usable).  ..._fence records an event
 * behind the march on its stream; the host polls ..._ready (1 = the frame is complete, 0 = still marching) or blocks in ..._wait before it
 * lets the engine sample the image -- the reference draws with textures finished in EARLIER passes (cloud_sky.gd:137-148), so the poll
 * at the start of the next pass normally finds the frame done. */
typedef struct csky_external_frame csky_external_frame;
int csky_external_frame_import_fd(csky_ctx* ctx, int opaque_fd, size_t allocation_bytes, size_t offset, size_t frame_bytes,
                                  csky_external_frame** out, void** d_ptr);
int csky_external_frame_import_semaphore_fd(csky_ctx* ctx, csky_external_frame* f, int opaque_fd);
int csky_external_frame_signal(csky_ctx* ctx, csky_external_frame* f, void* hip_stream);
int csky_external_frame_fence(csky_ctx* ctx, csky_external_frame* f, void* hip_stream);
int csky_external_frame_ready(csky_ctx* ctx, csky_external_frame* f);   /* 1 / 0, negative = error */
int csky_external_frame_wait(csky_ctx* ctx, csky_external_frame* f);
void csky_external_frame_release(csky_external_frame* f);

/* Read back the context's internal LUT copies (tests, the compositor, Texture2DRD.texture_update). */
int csky_read_transmittance(csky_ctx* ctx, uint16_t* out_rgba16f, int* w, int* h);
int csky_read_sky_lut(csky_ctx* ctx, uint16_t* out_rgba16f, int* w, int* h);

/* ---- sky compositor ("next" row: the consumer of the path) -----------------------------------------
 * clouds.gdshader:105-116 `sky()` with its helpers (:15-103): samples the two cloud textures through the inverse
 * hemi-octahedral map, cross-fades them by blend_amount, adds the atmosphere (two sky LUTs / 50) and the sun disk +
 * bloom attenuated by the transmittance LUT, fades to the atmosphere at the horizon.  Godot evaluates it per screen
 * pixel (EYEDIR); this entry point evaluates it for an equirectangular panorama (u -> azimuth, v -> elevation, y up).
 * cloud_from/to = blend_from_texture/blend_to_texture, sky_from/to = sky_blend_from/to_texture (clouds_material.tres),
 * all RGBA16F host buffers; source_transmittance is the context's LUT.  light_direction = LIGHT0_DIRECTION. */
typedef struct {
    int out_w, out_h;
    int cloud_w, cloud_h, sky_w, sky_h;
    float blend_amount;       /* clouds.gdshader:12, cloud_sky.gd:152 */
    float sun_disk_scale;     /* clouds.gdshader:13, clouds_sky.tres: 2.0 */
    float light_direction[3];
} csky_composite_params;
int csky_composite_sky(csky_ctx* ctx, const csky_composite_params* p, const uint16_t* cloud_from, const uint16_t* cloud_to,
                       const uint16_t* sky_from, const uint16_t* sky_to, uint16_t* out_rgba16f);
/* The same shader evaluated the way the engine evaluates it: one EYEDIR per SCREEN pixel of a perspective camera (clouds.gdshader:105-116 runs
 * with the viewport's EYEDIR).  basis = Camera3D.global_transform.basis, column-major (basis.x = right, basis.y = up, basis.z = back: the camera
 * looks down -z), fov_y_degrees = Camera3D.fov (vertical); the aspect ratio is out_w / out_h.  Pixel (i, j) -> NDC ((i+.5)/w*2-1, 1-(j+.5)/h*2)
 * -> view ray (x tan(fov/2) aspect, y tan(fov/2), -1) -> world -> normalised. */
typedef struct { float basis[9]; float fov_y_degrees; } csky_view;
int csky_composite_view(csky_ctx* ctx, const csky_composite_params* p, const csky_view* view, const uint16_t* cloud_from, const uint16_t* cloud_to,
                        const uint16_t* sky_from, const uint16_t* sky_to, uint16_t* out_rgba16f);

/* ---- frames in flight ---------------------------------------------------------------------------- */
/* Policy hint for the automatic segment / schedule choice: n = 2..8: the caller keeps n frames in flight by rotating n streams
 * between consecutive csky_render_*_device calls (always safe: per-frame state lives in eight-deep rings ordered by events); the
 * next frames then fill the tail of this one and fewer, longer wavefronts are the better choice for partial frames.  Default 1;
 * 2 is the best choice for whole and half frames; more only pays for one GPU's share of a split frame (1/4 share 0.49 -> 0.45 -> 0.42 ms,
 * 1/8 share 0.31 -> 0.25 -> 0.22 ms per frame with 2 -> 4 -> 8; profiles/r04/frames_in_flight_depth.txt).
 * The streams must map to different hardware queues: the library sets GPU_MAX_HW_QUEUES=16 at load time unless the host already set it
 * (the HIP runtime's default of 4 loses part of the overlap); that works when the library is loaded before the process's first HIP call.
 * With 2, whole-ray launches of 12 Ki - 64 Ki wavefronts (a 2048x1024 frame, half of it) run in the persistent form: one workgroup per
 * resident slot, wavefronts pop tiles from per-XCD sequences of the schedule and steal from the other XCDs at the end (kernels.hip,
 * clouds_kernel_persistent); frames are byte-identical either way.  Environment variable CSKY_PERSISTENT, read by csky_create, is the A/B
 * switch: 0 = never, 1 = this policy (default), 2 = every whole-ray launch. */
int csky_set_frames_in_flight(csky_ctx* ctx, int frames);

/* ---- multi-GPU: the devices of one node behind one handle ------------------------------
 * One host thread drives n devices.  Rays are independent and the reference already renders disjoint tiles addressed by
 * update_position with frozen parameters (cloud_sky.gd:54-55,142,156-161): device i of n renders the 8-row bands i, i+n, ...
 * (interleaved for balance) with the same push-constant block, inputs replicated, and its wavefronts store their pixels
 * STRAIGHT into the frame on the first device through xGMI peer access (where a device has none: csky_multi_last_warning) (64 contiguous bytes per tile row): there is no staging
 * buffer, no gather step and no host hop.  Every device renders its own copy of the two LUTs (36 K texels: cheaper than a
 * broadcast).  Events order the consumer stream on the first device behind all marches.  A device id may appear more than once
 * (two contexts sharing one GPU): meaningless for speed, it lets a single-GPU box exercise the n > 1 path.
 * csky_multi_ctx(m, i) gives the per-device context for the per-context settings (csky_set_march, csky_set_early_out, ...);
 * csky_multi_set_* apply one setting to all of them. */
typedef struct csky_multi csky_multi;
int csky_multi_create(csky_multi** out, const int* device_ids, int n_devices);
void csky_multi_destroy(csky_multi* m);
int csky_multi_device_count(const csky_multi* m);
csky_ctx* csky_multi_ctx(csky_multi* m, int i);
const char* csky_multi_last_error(const csky_multi* m);
/* "" or what csky_multi_create had to fall back on: a device without peer access to the first one (hipDeviceCanAccessPeer says no, or enabling it
 * failed) switches the WHOLE handle to the staged form (csky_multi_set_staged(1), which then cannot be switched off) and makes the first device
 * render the whole sky LUT.  Same frames, slower; the text names the devices. */
const char* csky_multi_last_warning(const csky_multi* m);
int csky_multi_set_noise(csky_multi* m, const uint8_t* large_rgba8, const uint8_t* small_rgb8, const uint8_t* weather_rgb8);
int csky_multi_set_noise_mips(csky_multi* m, const uint8_t* large_chain_rgba8, const uint8_t* small_chain_rgb8, const uint8_t* weather_rgb8);
/* n = 2..8: the caller keeps n frames in flight (per frame group) by rotating n consumer streams between consecutive
 * csky_multi_render_clouds_device calls: every device then rotates n streams / event sets as well (csky_set_frames_in_flight on every
 * context; the per-device rings are eight deep).  Default 1.  frames x groups <= 8. */
int csky_multi_set_frames_in_flight(csky_multi* m, int frames);
/* Frame groups, for THROUGHPUT workloads (a sequence of independent frames: BASELINE config 5's 64-frame sun sweep): the n devices are
 * split into `groups` groups of n/groups devices; consecutive csky_multi_render_clouds_device calls go to the groups in turn and the
 * devices of a group split that frame's bands (n/groups)-way.  A device's share is then `groups` times larger (fewer, fuller launches:
 * one GPU's 1/8 share of a 2048x1024 frame is 4 wavefronts per SIMD and latency-bound), at the price of `groups` frames of latency:
 * the caller keeps groups x frames_in_flight frames in flight on as many rotating streams and output buffers.  Every frame still lands
 * in d_out on the FIRST device.  With groups > 1, csky_multi_render_sky_lut renders the LUT on the devices of the NEXT frame's group
 * only: call it once per frame, before that frame's render call (the order of sky_lut.gd:43-52 / cloud_sky.gd:187).  Default 1 = every
 * device works on every frame (the latency-optimal split, BASELINE config 4).  groups must divide the device count. */
int csky_multi_set_groups(csky_multi* m, int groups);
/* 1: devices other than the first render into a local band buffer and one strided peer copy per device moves the bands into the frame
 * behind the march, instead of the march's wavefronts storing straight into the first device's memory over xGMI.  Same frame, one
 * extra pass over 1/n of it; for nodes where fine-grained remote stores stall.  Also switched on by CSKY_MULTI_STAGED=1 in the
 * environment at csky_multi_create.  Default 0. */
int csky_multi_set_staged(csky_multi* m, int staged);
int csky_multi_set_march(csky_multi* m, int primary_steps, int light_steps);
/* sky_lut.gd:122-148 for the handle: device i renders rows i, i + n, ... of the LUT and stores them, like its bands, straight into the LUT of the
 * first device (csky_read_sky_lut / csky_copy_sky_lut_device on csky_multi_ctx(m, 0) give the whole LUT and wait for every writer); no device's
 * frame set-up reads that copy, each renders the few texels it filters itself (see csky_render_sky_lut_rows_device), on every device whatever the
 * group layout.  The other contexts of the handle hold no LUT (CSKY_ERR_STATE from their csky_read_sky_lut). */
int csky_multi_render_sky_lut(csky_multi* m, const csky_sky_params* p);
/* Whole tile [0,tile_w) x [0,tile_h) into d_out on the FIRST device (row pitch in bytes), asynchronously: `hip_stream` (a stream of
 * the first device; NULL = that context's own stream) is ordered behind every device's march.  tile_h must be a multiple of 8 (bands are
 * 8 rows; both forms reject other heights up front, unlike the single-device csky_render_clouds, which takes ragged tiles). */
int csky_multi_render_clouds_device(csky_multi* m, const csky_cloud_params* p, int tile_w, int tile_h, void* d_out_rgba16f,
                                    size_t row_pitch_bytes, void* hip_stream);
/* Host-buffer form: renders as above into an internal frame on the first device, copies it out, blocks. */
int csky_multi_render_clouds(csky_multi* m, const csky_cloud_params* p, int tile_w, int tile_h, uint16_t* out_rgba16f, size_t row_pitch_bytes);
int csky_multi_sync(csky_multi* m);
/* Asynchronous host form over the handle (see csky_submit_clouds): frames land in pinned host memory of the first device's context.
 * slots = groups x frames in flight per group (csky_multi_set_host_ring sets the latter). */
int csky_multi_set_host_ring(csky_multi* m, int slots);
int csky_multi_submit_clouds(csky_multi* m, const csky_cloud_params* p, int tile_w, int tile_h, int64_t* ticket);
int csky_multi_collect(csky_multi* m, int64_t ticket, const uint16_t** frame_rgba16f, size_t* bytes);

/* ---- asset layer (host only; usable without a GPU) ------------------------------------------------
 * What the reference gets from Godot's importers (weather.bmp.import, worlnoise.bmp.import,
 * perlworlnoise.tga.import).  perlworlnoise.tga is missing from the reference checkout, hence the
 * deterministic generator. */
int csky_load_bmp_rgb8(const char* path, int* w, int* h, uint8_t* out_rgb8, size_t out_capacity);
/* Truevision TGA, true colour 24/32 bpp, uncompressed or RLE (the container of cloud_sky/perlworlnoise.tga) -> RGBA8. */
int csky_load_tga_rgba8(const char* path, int* w, int* h, uint8_t* out_rgba8, size_t out_capacity);
int csky_strip_to_volume(const uint8_t* strip, int n, int ch, uint8_t* vol);
int csky_generate_shape_noise(uint32_t seed, int n, uint8_t* out_rgba8);
/* The generator's knobs (README.md:30 TODO 3: "a noise generator so custom noise can be created and tweaked").  csky_generate_shape_noise uses
 * csky_shape_noise_default_params, the calibration every benchmark and parity input is made with; other settings are for looking at what the
 * missing asset's character does to the picture (tools/demo_scene.py).  R = clamp((remap(clamp(perlin_fbm * perlin_gain + 0.5), 0, 1,
 * G * dilate, 1) - centre) * contrast + offset); G / B / A = inverted Worley fBm at worley_freq x 1 / 2 / 4. */
typedef struct csky_shape_noise_params {
    int32_t perlin_freq, perlin_octaves, worley_freq;
    float perlin_gain, dilate, centre, contrast, offset;
} csky_shape_noise_params;
void csky_shape_noise_default_params(csky_shape_noise_params* p);
int csky_check_shape_noise_params(const csky_shape_noise_params* p, int n);
int csky_generate_shape_noise_tuned(uint32_t seed, int n, const csky_shape_noise_params* params, uint8_t* out_rgba8);
int csky_generate_shape_noise_tuned_device(csky_ctx* ctx, uint32_t seed, int n, const csky_shape_noise_params* params, uint8_t* out_rgba8);
/* The same generator as a HIP kernel (one voxel per lane): byte-identical output, ~1 ms for 128^3 (README.md:30 TODO 3). */
int csky_generate_shape_noise_device(csky_ctx* ctx, uint32_t seed, int n, uint8_t* out_rgba8);
/* A generated 32^3 RGB detail volume in the role of worlnoise.bmp (three tileable inverted-Worley fBm channels calibrated on the asset's
 * statistics, noise_core.h), host and GPU (byte-identical): README.md:30 TODO 3 "generate the noise on the GPU". */
int csky_generate_detail_noise(uint32_t seed, int n, uint8_t* out_rgb8);
int csky_generate_detail_noise_device(csky_ctx* ctx, uint32_t seed, int n, uint8_t* out_rgb8);
size_t csky_mip_offset(int n, int level, int ch);
/* 3-D mip chain (mipmaps/generate=true of the .import files): 2x2x2 box, (sum + 4) >> 3.  `vol` holds level 0 on entry and has room for
 * csky_mip_offset(n, levels, ch) bytes.  csky_build_mips runs on the host, csky_build_mips_device on the GPU (byte-identical); since round 2
 * csky_set_noise builds its chains and its device layouts on the GPU itself (kernels.hip::launch_mip_chain / launch_bake). */
int csky_build_mips(uint8_t* vol, int n, int ch, int levels);
int csky_build_mips_device(csky_ctx* ctx, uint8_t* vol, int n, int ch, int levels);
/* ---- what Godot's importer wrote (godot_import.cpp; host only) ------------------------------------
 * The reference's noise textures are imported with compress/mode=2, compress/high_quality=true (weather.bmp.import:19-20,
 * worlnoise.bmp.import:19-20, perlworlnoise.tga.import:19-20), i.e. as BPTC (BC7) blocks in .godot/imported/<name>-<md5>.bptc.ctex
 * / .ctex3d: what the reference's samplers return are the DECODED blocks.  Decoding is fixed by the format (all 8 block modes,
 * checked against an independent decoder in tests/test_godot_import.py); the engine's encoder is not reproduced, so a host that
 * wants the reference's exact texels loads the imported files:
 *   csky_decode_bc7   w x h pixels from ceil(w/4) x ceil(h/4) 16-byte blocks (row-major) -> RGBA8
 *   csky_load_ctex    CompressedTexture2D ("GST2"): *levels images (level 0 first, then the stored mips) back to back as RGBA8
 *   csky_load_ctex3d  CompressedTexture3D ("GSTL"): the d slices of level 0, then the slices of every stored mip level, as RGBA8
 * Raw (uncompressed R8/RGB8/RGBA8) and BPTC_RGBA payloads are read; PNG/WebP/Basis payloads are refused.  out may be NULL to query the
 * sizes.  The container layout follows the engine's loader (Godot 4.2 scene/resources/compressed_texture.cpp); no imported file
 * ships with the reference, so only the BC7 decoder is pinned by an outside implementation. */
int csky_decode_bc7(const uint8_t* blocks, int w, int h, uint8_t* out_rgba8);
int csky_load_ctex(const char* path, int* w, int* h, int* levels, uint8_t* out_rgba8, size_t out_capacity);
int csky_load_ctex3d(const char* path, int* w, int* h, int* d, int* levels, uint8_t* out_rgba8, size_t out_capacity);
const char* csky_assets_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDSKY_H */
