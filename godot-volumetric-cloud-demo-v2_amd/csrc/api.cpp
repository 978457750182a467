// api.cpp -- the C ABI of libcloudsky (include/cloudsky.h; the measurement / tuning / test entry points: include/cloudsky_internal.h): context, device memory, texture baking, launches.
// Mirrors the resource ownership of the reference's GDScript drivers: cloud_sky.gd (`_initialize_compute_code`,
// `_render_process`, `cleanup`), sky_lut.gd (`render_lut`), transmittance_lut.gd (`_initialize_compute_code`).
// There is no CPU render path here: every render entry point needs a live HIP device.
#include <hip/hip_runtime_api.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>
#include <unistd.h>
#include <sys/syscall.h>
#include <cerrno>
#include "../../include/cloudsky_internal.h"
#include "kernels.h"
#include "bake.h"
#include "bake_core.h"
#include "cloud_core.h"

using namespace csky;

// Two frames in flight (csky_set_frames_in_flight) need the caller's two streams on DIFFERENT hardware queues.  The HIP runtime
// multiplexes a process's streams onto GPU_MAX_HW_QUEUES queues (default 4) in first-use order, and with the library's two internal streams
// plus a host's own, four are not enough (measured: no overlap at 4, overlap at 8).  The runtime reads the variable when it initialises, at
// the first HIP call; this runs when libcloudsky.so is loaded, so a host that cannot set environment variables (a GDExtension inside Godot)
// still gets the overlap as long as it has not used HIP before loading the library.  An existing value is never overwritten.
// Opt-out: CSKY_NO_ENV=1 in the environment leaves the process's environment alone (a host that loads other HIP users and wants the runtime's
// defaults); csky_set_frames_in_flight then warns through csky_last_warning when the variable is not in effect.
__attribute__((constructor)) static void csky_runtime_defaults() {
    const char* no = getenv("CSKY_NO_ENV");
    if (no && no[0] && no[0] != '0') return;
    if (!getenv("GPU_MAX_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0);
}

static_assert(sizeof(csky_cloud_params) == sizeof(CloudParams), "ABI struct mismatch");

// Depth of the per-frame rings (frame constants, launch order, cost feedback, pop counters, events): the number of frames a caller may keep
// in flight on as many streams (csky_set_frames_in_flight).  The slots rotate over all RING entries whatever that number is.
constexpr int RING = 8;
constexpr int HOST_RING = 8;   // pinned host frames of the asynchronous host form (a single context uses up to RING of them, csky_multi up to groups x frames in flight)

struct csky_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_copy = nullptr;
    // noise set (cloud_sky.gd:298-341)
    uint8_t* d_raw_large = nullptr; uint8_t* d_raw_small = nullptr; uint8_t* d_raw_weather = nullptr; uint8_t* d_bake_meta = nullptr;   // 8-bit mip chains (inputs of the device bake)
    ShapeTexel* d_shape = nullptr; unsigned long long inexact_coeffs = 0; uint4* d_detail = nullptr; uint4* d_weather = nullptr; uint16_t* d_detail_h = nullptr; bool have_noise = false;
    float* d_brick = nullptr;                                               // CSKY_BRICK_BOUND experiment build only
    // exact cells (bake_core.h): fp32-coefficient layouts, built when a coefficient of the bound textures does not fit fp16 (or exact_cells == 1)
    float4* d_shape32 = nullptr; float4* d_detail32 = nullptr; float4* d_weather32 = nullptr; bool cell32 = false; int exact_cells = 0;
    uint32_t shape_off[SHAPE_LEVELS] = {}, detail_off[DETAIL_LEVELS] = {};
    float detail_lod5 = 0.0f;
    double w_rmin = 0.0, w_rmax = 1.0, w_bmax = 1.0;   // range of the weather map's cloud-type / coverage channels
    float win_cov = -1e30f, win_lo = -1.0f, win_hi = 2.0f; bool use_window = true;
    // LUTs: RGBA16F image + float4 copy of the rounded values
    uint16_t* d_trans_h = nullptr; float4* d_trans_f = nullptr; int tw = 0, th = 0; bool have_trans = false;
    uint16_t* d_sky_h = nullptr; float4* d_sky_f = nullptr; int sw = 0, sh = 0; bool have_sky = false;   // = ring slot sky_cur
    FrameConsts* d_fc = nullptr;                                                                          // = ring slot fc_cur
    // Frame prologue pipeline.  The sky LUT and the frame set-up of frame k+1 are small dependent kernels; enqueued behind the
    // cloud kernel of frame k they cost their run time plus two launch gaps per frame (6 % of one GPU's 1/8-frame share).  They
    // run on the context's own prologue stream instead, into the next slot of a ring (the sky LUT two deep: every reader of it runs on `pro`;
    // the frame constants RING deep: the marches of up to RING frames in flight read them) (the reference keeps three-deep
    // texture rings for the same reason, sky_lut.gd:143-146), so they overlap the march of the previous frame; events order
    // set-up -> clouds (ev_setup) and clouds -> the next writer of that slot (ev_clouds).  All sky-LUT readers run on `pro`.
    hipStream_t pro = nullptr;
    uint16_t* sky_h_ring[2] = {nullptr, nullptr}; float4* sky_f_ring[2] = {nullptr, nullptr}; int sky_cur = 0;
    // csky_render_sky_lut_rows_device: the LUT of sun sky_sun exists only as the rows the caller's buffer received (one rank of an N-way frame
    // split); the texels this context's frame set-up filters are rendered by the set-up kernel itself (clouds_dev)
    bool sky_partial = false; float sky_sun[3] = {0, 1, 0}; int psw = 0, psh = 0;
    // csky_multi_render_sky_lut: the whole LUT IS in this context's memory (ring slot sky_cur), written row by row by the devices of the handle;
    // readers of the memory copy wait for those writers first.  (sky_partial stays set: the frame set-ups never read the memory copy.)
    bool sky_in_memory = false; std::vector<hipEvent_t> lut_writers;
    FrameConsts* fc_ring[RING] = {}; int fc_cur = 0;
    hipEvent_t ev_setup[RING] = {}, ev_clouds[RING] = {}; bool clouds_pending[RING] = {};
    unsigned long long* d_stats = nullptr;
    uint2* d_frame = nullptr; size_t frame_px = 0;  // internal frame for the host-buffer form / timing
    int primary_steps = 128, light_steps = 6;        // clouds.glsl:228, :186
    float early_eps = 0.0f;
    int variant = CSKY_DEFAULT_VARIANT;
    int sched_mode = -1;                              // -1 = auto (5 for large launches, 2 for small ones)
    int segments = 0;                                 // ray segments per ray: 0 = auto, 1, 2, 4
    int frames_in_flight = 1;                         // policy hint (csky_set_frames_in_flight): the caller alternates that many streams
    // static workgroup order (physical workgroup -> slab), written on the device, one table per ring slot (= frame parity, so two
    // frames in flight with different geometries never share one), cached per launch geometry
    uint32_t* d_order_ring[RING] = {}; size_t order_cap[RING] = {}; int order_grid_ring[RING] = {};
    long long order_key_ring[RING][4];     // csky_create fills them with -1
    // cost-feedback schedule (mode 7): per-workgroup costs of the last launch -> heaviest-first order of the next one
    uint32_t* d_wg_cost = nullptr; uint32_t* d_lpt_order = nullptr; uint32_t* d_lpt_hist = nullptr; size_t lpt_cap = 0;
    uint32_t* d_heads = nullptr; int persistent = 1; int resident_wgs = 0;   // persistent launches: 2 ring slots x (8 per-XCD pop counters + exit counter)
    bool lpt_valid[RING] = {}; long long lpt_key[RING][11];   // csky_create fills the keys with -1
    // optional per-launch timing of the cloud kernel (csky_set_kernel_timing): HIP event pairs recorded around the launch on ITS stream
    bool kt_on = false; std::vector<hipEvent_t> kt_ev; int kt_count = 0;   // the event pool grows on demand (clouds_dev)
    uint8_t* d_composite = nullptr; size_t composite_cap = 0;              // grow-only scratch of csky_composite_sky
    csky_cloud_stats last_stats = {0, 0, 0};
    // asynchronous host form (csky_submit_clouds / csky_collect): a ring of pinned host frames + device frames on rotating internal streams
    struct HostSlot { hipStream_t s = nullptr; hipEvent_t done = nullptr; uint2* d = nullptr; void* h = nullptr; size_t px = 0; long long ticket = -1; int w = 0, hh = 0; bool busy = false; };
    HostSlot hring[HOST_RING]; int hslots = 2; long long next_ticket = 0;
    char err[512] = {0};
    char warn[512] = {0};                              // csky_last_warning: text of the last call that succeeded with a caveat (never mixed into err)
};

namespace {
constexpr size_t CSKY_STATS_WORDS = 2 + 128;             // [0..1] the kernel's own tallies; then 256 32-bit basic-block counters of the census build (tools/isa_profile.py; zero in the product build)
thread_local char g_err[512];

int fail(csky_ctx* c, int code, const char* fmt, ...) {
    char* dst = c ? c->err : g_err;
    va_list ap; va_start(ap, fmt); vsnprintf(dst, 512, fmt, ap); va_end(ap);
    if (code == CSKY_ERR_HIP) (void)hipGetLastError();       // the runtime's last-error slot is sticky: a failed call must not resurface as the "launch error" of a later kernel
    return code;
}
#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), CSKY_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

int bind(csky_ctx* c) { HIPCHK(c, hipSetDevice(c->device)); return CSKY_OK; }

template <class T> int dev_alloc(csky_ctx* c, T** p, size_t count) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return CSKY_OK;
}

int ensure_trans(csky_ctx* c, int w, int h) {
    if (c->d_trans_h && c->tw == w && c->th == h) return CSKY_OK;
    int rc; if ((rc = dev_alloc(c, &c->d_trans_h, (size_t)w * h * 4))) return rc;
    if ((rc = dev_alloc(c, &c->d_trans_f, (size_t)w * h))) return rc;
    c->tw = w; c->th = h; c->have_trans = false; return CSKY_OK;
}
int ensure_sky(csky_ctx* c, int w, int h) {
    if (c->d_sky_h && c->sw == w && c->sh == h) return CSKY_OK;
    if (c->pro) HIPCHK(c, hipStreamSynchronize(c->pro));     // a size change is rare: drain the prologue stream, rebuild both slots
    for (int k = 0; k < 2; k++) {
        int rc; if ((rc = dev_alloc(c, &c->sky_h_ring[k], (size_t)w * h * 4))) return rc;
        if ((rc = dev_alloc(c, &c->sky_f_ring[k], (size_t)w * h))) return rc;
    }
    c->sky_cur = 0; c->d_sky_h = c->sky_h_ring[0]; c->d_sky_f = c->sky_f_ring[0];
    c->sw = w; c->sh = h; c->have_sky = false; return CSKY_OK;
}
int ensure_frame(csky_ctx* c, size_t px) {
    if (c->d_frame && c->frame_px >= px) return CSKY_OK;
    int rc; if ((rc = dev_alloc(c, &c->d_frame, px))) return rc;
    c->frame_px = px; return CSKY_OK;
}

int render_trans_dev(csky_ctx* c, int w, int h, hipStream_t s) {
    int rc; if ((rc = ensure_trans(c, w, h))) return rc;
    HIPCHK(c, launch_transmittance(w, h, c->d_trans_h, c->d_trans_f, s));
    c->have_trans = true; return CSKY_OK;
}

TexSet texset(const csky_ctx* c) {
    TexSet t;
#ifdef CSKY_BRICK_BOUND
    t.brick = c->d_brick;
#endif
    t.shape = c->d_shape; t.detail = c->d_detail; t.weather = c->d_weather; t.sky = c->d_sky_f; t.sky_w = c->sw; t.sky_h = c->sh; t.detail_lod5 = c->detail_lod5; t.detail_h = c->d_detail_h; t.detail_lds = nullptr;
    return t;
}

TexSet32 texset32(const csky_ctx* c) {
    TexSet32 t;
    static_cast<TexSet&>(t) = texset(c);
    t.shape32 = c->d_shape32; t.detail32 = c->d_detail32; t.weather32 = c->d_weather32;
    return t;
}

int check_bands(csky_ctx* c, const csky_bands* b, int tile_w) {
    if (tile_w < 1 || !b || b->band_rows < 1 || b->n_bands < 0 || b->first_band < 0 || b->band_stride < 1)
        return fail(c, CSKY_ERR_INVALID, "render_clouds: bad tile/bands description");
    return CSKY_OK;
}

// Static workgroup order for ring slot `slot` (kernels.hip::static_order_kernel; modes 1, 2, 5).  Measured on the headline frame
// (queue kernel, round 1): 5 (slab rows round-robin over the XCDs) 3.93 ms, 1 (contiguous eighths) 4.80 ms, 2 (natural) 4.92 ms;
// azimuth-wedge and horizon-first orders (5.3-5.8 / 4.77 ms) were dropped in round 2.  The table depends on the launch geometry
// only (not on update_position: the reference's tile walk re-uses it) and is written by a kernel on the launch's stream.
int ensure_order(csky_ctx* c, int slot, int mode, int tile_w, int tiles_x, int slabs, hipStream_t s) {
    const int nblocks = tiles_x * slabs;
    int grid;
    if (mode == 2) grid = nblocks;
    else if (mode == 1) grid = ((nblocks + 7) >> 3) * 8;
    else grid = ((slabs + 7) >> 3) * tiles_x * 8;
    const long long key[4] = {tile_w, slabs, mode, grid};
    if (c->d_order_ring[slot] && memcmp(key, c->order_key_ring[slot], sizeof key) == 0) return CSKY_OK;
    if (c->order_cap[slot] < (size_t)grid) {
        // growing is rare and costs a device-wide wait (an older launch may still read the old table): grow EVERY slot's table now, so that it
        // happens once, at the first frame of a geometry, and not again at the first use of each of the other ring slots (with rings eight deep
        // and a five-frame warm-up that was three synchronisations inside a timed region)
        HIPCHK(c, hipDeviceSynchronize());
        for (int k = 0; k < RING; k++) {
            if (c->order_cap[k] >= (size_t)grid) continue;
            if (c->d_order_ring[k]) { (void)hipFree(c->d_order_ring[k]); c->d_order_ring[k] = nullptr; c->order_cap[k] = 0; }
            HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_order_ring[k]), (size_t)grid * sizeof(uint32_t)));
            c->order_cap[k] = (size_t)grid;
            for (long long& v : c->order_key_ring[k]) v = -1;
        }
    }
    // the last reader of this slot's table is the march of two frames ago; the caller has already ordered `s` behind it (ev_clouds -> pro ->
    // ev_setup -> s), exactly like the frame constants of the slot
    HIPCHK(c, launch_static_order(mode, tiles_x, slabs, grid, c->d_order_ring[slot], s));
    c->order_grid_ring[slot] = grid;
    memcpy(c->order_key_ring[slot], key, sizeof key);
    return CSKY_OK;
}

// frame_setup + clouds on stream s into d_out (compact rows).  stats: optional device counters.

// The event pool of csky_set_kernel_timing grows to `want` events.  All or nothing: a hipEventCreate failing part-way used to leave null
// entries in the pool for later timed launches to record on; now the new events made so far are destroyed and the pool keeps its old size.
static int grow_timing_pool(csky_ctx* c, size_t want) {
    const size_t old_n = c->kt_ev.size();
    if (want <= old_n) return CSKY_OK;
    c->kt_ev.resize(want, nullptr);
    for (size_t i = old_n; i < want; i++) {
        const hipError_t e = hipEventCreate(&c->kt_ev[i]);
        if (e != hipSuccess) {
            for (size_t j = old_n; j < i; j++) (void)hipEventDestroy(c->kt_ev[j]);
            c->kt_ev.resize(old_n);
            return fail(c, CSKY_ERR_HIP, "kernel timing: hipEventCreate: %s", hipGetErrorString(e));
        }
    }
    return CSKY_OK;
}
int clouds_dev(csky_ctx* c, const csky_cloud_params* p, int tile_w, const csky_bands* b, uint2* d_out, size_t pitch_bytes, hipStream_t s,
               unsigned long long* d_stats, bool setup, bool out_full = false) {
    if (!p) return fail(c, CSKY_ERR_INVALID, "render_clouds: params is NULL");
    if (!c->have_noise) return fail(c, CSKY_ERR_STATE, "render_clouds: csky_set_noise has not been called");
    if (!c->have_sky) return fail(c, CSKY_ERR_STATE, "render_clouds: no sky LUT yet (call csky_render_sky_lut first; cloud_sky.gd:187,242)");
    if (!(p->texture_size[0] >= 1.0f) || !(p->texture_size[1] >= 1.0f)) return fail(c, CSKY_ERR_INVALID, "render_clouds: texture_size must be >= 1");
    int rc; if ((rc = check_bands(c, b, tile_w))) return rc;
    if (pitch_bytes % 8 || pitch_bytes < (size_t)tile_w * 8) return fail(c, CSKY_ERR_INVALID, "render_clouds: row pitch must be a multiple of 8 and >= tile_w*8");
    if (b->n_bands == 0) return CSKY_OK;
    CloudParams cp; memcpy(&cp, p, sizeof cp);
    if (setup) {
        if (c->win_cov != cp.cloud_coverage) {          // height window of the exact reject (bake.h), cached per coverage value
            height_window((double)cp.cloud_coverage, c->w_rmin, c->w_rmax, c->w_bmax, c->win_lo, c->win_hi);
            c->win_cov = cp.cloud_coverage;
        }
        const float lo = c->use_window ? c->win_lo : -1.0f, hi = c->use_window ? c->win_hi : 2.0f;
        // frame set-up on the prologue stream into the other constants slot (its last reader, the march two frames ago, must be done)
        const int f = (c->fc_cur + 1) % RING;
        if (c->clouds_pending[f]) HIPCHK(c, hipStreamWaitEvent(c->pro, c->ev_clouds[f], 0));
        // cloud-type range of the weather map (texel values 0..255): all >= 128 or all <= 127 fixes the branch of the height gradient
        const int ctm = !c->use_window ? 0 : (c->w_rmin * 255.0 >= 127.5 ? 1 : (c->w_rmax * 255.0 <= 127.5 ? 2 : 0));
        if (c->sky_partial)                              // no LUT in memory: the set-up renders the texels of its three taps (clouds.glsl:163-167) itself
            HIPCHK(c, launch_frame_setup_taps(cp, c->sky_sun, c->d_trans_f, c->tw, c->th, c->psw, c->psh, c->primary_steps, c->light_steps, c->early_eps, lo, hi, ctm, c->fc_ring[f], c->pro));
        else
            HIPCHK(c, launch_frame_setup(cp, c->d_sky_f, c->sw, c->sh, c->primary_steps, c->light_steps, c->early_eps, lo, hi, ctm, c->fc_ring[f], c->pro));
        HIPCHK(c, hipEventRecord(c->ev_setup[f], c->pro));
        c->fc_cur = f; c->d_fc = c->fc_ring[f];
        // (Round 5 bounded what folding the set-up INTO the march launch could return by simply not waiting here -- legal in a timing run with constant
        // parameters: whole frame one at a time 2.014 -> 2.000 ms, a 1/8 share 0.408 -> 0.404, eight in flight 0.224 -> 0.227: the prologue of frame k + 1
        // already runs under the march of frame k; profiles/r05/rows_overlap_ab.txt.)
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_setup[f], 0));
    }
    RenderGeom g; g.tile_w = tile_w; g.band_rows = b->band_rows; g.first_band = b->first_band; g.band_stride = b->band_stride; g.n_bands = b->n_bands;
    g.pitch_px = (uint32_t)(pitch_bytes / 8); g.out_full = out_full ? 1 : 0;
    // Launch-size policy: ray segments (more, shorter wavefronts) when the launch is too small to fill the chip with whole-ray
    // wavefronts, and the cost-feedback order (mode 7) when it is only a few resident workgroups deep.  Measured with
    // profiles/r01/launch_size_crossover_compact.txt (the script was retired with round 5's tools purge), "compact" variant, kernel ms at 256 / 1024 / 4096 / 8192 / 16384 / 32768 tiles of 8x8 rays
    // (= 1/128 .. 1/1 of the headline frame; profiles/r01/launch_size_crossover_compact.txt):
    //   whole rays, slab rows per XCD   (seg 1, sched 5)    0.50  0.56  0.58  0.94  1.39  2.15   <- full frames
    //   whole rays, cost feedback       (seg 1, sched 7)    0.50  0.51  0.66  0.80  1.10  2.18   <- 1/2 frame (one of 2 GPUs)
    //   2 step-range segments, feedback (seg 2, sched 7)    0.33  0.45  0.53  0.65  1.20  2.37   <- 1/4 frame
    //   4 step-range segments, feedback (seg 4, sched 7)    0.22  0.30  0.42  0.75  1.43  2.81   <- 1/8 frame
    //   4 step-range segments, natural  (seg 4, sched 2)    0.22  0.28  0.50  0.81  1.48  2.83
    //   4 interleaved segments, natural (seg 5, sched 2)    0.16  0.30  0.80  1.52  2.95  5.75   <- latency: the reference's 96x96 tiles
    // A lone wavefront is bound by its chain of dependent gathers, so small launches want more, shorter wavefronts; large
    // launches want the fewest instructions.  (The "queue" variant keeps its own, earlier crossovers: 6144 / 1536 wavefronts.)
    const long long waves = ((long long)(tile_w + 7) / 8) * (((long long)b->n_bands * b->band_rows + 7) / 8);
    const int variant = c->cell32 ? 3 : c->variant;              // exact cells exist for the compact whole-ray kernel only
    const bool queued = variant == 1 || variant == 3;
    int seg = c->cell32 ? 1 : (queued ? c->segments : 1);
    int auto_mode;
    // (the policy below is written for the kernel that RUNS: `variant`, not c->variant -- exact cells always march with the compact whole-ray kernel)
    if (variant == 3 && c->frames_in_flight >= 2) {
        // the caller keeps two frames in flight on two streams (csky_set_frames_in_flight): the next frame's workgroups fill this
        // launch's tail, so fewer, longer wavefronts win (tools/share_matrix.py, ms per frame at 1/2, 1/4, 1/8, 1/16 of the frame):
        //   seg 1: 0.96 (s5) 0.52 (s7) 0.42 0.35    seg 2: 1.18 0.61 0.34 (s7) 0.29    seg 4: 1.27 0.75 0.39 0.22 (s7)
        // With three or four frames in flight (the rings are four deep) a 1/8 share is best marched as whole rays:
        //   1/8 frame, cost feedback, x2 / x3 / x4:  seg 1 0.330 0.284 0.279   seg 2 0.313 0.352 0.355   (1/4 frame and larger: no gain over x2)
        // Round 4, rings eight deep (16 hardware queues): a 1/8 share x4 / x6 / x8: 0.247 / 0.228 / 0.224 ms (whole rays, static order; cost
        // feedback 0.242 / 0.226 / 0.222 once every ring slot has its previous launch's costs: not in a short run), a 1/4 share x2 / x4 / x8:
        // 0.487 / 0.446 / 0.415 static, 0.444 / 0.436 / 0.428 feedback; 1/2 and whole frames gain nothing beyond x2 (profiles/r04/frames_in_flight_depth.txt).
        const int whole_from = c->frames_in_flight >= 3 ? 3072 : 6144;
        if (queued && seg == 0) seg = waves >= whole_from ? 1 : (waves >= 3072 ? 2 : (waves >= 768 ? 4 : 5));
        auto_mode = (waves >= 12288 || (c->frames_in_flight >= 6 && waves >= 3072)) ? 5 : (waves >= 768 ? 7 : 2);
    } else if (variant == 3) {
        if (queued && seg == 0) seg = waves >= 12288 ? 1 : (waves >= 6144 ? 2 : (waves >= 768 ? 4 : 5));
        auto_mode = waves >= 24576 ? 5 : (waves >= 1536 ? 7 : 2);
    } else {
        if (queued && seg == 0) seg = waves >= 6144 ? 1 : (waves >= 1536 ? 4 : 5);
        auto_mode = waves >= 6144 ? 5 : 2;
    }
    if (variant == 2) seg = 16;
    int mode = c->sched_mode >= 0 ? c->sched_mode : auto_mode;
    const int bw = seg == 5 ? 8 : (seg == 16 ? 128 : 32 / seg);   // workgroup footprint = bw x 8 pixels (seg 5: one tile; seg 16: the 16-wavefront "lds" strip)
    const int tiles_x = (g.tile_w + bw - 1) / bw, slabs = (g.n_bands * g.band_rows + 7) >> 3, nblocks = tiles_x * slabs;
    bool feedback = mode == 7 && queued && seg != 5;             // kernels that record per-workgroup costs
    if (mode == 7 && !feedback) mode = waves >= 12288 ? 5 : 2;
    const int static_mode = mode == 7 ? (waves >= 12288 ? 5 : 2) : mode;   // order of the first launch of a geometry under feedback
    const int slot = c->fc_cur;
    // Persistent launch form (kernels.hip::clouds_kernel_persistent): as many workgroups as the chip holds, their wavefronts pop
    // footprints from per-XCD sequences of the launch order and, at the end, from the other XCDs' sequences.  Measured
    // (profiles/r02/persistent_launch_ab.txt), ms per frame plain -> persistent: whole frame with two frames
    // in flight 1.806 -> 1.724 (bench.py, alternating runs), 1/2 frame 0.938 -> 0.882; one frame at a time 2.12 -> 2.17 (plain
    // launches refill freed slots at least as well when nothing else is in flight), 1/4 frame 0.477 -> 0.539 (barely deeper than
    // the resident grid), 4096x2048 6.13 -> 6.19 (no tail to fill), cost-feedback order 1.90 -> 2.09.  So: whole-ray launches of
    // 12 Ki to 64 Ki wavefronts (they run in the static XCD-row order) while the caller keeps two frames in flight.
    const bool persist = !c->cell32 && seg == 1 && variant == 3 && (c->persistent == 2 || (c->persistent == 1 && !feedback && c->frames_in_flight >= 2 && waves >= 12288 && waves <= 65536));
    uint32_t* const heads = persist ? c->d_heads + slot * 16 : nullptr;
    const int resident = c->resident_wgs;
    TexSet32 t32; const TexSet32* t32p = nullptr;
    if (c->cell32) { t32 = texset32(c); t32p = &t32; }
    hipEvent_t* kt = nullptr;                                    // timing pair of this launch (csky_set_kernel_timing)
    if (c->kt_on) {
        if ((size_t)c->kt_count * 2 + 2 > c->kt_ev.size()) {    // the pool grows on demand: no launch is ever dropped from the sum
            if ((rc = grow_timing_pool(c, c->kt_ev.empty() ? 512 : c->kt_ev.size() * 2))) return rc;
        }
        kt = &c->kt_ev[(size_t)c->kt_count * 2]; c->kt_count++;
    }
    if ((rc = ensure_order(c, slot, static_mode, g.tile_w, tiles_x, slabs, s))) return rc;
    uint32_t* const d_static = c->d_order_ring[slot];
    const int static_grid = c->order_grid_ring[slot];
    if (!feedback) {
        if (kt) HIPCHK(c, hipEventRecord(kt[0], s));
        {   // a failed persistent launch may leave the slot's pop counters armed: re-zero them so that the next launch on this slot starts clean
            const hipError_t le = launch_clouds(variant, seg, texset(c), c->d_fc, g, d_static, static_grid, d_out, d_stats, nullptr, s, heads, resident, t32p);
            if (le != hipSuccess) { if (heads) (void)hipMemsetAsync(heads, 0, 16 * sizeof(uint32_t), s); return fail(c, CSKY_ERR_HIP, "cloud kernel launch failed: %s", hipGetErrorString(le)); }
        }
        if (kt) HIPCHK(c, hipEventRecord(kt[1], s));
        HIPCHK(c, hipEventRecord(c->ev_clouds[c->fc_cur], s)); c->clouds_pending[c->fc_cur] = true;
        return CSKY_OK;
    }
    // mode 7: this launch runs in the order derived from the costs of the previous launch ON THE SAME RING SLOT (same geometry and
    // view), records its own costs and derives the next order from them.  The first launch of a geometry uses the static order.  Costs,
    // order and sort scratch are per ring slot (= per frame parity, like the frame constants), so two frames in flight on two
    // streams never share them; reuse of a slot is ordered by ev_clouds, recorded below after the sort.
    const size_t need = (size_t)(static_grid > nblocks ? static_grid : nblocks);
    if (c->lpt_cap < need) {
        HIPCHK(c, hipDeviceSynchronize());                   // (re)allocation is rare; frames may be in flight on other streams
        (void)hipFree(c->d_wg_cost); (void)hipFree(c->d_lpt_order); c->d_wg_cost = c->d_lpt_order = nullptr; c->lpt_cap = 0;
        for (int k = 0; k < RING; k++) c->lpt_valid[k] = false;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_wg_cost), RING * need * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_lpt_order), RING * need * sizeof(uint32_t)));
        if (!c->d_lpt_hist) {
            HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_lpt_hist), RING * 2048 * sizeof(uint32_t)));
            HIPCHK(c, hipMemset(c->d_lpt_hist, 0, RING * 2048 * sizeof(uint32_t)));
        }
        HIPCHK(c, hipMemset(c->d_wg_cost, 0, RING * need * sizeof(uint32_t)));       // the sort kernels leave both zeroed afterwards
        c->lpt_cap = need;
    }
    uint32_t* const cost = c->d_wg_cost + (size_t)slot * c->lpt_cap;
    uint32_t* const lorder = c->d_lpt_order + (size_t)slot * c->lpt_cap;
    // the costs belong to one view of one tile: same launch geometry AND same place in the texture (a tile walk never reuses them)
    const long long fkey[11] = {g.tile_w, g.band_rows, g.first_band, g.band_stride, g.n_bands, (long long)cp.texture_size[0], (long long)cp.texture_size[1],
                                (long long)cp.update_position[0], (long long)cp.update_position[1], mode * 16 + static_mode, seg};
    if (memcmp(c->lpt_key[slot], fkey, sizeof fkey) != 0) { c->lpt_valid[slot] = false; memcpy(c->lpt_key[slot], fkey, sizeof fkey); }
    const uint32_t* const use_order = c->lpt_valid[slot] ? lorder : d_static;
    const int use_grid = c->lpt_valid[slot] ? nblocks : static_grid;
    if (kt) HIPCHK(c, hipEventRecord(kt[0], s));
    // (Round 6, measured and removed, profiles/r06/split_tail_ab.txt: this launch as TWO -- the heaviest 60-90 % of the feedback order here, the lightest 10-40 % on a
    // second stream at equal or lowest priority to back-fill the tail: one frame at a time 2.00 -> 2.00-2.05 ms, frames identical.  The tail is the last wavefronts'
    // serial chains, not idle slots; the next frame fills it.)
    {
        const hipError_t le = launch_clouds(variant, seg, texset(c), c->d_fc, g, use_order, use_grid, d_out, d_stats, cost, s, heads, resident, t32p);
        if (le != hipSuccess) { if (heads) (void)hipMemsetAsync(heads, 0, 16 * sizeof(uint32_t), s); return fail(c, CSKY_ERR_HIP, "cloud kernel launch failed: %s", hipGetErrorString(le)); }
    }
    if (kt) HIPCHK(c, hipEventRecord(kt[1], s));
    int shift = 0;
    while ((((long long)256 * (c->primary_steps + 16)) >> shift) >= 1024) shift++;     // largest cost: 4 wavefronts x 64 rays x (steps + 16)
    HIPCHK(c, launch_lpt_order(cost, nblocks, shift, c->d_lpt_hist + slot * 2048, lorder, s));
    c->lpt_valid[slot] = true;
    HIPCHK(c, hipEventRecord(c->ev_clouds[c->fc_cur], s)); c->clouds_pending[c->fc_cur] = true;
    return CSKY_OK;
}

}  // namespace

extern "C" {

int csky_abi_version(void) { return CSKY_ABI_VERSION; }

int csky_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* csky_last_error(const csky_ctx* ctx) { return ctx ? ctx->err : g_err; }
const char* csky_last_warning(const csky_ctx* ctx) { return ctx ? ctx->warn : ""; }

int csky_create(csky_ctx** out, int device_id) {
    if (!out) return fail(nullptr, CSKY_ERR_INVALID, "csky_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, CSKY_ERR_NO_DEVICE, "csky_create: no HIP device available (%s); libcloudsky has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, CSKY_ERR_INVALID, "csky_create: device_id %d out of range [0,%d)", device_id, n);
    csky_ctx* c = new (std::nothrow) csky_ctx();
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_create: out of host memory");
    c->device = device_id;
    auto bail = [&](const char* what, hipError_t err) { fail(nullptr, CSKY_ERR_HIP, "csky_create: %s failed: %s", what, hipGetErrorString(err)); csky_destroy(c); return CSKY_ERR_HIP; };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    // (a HIGH-PRIORITY prologue stream was measured in round 2: whole frames with two frames in flight 1.78 -> 2.02 ms, one rank's 1/8 share
    // 0.329 -> 0.335 ms: the priority queue breaks the overlap of the two frame streams.  Plain stream.)
    if ((e = hipStreamCreateWithFlags(&c->pro, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    for (int k = 0; k < RING; k++) {
        for (long long& v : c->order_key_ring[k]) v = -1;
        for (long long& v : c->lpt_key[k]) v = -1;
        if ((e = hipEventCreateWithFlags(&c->ev_setup[k], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&c->ev_clouds[k], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipMalloc(reinterpret_cast<void**>(&c->fc_ring[k]), sizeof(FrameConsts))) != hipSuccess) return bail("hipMalloc", e);
    }
    c->d_fc = c->fc_ring[0];
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_stats), CSKY_STATS_WORDS * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_heads), RING * 16 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(c->d_heads, 0, RING * 16 * sizeof(uint32_t))) != hipSuccess) return bail("hipMemset", e);   // persistent launches leave them zero
    { int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id); c->resident_wgs = (cus > 0 ? cus : 256) * cloud_resident_workgroups_per_cu(); }
    // A/B switch (CSKY_PERSISTENT; the A/B of profiles/r02/persistent_launch_ab.txt): 0 = never, 1 = the policy of clouds_dev (default), 2 = every whole-ray launch
    if (const char* pe = getenv("CSKY_PERSISTENT")) c->persistent = atoi(pe);
    if (const char* pe = getenv("CSKY_PERSISTENT_WGS")) { const int n = atoi(pe); if (n > 0) c->resident_wgs = n; }
    *out = c;
    return CSKY_OK;
}

void csky_destroy(csky_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();                              // launches may sit on caller streams too
    void* ptrs[] = {c->d_shape, c->d_detail, c->d_weather, c->d_trans_h, c->d_trans_f, c->sky_h_ring[0], c->sky_h_ring[1], c->sky_f_ring[0], c->sky_f_ring[1],
                    c->d_stats, c->d_frame, c->d_composite, c->d_raw_large, c->d_raw_small, c->d_raw_weather, c->d_bake_meta, c->d_detail_h, c->d_wg_cost, c->d_lpt_order, c->d_lpt_hist, c->d_heads, c->d_shape32, c->d_detail32, c->d_weather32};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int k = 0; k < RING; k++) {
        if (c->fc_ring[k]) (void)hipFree(c->fc_ring[k]);
        if (c->d_order_ring[k]) (void)hipFree(c->d_order_ring[k]);
        if (c->ev_setup[k]) (void)hipEventDestroy(c->ev_setup[k]);
        if (c->ev_clouds[k]) (void)hipEventDestroy(c->ev_clouds[k]);
    }
    for (auto& hs : c->hring) {
        if (hs.d) (void)hipFree(hs.d);
        if (hs.h) (void)hipHostFree(hs.h);
        if (hs.done) (void)hipEventDestroy(hs.done);
        if (hs.s) (void)hipStreamDestroy(hs.s);
    }
    hipEvent_t evs[] = {c->ev0, c->ev1, c->ev_copy};
    for (hipEvent_t ev : evs) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->kt_ev) if (ev) (void)hipEventDestroy(ev);
    if (c->pro) (void)hipStreamDestroy(c->pro);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int set_noise_impl(csky_ctx* c, const uint8_t* large_rgba8, const uint8_t* small_rgb8, const uint8_t* weather_rgb8, bool chains_given) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_noise: ctx is NULL");
    if (!large_rgba8 || !small_rgb8 || !weather_rgb8) return fail(c, CSKY_ERR_INVALID, "csky_set_noise: NULL texture pointer");
    int rc; if ((rc = bind(c))) return rc;
    // The level-0 textures go to the device as they are (9.2 MB); mip chains (mipmaps/generate=true, perlworlnoise.tga.import:24,
    // worlnoise.bmp.import:24) and the device layouts are built there (kernels.hip::launch_mip_chain / launch_bake).
    const size_t large_l0 = (size_t)SHAPE_N * SHAPE_N * SHAPE_N * 4, small_l0 = (size_t)DETAIL_N * DETAIL_N * DETAIL_N * 3, weather_b = (size_t)WEATHER_N * WEATHER_N * 3;
    const size_t large_chain = chain_offset(SHAPE_N, SHAPE_LEVELS, 4), small_chain = chain_offset(DETAIL_N, DETAIL_LEVELS, 3);
    size_t shape_total = 0, detail_total = 0;
    for (int l = 0; l < SHAPE_LEVELS; l++) { c->shape_off[l] = (uint32_t)shape_total; const size_t n = SHAPE_N >> l; shape_total += n * n * n; }
    for (int l = 0; l < DETAIL_LEVELS; l++) { c->detail_off[l] = (uint32_t)detail_total; const size_t n = DETAIL_N >> l; detail_total += n * n * n; }
    for (int l = 0; l < SHAPE_LEVELS; l++) if (c->shape_off[l] != shape_level_offset(l)) return fail(c, CSKY_ERR_INVALID, "internal: shape mip offset mismatch");
    for (int l = 0; l < DETAIL_LEVELS; l++) if (c->detail_off[l] != detail_level_offset(l)) return fail(c, CSKY_ERR_INVALID, "internal: detail mip offset mismatch");
    if (detail_total != (size_t)DETAIL_CHAIN_TEXELS) return fail(c, CSKY_ERR_INVALID, "internal: detail chain size");
    HIPCHK(c, hipDeviceSynchronize());                        // frames reading the old textures may be in flight on caller streams
    c->have_noise = false; c->win_cov = -1e30f;               // a failure below leaves freed / half-baked textures: no render until a later call succeeds
    if ((rc = dev_alloc(c, &c->d_raw_large, large_chain))) return rc;
    if ((rc = dev_alloc(c, &c->d_raw_small, small_chain))) return rc;
    if ((rc = dev_alloc(c, &c->d_raw_weather, weather_b))) return rc;
    if ((rc = dev_alloc(c, &c->d_shape, shape_total))) return rc;
    if ((rc = dev_alloc(c, &c->d_detail, detail_total))) return rc;
    if ((rc = dev_alloc(c, &c->d_weather, (size_t)WEATHER_N * WEATHER_N))) return rc;
    if ((rc = dev_alloc(c, &c->d_detail_h, detail_total + 8))) return rc;
    if (!c->d_bake_meta) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_bake_meta), 32));
    struct { unsigned long long inexact; int range[3]; int pad; } meta = {0ull, {255, 0, 0}, 0};
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->d_bake_meta, &meta, sizeof meta, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_raw_large, large_rgba8, chains_given ? large_chain : large_l0, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_raw_small, small_rgb8, chains_given ? small_chain : small_l0, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_raw_weather, weather_rgb8, weather_b, hipMemcpyHostToDevice, s));
    if (!chains_given) {                                      // csky_set_noise_mips: the caller's chains (e.g. the importer's own, csky_load_ctex3d) are used as they are
        HIPCHK(c, launch_mip_chain(c->d_raw_large, SHAPE_N, 4, SHAPE_LEVELS, s));
        HIPCHK(c, launch_mip_chain(c->d_raw_small, DETAIL_N, 3, DETAIL_LEVELS, s));
    }
    HIPCHK(c, launch_bake(c->d_raw_large, c->d_raw_small, c->d_raw_weather, c->d_shape, c->d_detail, c->d_detail_h, c->d_weather,
                          reinterpret_cast<unsigned long long*>(c->d_bake_meta), reinterpret_cast<int*>(c->d_bake_meta + 8), s));
#ifdef CSKY_BRICK_BOUND
    {   // per 8^3 brick (+1 apron on the high side, REPEAT): bmax = (rmax + 1 - fmin) / (2 - fmin), a hair above (the kernel's rcp is approximate)
        std::vector<float> tab(16 * 16 * 16);
        for (int bz = 0; bz < 16; bz++) for (int by = 0; by < 16; by++) for (int bx = 0; bx < 16; bx++) {
            int rm = 0, fm = 1 << 30;
            for (int z = bz * 8; z <= bz * 8 + 8; z++) for (int y = by * 8; y <= by * 8 + 8; y++) for (int x = bx * 8; x <= bx * 8 + 8; x++) {
                const uint8_t* tx = large_rgba8 + ((((size_t)(z & 127) * 128 + (y & 127)) * 128 + (x & 127)) * 4);
                rm = std::max(rm, (int)tx[0]); fm = std::min(fm, 5 * tx[1] + 2 * tx[2] + tx[3]);
            }
            const float r = rm * (1.0f / 255.0f), f = fm * (1.0f / (8.0f * 255.0f));
            tab[((size_t)bz * 16 + by) * 16 + bx] = (r + (1.0f - f)) / (2.0f - f) * 1.000004f;
        }
        if (!c->d_brick) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_brick), tab.size() * sizeof(float)));
        HIPCHK(c, hipMemcpy(c->d_brick, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
#endif
    uint8_t t5[3] = {0, 0, 0};                                // detail LOD 5 is one texel: every tap at that level returns it (cloud_core.h::detail_tap)
    HIPCHK(c, hipMemcpyAsync(&meta, c->d_bake_meta, sizeof meta, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(t5, c->d_raw_small + chain_offset(DETAIL_N, 5, 3), 3, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    c->inexact_coeffs = meta.inexact;
    c->detail_lod5 = (float)(5 * t5[0] + 2 * t5[1] + t5[2]) * (1.0f / (8.0f * 255.0f));
    c->w_rmin = meta.range[0] / 255.0; c->w_rmax = meta.range[1] / 255.0; c->w_bmax = meta.range[2] / 255.0; c->win_cov = -1e30f;   // channel ranges for the height-window reject
    c->warn[0] = 0;
    // Textures whose cells do not fit fp16 (white noise, checkerboards: csky_noise_inexact_coeffs() > 0) are marched on EXACT cells: the same
    // polynomial with fp32 coefficients (bake_core.h), twice the bytes per tap, the compact whole-ray kernel on TexSet32.  (Rounds 1-3 marched
    // them on the rounded fp16 cells and warned)  csky_set_exact_cells(1) asks for them regardless (A/B, tests).
    c->cell32 = c->inexact_coeffs != 0 || c->exact_cells == 1;
    if (c->cell32) {
        if ((rc = dev_alloc(c, &c->d_shape32, shape_total * 4))) return rc;
        if ((rc = dev_alloc(c, &c->d_detail32, detail_total * 2))) return rc;
        if ((rc = dev_alloc(c, &c->d_weather32, (size_t)WEATHER_N * WEATHER_N * 2))) return rc;
        HIPCHK(c, launch_bake32(c->d_raw_large, c->d_raw_small, c->d_raw_weather, c->d_shape32, c->d_detail32, c->d_weather32, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (c->inexact_coeffs)
            snprintf(c->warn, sizeof c->warn, "csky_set_noise: %llu finite-difference coefficients of these textures do not fit fp16: marching on exact fp32 cells "
                     "(twice the bytes per tap; the whole-ray compact kernel)", c->inexact_coeffs);
    } else {
        // textures that fit fp16 again: the ~150 MB of exact cells of an earlier bind are not kept until csky_destroy (nothing is in flight: the device was synchronised above)
        if (c->d_shape32) { (void)hipFree(c->d_shape32); c->d_shape32 = nullptr; }
        if (c->d_detail32) { (void)hipFree(c->d_detail32); c->d_detail32 = nullptr; }
        if (c->d_weather32) { (void)hipFree(c->d_weather32); c->d_weather32 = nullptr; }
    }
    c->have_noise = true;
    return CSKY_OK;
}

int csky_set_noise(csky_ctx* c, const uint8_t* large_rgba8, const uint8_t* small_rgb8, const uint8_t* weather_rgb8) {
    return set_noise_impl(c, large_rgba8, small_rgb8, weather_rgb8, false);
}
int csky_set_noise_mips(csky_ctx* c, const uint8_t* large_chain_rgba8, const uint8_t* small_chain_rgb8, const uint8_t* weather_rgb8) {
    return set_noise_impl(c, large_chain_rgba8, small_chain_rgb8, weather_rgb8, true);
}

int csky_read_baked_texture(csky_ctx* c, int which, void* out, size_t capacity, size_t* bytes) {
    if (!c || !bytes) return fail(c, CSKY_ERR_INVALID, "csky_read_baked_texture: NULL argument");
    if (!c->have_noise) return fail(c, CSKY_ERR_STATE, "csky_read_baked_texture: csky_set_noise has not been called");
    int rc; if ((rc = bind(c))) return rc;
    size_t shape_total = 0;
    for (int l = 0; l < SHAPE_LEVELS; l++) { const size_t n = SHAPE_N >> l; shape_total += n * n * n; }
    const void* src = nullptr; size_t n = 0;
    switch (which) {
        case 0: src = c->d_shape; n = shape_total * sizeof(ShapeTexel); break;
        case 1: src = c->d_detail; n = (size_t)DETAIL_CHAIN_TEXELS * sizeof(uint4); break;
        case 2: src = c->d_weather; n = (size_t)WEATHER_N * WEATHER_N * sizeof(uint4); break;
        case 3: src = c->d_raw_large; n = chain_offset(SHAPE_N, SHAPE_LEVELS, 4); break;
        case 4: src = c->d_raw_small; n = chain_offset(DETAIL_N, DETAIL_LEVELS, 3); break;
        default: return fail(c, CSKY_ERR_INVALID, "csky_read_baked_texture: which must be 0..4");
    }
    *bytes = n;
    if (!out) return CSKY_OK;
    if (capacity < n) return fail(c, CSKY_ERR_INVALID, "csky_read_baked_texture: buffer too small (%zu < %zu)", capacity, n);
    HIPCHK(c, hipMemcpy(out, src, n, hipMemcpyDeviceToHost));
    return CSKY_OK;
}

int csky_test_sqrt_shell(csky_ctx* c, const float* in, float* out, size_t n) {
    if (!c || !in || !out) return fail(c, CSKY_ERR_INVALID, "csky_test_sqrt_shell: NULL argument");
    int rc; if ((rc = bind(c))) return rc;
    float* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), 2 * n * sizeof(float) + 16));
    hipError_t e = hipMemcpyAsync(d, in, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_sqrt_shell(d, d + n, n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + n, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, CSKY_ERR_HIP, "csky_test_sqrt_shell: %s", hipGetErrorString(e));
    return CSKY_OK;
}

int csky_census_clouds(csky_ctx* c, const csky_cloud_params* p, int tile_w, const csky_bands* bands, uint32_t* counts, int n) {
    if (!c || !counts || n < 1 || n > 256) return fail(c, CSKY_ERR_INVALID, "csky_census_clouds: bad arguments (1 .. 256 counters)");
    int rc; if ((rc = bind(c))) return rc;
    if ((rc = check_bands(c, bands, tile_w))) return rc;
    const size_t rows = (size_t)bands->n_bands * bands->band_rows;
    if ((rc = ensure_frame(c, (size_t)tile_w * (rows ? rows : 1)))) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_stats, 0, CSKY_STATS_WORDS * sizeof(unsigned long long), c->stream));
    if ((rc = clouds_dev(c, p, tile_w, bands, c->d_frame, (size_t)tile_w * 8, c->stream, c->d_stats, true))) return rc;
    HIPCHK(c, hipMemcpyAsync(counts, reinterpret_cast<const char*>(c->d_stats) + 16, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CSKY_OK;
}

int csky_build_mips_device(csky_ctx* c, uint8_t* vol, int n, int ch, int levels) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_build_mips_device: ctx is NULL");
    if (!vol || n < 1 || n > 1024 || (n & (n - 1)) || ch < 1 || ch > 4 || levels < 1 || (n >> (levels - 1)) < 1) return fail(c, CSKY_ERR_INVALID, "csky_build_mips_device: bad arguments");
    int rc; if ((rc = bind(c))) return rc;
    const size_t total = chain_offset(n, levels, ch), l0 = (size_t)n * n * n * ch;
    uint8_t* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), total));
    hipError_t e = hipMemcpyAsync(d, vol, l0, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_mip_chain(d, n, ch, levels, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(vol + l0, d + l0, total - l0, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, CSKY_ERR_HIP, "csky_build_mips_device: %s", hipGetErrorString(e));
    return CSKY_OK;
}

int csky_generate_detail_noise_device(csky_ctx* c, uint32_t seed, int n, uint8_t* out_rgb8) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_generate_detail_noise_device: ctx is NULL");
    if (!out_rgb8 || n < 8 || n > 256 || (n & (n - 1))) return fail(c, CSKY_ERR_INVALID, "csky_generate_detail_noise_device: n must be a power of two in [8, 256]");
    int rc; if ((rc = bind(c))) return rc;
    const size_t bytes = (size_t)n * n * n * 3;
    uint8_t* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), bytes));
    hipError_t e = launch_detail_noise(seed, n, d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_rgb8, d, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, CSKY_ERR_HIP, "csky_generate_detail_noise_device: %s", hipGetErrorString(e));
    return CSKY_OK;
}

int csky_encode_bc7(csky_ctx* c, const uint8_t* rgba8, int w, int h, int n_images, uint8_t* blocks_out) { return csky_encode_bc7_quality(c, rgba8, w, h, n_images, 0, blocks_out); }
int csky_encode_bc7_quality(csky_ctx* c, const uint8_t* rgba8, int w, int h, int n_images, int quality, uint8_t* blocks_out) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_encode_bc7: ctx is NULL");
    if (quality < 0 || quality > 1) return fail(c, CSKY_ERR_INVALID, "csky_encode_bc7_quality: quality is 0 or 1");
    if (!rgba8 || !blocks_out || w < 1 || h < 1 || n_images < 1 || w > 16384 || h > 16384 || n_images > 16384) return fail(c, CSKY_ERR_INVALID, "csky_encode_bc7: NULL argument or size out of range");
    int rc; if ((rc = bind(c))) return rc;
    const size_t in_bytes = (size_t)w * h * 4 * (size_t)n_images, out_bytes = (size_t)((w + 3) / 4) * ((h + 3) / 4) * 16 * (size_t)n_images;
    const size_t in_pad = (in_bytes + 15) & ~(size_t)15;        // the blocks follow the texels, 16-byte aligned
    uint8_t* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), in_pad + out_bytes));
    uint8_t* d_out = d + in_pad;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMemcpyAsync(d, rgba8, in_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_bc7_encode(d, w, h, n_images, quality, reinterpret_cast<uint4*>(d_out), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(blocks_out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, CSKY_ERR_HIP, "csky_encode_bc7: %s", hipGetErrorString(e));
    return CSKY_OK;
}

int csky_noise_inexact_coeffs(csky_ctx* c, uint64_t* n) {
    if (!c || !n) return CSKY_ERR_INVALID;
    if (!c->have_noise) return fail(c, CSKY_ERR_STATE, "csky_noise_inexact_coeffs: csky_set_noise has not been called");
    *n = (uint64_t)c->inexact_coeffs;
    return CSKY_OK;
}

int csky_set_march(csky_ctx* c, int primary_steps, int light_steps) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_march: ctx is NULL");
    if (primary_steps < 1 || primary_steps > 1024 || light_steps < 0 || light_steps > 6)
        return fail(c, CSKY_ERR_INVALID, "csky_set_march: primary_steps in [1,1024], light_steps in [0,6] (RANDOM_VECTORS has 6 entries, clouds.glsl:140)");
    c->primary_steps = primary_steps; c->light_steps = light_steps; return CSKY_OK;
}

int csky_set_early_out(csky_ctx* c, float eps) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_early_out: ctx is NULL");
    if (!(eps >= 0.0f) || eps > 0.5f) return fail(c, CSKY_ERR_INVALID, "csky_set_early_out: eps must be in [0, 0.5]");
    c->early_eps = eps; return CSKY_OK;
}

int csky_set_variant(csky_ctx* c, int variant) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_variant: ctx is NULL");
    if (variant < -1 || variant >= cloud_variant_count()) return fail(c, CSKY_ERR_INVALID, "csky_set_variant: unknown variant %d", variant);
    c->variant = variant < 0 ? CSKY_DEFAULT_VARIANT : variant; return CSKY_OK;
}
int csky_set_schedule(csky_ctx* c, int mode) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_schedule: ctx is NULL");
    if (!(mode == -1 || mode == 1 || mode == 2 || mode == 5 || mode == 7))
        return fail(c, CSKY_ERR_INVALID, "csky_set_schedule: mode must be -1 (auto), 1, 2, 5 or 7 (see cloudsky.h)");
    c->sched_mode = mode; return CSKY_OK;
}
int csky_set_frames_in_flight(csky_ctx* c, int frames) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_frames_in_flight: ctx is NULL");
    if (frames < 1 || frames > RING) return fail(c, CSKY_ERR_INVALID, "csky_set_frames_in_flight: 1 .. 8 (the rings are eight deep)");
    c->frames_in_flight = frames;
    c->warn[0] = 0;
    if (frames >= 2) {   // CSKY_OK, but never silent: without enough hardware queues the frames' streams share one and do not overlap (csky_last_warning)
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        const int qn = q ? atoi(q) : 4;                         // the HIP runtime's default
        if (qn < frames + 1)                                   // the frames' streams + the prologue stream
            snprintf(c->warn, sizeof c->warn, "csky_set_frames_in_flight: GPU_MAX_HW_QUEUES is %s: %d streams of consecutive frames plus the prologue stream may share a hardware "
                     "queue and not overlap; set it (16 is what libcloudsky sets at load time unless CSKY_NO_ENV=1) before the process's first HIP call", q ? q : "unset (4)", frames);
    }
    return CSKY_OK;
}
int csky_set_segments(csky_ctx* c, int segments) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_segments: ctx is NULL");
    if (segments != 0 && segments != 1 && segments != 2 && segments != 4 && segments != 5) return fail(c, CSKY_ERR_INVALID, "csky_set_segments: 0 (auto), 1, 2, 4 (step ranges) or 5 (4 interleaved)");
    c->segments = segments; return CSKY_OK;
}
int csky_set_exact_cells(csky_ctx* c, int mode) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_exact_cells: ctx is NULL");
    if (mode != 0 && mode != 1) return fail(c, CSKY_ERR_INVALID, "csky_set_exact_cells: 0 (only when a coefficient does not fit fp16) or 1 (always)");
    c->exact_cells = mode; return CSKY_OK;                     // takes effect at the next csky_set_noise*
}
int csky_set_height_window(csky_ctx* c, int enabled) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_height_window: ctx is NULL");
    c->use_window = enabled != 0; return CSKY_OK;
}
int csky_variant_count(void) { return cloud_variant_count(); }
const char* csky_variant_name(int v) { return cloud_variant_name(v); }

int csky_sync(csky_ctx* c) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_sync: ctx is NULL");
    int rc; if ((rc = bind(c))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->pro));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CSKY_OK;
}

int csky_render_transmittance(csky_ctx* c, const csky_transmittance_params* p, uint16_t* out) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_render_transmittance: ctx is NULL");
    if (!p) return fail(c, CSKY_ERR_INVALID, "csky_render_transmittance: params is NULL");
    const int w = (int)p->texture_size[0], h = (int)p->texture_size[1];
    if (w < 1 || h < 1 || w > 8192 || h > 8192) return fail(c, CSKY_ERR_INVALID, "csky_render_transmittance: texture_size out of range");
    int rc; if ((rc = bind(c))) return rc;
    // sky LUTs in flight read the old transmittance LUT: whole ones and the set-ups' own texels on the prologue stream, a rank's rows
    // (csky_render_sky_lut_rows_device) on CALLER streams; the LUT is rendered once at load (transmittance_lut.gd:15-18), so wait for the device
    HIPCHK(c, hipDeviceSynchronize());
    if ((rc = render_trans_dev(c, w, h, c->stream))) return rc;
    if (out) HIPCHK(c, hipMemcpyAsync(out, c->d_trans_h, (size_t)w * h * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CSKY_OK;
}

int csky_render_sky_lut_device(csky_ctx* c, const csky_sky_params* p, void* hip_stream) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_render_sky_lut: ctx is NULL");
    if (!p) return fail(c, CSKY_ERR_INVALID, "csky_render_sky_lut: params is NULL");
    const int w = (int)p->texture_size[0], h = (int)p->texture_size[1];
    if (w < 1 || h < 1 || w > 8192 || h > 8192) return fail(c, CSKY_ERR_INVALID, "csky_render_sky_lut: texture_size out of range");
    int rc; if ((rc = bind(c))) return rc;
    (void)hip_stream;   // the LUT has no inputs of the caller's: it is rendered on the prologue stream and its consumers are ordered by events
    if (!c->have_trans && (rc = render_trans_dev(c, 256, 64, c->pro))) return rc;   // transmittance_lut.gd:6 default size
    if ((rc = ensure_sky(c, w, h))) return rc;
    const int k = (c->have_sky && c->sky_in_memory) ? c->sky_cur ^ 1 : c->sky_cur;  // the other ring slot: frame set-ups still reading the current one are ahead on `pro`
    HIPCHK(c, launch_sky_lut(w, h, p->sun_direction, c->d_trans_f, c->tw, c->th, c->sky_h_ring[k], c->sky_f_ring[k], c->pro));
    c->sky_cur = k; c->d_sky_h = c->sky_h_ring[k]; c->d_sky_f = c->sky_f_ring[k];
    c->have_sky = true; c->sky_partial = false; c->sky_in_memory = true; c->lut_writers.clear();
    return CSKY_OK;
}
int csky_render_sky_lut_rows_device(csky_ctx* c, const csky_sky_params* p, int first_row, int row_stride, void* d_rows_out, size_t capacity_bytes, void* hip_stream) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_render_sky_lut_rows_device: ctx is NULL");
    if (!p || !d_rows_out) return fail(c, CSKY_ERR_INVALID, "csky_render_sky_lut_rows_device: NULL argument");
    const int w = (int)p->texture_size[0], h = (int)p->texture_size[1];
    if (w < 1 || h < 1 || w > 8192 || h > 8192) return fail(c, CSKY_ERR_INVALID, "csky_render_sky_lut_rows_device: texture_size out of range");
    if (first_row < 0 || row_stride < 1 || first_row >= row_stride) return fail(c, CSKY_ERR_INVALID, "csky_render_sky_lut_rows_device: need 0 <= first_row < row_stride");
    const int n_rows = first_row < h ? (h - first_row + row_stride - 1) / row_stride : 0;
    if (capacity_bytes < (size_t)n_rows * w * 8) return fail(c, CSKY_ERR_INVALID, "csky_render_sky_lut_rows_device: %zu bytes given, %d rows of %d bytes needed", capacity_bytes, n_rows, w * 8);
    int rc; if ((rc = bind(c))) return rc;
    if (!c->have_trans) {                                       // (rendered on the prologue stream: the caller's stream reads it)
        if ((rc = render_trans_dev(c, 256, 64, c->pro))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->pro));
    }
    // the rows have no consumer inside the library: they are rendered on the CALLER's stream, in order with the bands they travel with.
    // (Round 5 measured them on a side stream BESIDE the march that follows, joined behind it: a 1/8 share one frame at a time 0.409 -> 0.460 ms, eight
    // in flight 0.241 -> 0.244: two more cross-stream hops cost more than the rows they take off the critical path; profiles/r05/rows_overlap_ab.txt.)
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    HIPCHK(c, launch_sky_lut_rows(w, h, first_row, row_stride, p->sun_direction, c->d_trans_f, c->tw, c->th, reinterpret_cast<uint2*>(d_rows_out), nullptr, s));
    for (int i = 0; i < 3; i++) c->sky_sun[i] = p->sun_direction[i];
    c->psw = w; c->psh = h; c->sky_partial = true; c->have_sky = true; c->sky_in_memory = false; c->lut_writers.clear();
    return CSKY_OK;
}

int csky_render_sky_lut(csky_ctx* c, const csky_sky_params* p, uint16_t* out) {
    int rc = csky_render_sky_lut_device(c, p, nullptr);
    if (rc) return rc;
    if (out) HIPCHK(c, hipMemcpyAsync(out, c->d_sky_h, (size_t)c->sw * c->sh * 8, hipMemcpyDeviceToHost, c->pro));
    HIPCHK(c, hipStreamSynchronize(c->pro));
    return CSKY_OK;
}

int csky_render_clouds_device(csky_ctx* c, const csky_cloud_params* p, int tile_w, const csky_bands* bands, void* d_out, size_t pitch, void* hip_stream) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_render_clouds_device: ctx is NULL");
    if (!d_out) return fail(c, CSKY_ERR_INVALID, "csky_render_clouds_device: d_out is NULL");
    int rc; if ((rc = bind(c))) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return clouds_dev(c, p, tile_w, bands, (uint2*)d_out, pitch, s, nullptr, true);
}

int csky_render_clouds(csky_ctx* c, const csky_cloud_params* p, int tile_w, int tile_h, uint16_t* out, size_t pitch) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_render_clouds: ctx is NULL");
    if (tile_w < 1 || tile_h < 1) return fail(c, CSKY_ERR_INVALID, "csky_render_clouds: empty tile");
    if (out && (pitch < (size_t)tile_w * 8)) return fail(c, CSKY_ERR_INVALID, "csky_render_clouds: row_pitch_bytes < tile_w*8");
    int rc; if ((rc = bind(c))) return rc;
    if ((rc = ensure_frame(c, (size_t)tile_w * tile_h))) return rc;
    csky_bands b = {tile_h, 0, 1, 1};
    HIPCHK(c, hipMemsetAsync(c->d_stats, 0, 16, c->stream));
    if ((rc = clouds_dev(c, p, tile_w, &b, c->d_frame, (size_t)tile_w * 8, c->stream, c->d_stats, true))) return rc;
    unsigned long long st[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(st, c->d_stats, 16, hipMemcpyDeviceToHost, c->stream));
    if (out) HIPCHK(c, hipMemcpy2DAsync(out, pitch, c->d_frame, (size_t)tile_w * 8, (size_t)tile_w * 8, tile_h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->last_stats.rays = (uint64_t)tile_w * tile_h; c->last_stats.incloud_samples = st[0]; c->last_stats.primary_samples = st[1] * (uint64_t)c->primary_steps;
    return CSKY_OK;
}

// ---- asynchronous host form: submit / collect over a ring of pinned frames (cloudsky.h) -----------------------------------------
namespace {
int host_slot_prepare(csky_ctx* c, csky_ctx::HostSlot& hs, size_t px, bool need_device) {
    if (!hs.s) HIPCHK(c, hipStreamCreateWithFlags(&hs.s, hipStreamNonBlocking));
    if (!hs.done) HIPCHK(c, hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
    if (hs.px < px) {
        HIPCHK(c, hipStreamSynchronize(hs.s));
        if (hs.d) { (void)hipFree(hs.d); hs.d = nullptr; }
        if (hs.h) { (void)hipHostFree(hs.h); hs.h = nullptr; }
        hs.px = 0;
        if (need_device) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&hs.d), px * 8));
        HIPCHK(c, hipHostMalloc(&hs.h, px * 8, hipHostMallocDefault));       // pinned: the device-to-host copy is a real asynchronous DMA
        hs.px = px;
    } else if (need_device && !hs.d) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&hs.d), hs.px * 8));
    }
    return CSKY_OK;
}
}  // namespace

int csky_set_host_ring(csky_ctx* c, int slots) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_host_ring: ctx is NULL");
    if (slots < 1 || slots > RING) return fail(c, CSKY_ERR_INVALID, "csky_set_host_ring: 1 .. 8 frames");
    for (auto& hs : c->hring) if (hs.busy) return fail(c, CSKY_ERR_STATE, "csky_set_host_ring: collect the outstanding tickets first");
    c->hslots = slots;
    return csky_set_frames_in_flight(c, slots >= 2 ? 2 : 1);    // launch policy: two frames in flight is the best choice for whole frames (csky_set_frames_in_flight)
}

int csky_submit_clouds(csky_ctx* c, const csky_cloud_params* p, int tile_w, int tile_h, int64_t* ticket) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_submit_clouds: ctx is NULL");
    if (!ticket) return fail(c, CSKY_ERR_INVALID, "csky_submit_clouds: ticket is NULL");
    if (tile_w < 1 || tile_h < 1 || tile_w > 16384 || tile_h > 16384) return fail(c, CSKY_ERR_INVALID, "csky_submit_clouds: tile size out of range");
    int rc; if ((rc = bind(c))) return rc;
    csky_ctx::HostSlot& hs = c->hring[c->next_ticket % c->hslots];
    if (hs.busy) return fail(c, CSKY_ERR_STATE, "csky_submit_clouds: %d frames are already in flight (csky_set_host_ring); collect ticket %lld first", c->hslots, (long long)hs.ticket);
    const size_t px = (size_t)tile_w * tile_h;
    if ((rc = host_slot_prepare(c, hs, px, true))) return rc;
    const csky_bands b = {tile_h, 0, 1, 1};
    if ((rc = clouds_dev(c, p, tile_w, &b, hs.d, (size_t)tile_w * 8, hs.s, nullptr, true))) return rc;
    HIPCHK(c, hipMemcpyAsync(hs.h, hs.d, px * 8, hipMemcpyDeviceToHost, hs.s));
    HIPCHK(c, hipEventRecord(hs.done, hs.s));
    hs.busy = true; hs.ticket = c->next_ticket; hs.w = tile_w; hs.hh = tile_h;
    *ticket = c->next_ticket++;
    return CSKY_OK;
}

int csky_collect(csky_ctx* c, int64_t ticket, const uint16_t** frame, size_t* bytes) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_collect: ctx is NULL");
    if (!frame) return fail(c, CSKY_ERR_INVALID, "csky_collect: frame is NULL");
    csky_ctx::HostSlot* hs = nullptr;
    for (auto& k : c->hring) if (k.busy && k.ticket == ticket) hs = &k;
    if (!hs) return fail(c, CSKY_ERR_STATE, "csky_collect: ticket %lld is not outstanding (never submitted, or collected already)", (long long)ticket);
    int rc; if ((rc = bind(c))) return rc;
    HIPCHK(c, hipEventSynchronize(hs->done));
    hs->busy = false;
    *frame = reinterpret_cast<const uint16_t*>(hs->h);
    if (bytes) *bytes = (size_t)hs->w * hs->hh * 8;
    return CSKY_OK;
}

int csky_poll(csky_ctx* c, int64_t ticket) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_poll: ctx is NULL");
    for (auto& k : c->hring)
        if (k.busy && k.ticket == ticket) {
            const hipError_t e = hipEventQuery(k.done);
            if (e == hipSuccess) return 1;
            if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
            return fail(c, CSKY_ERR_HIP, "csky_poll: %s", hipGetErrorString(e));
        }
    return fail(c, CSKY_ERR_STATE, "csky_poll: ticket %lld is not outstanding", (long long)ticket);
}

// ---- zero-copy interop: a frame that lives in memory another API allocated (cloudsky.h; gdext/unverified/zero_copy_vulkan.c is the Vulkan half) ----
struct csky_external_frame { int device = 0; hipExternalMemory_t mem = nullptr; void* d_ptr = nullptr; size_t bytes = 0; hipExternalSemaphore_t sem = nullptr; hipEvent_t fence = nullptr; bool fenced = false; };

// Does descriptor `a` still name the open file description that `b` names?  1 yes, 0 no, -1 cannot tell.  kcmp(KCMP_FILE) compares the kernel objects
// themselves (device + inode identity is the same for EVERY dma-buf / anon-inode descriptor on kernels that share one anon inode, so it cannot
// tell "ours" from a foreign descriptor that reused the number); where the kernel has no kcmp the answer is "cannot tell".
static int same_open_file(int a, int b) {
#ifdef SYS_kcmp
    const long r = syscall(SYS_kcmp, (long)getpid(), (long)getpid(), 0L /* KCMP_FILE */, (long)a, (long)b);
    if (r == 0) return 1;
    if (r > 0) return 0;
    if (errno == EBADF) return 0;                            // `a` is closed: whoever was handed the number consumed it
#endif
    return -1;
}

int csky_external_frame_import_fd(csky_ctx* c, int opaque_fd, size_t allocation_bytes, size_t offset, size_t frame_bytes, csky_external_frame** out, void** d_ptr) {
    if (!c || !out || !d_ptr) return fail(c, CSKY_ERR_INVALID, "csky_external_frame_import_fd: NULL argument");
    *out = nullptr; *d_ptr = nullptr;
    if (opaque_fd < 0 || frame_bytes == 0 || frame_bytes > allocation_bytes || offset > allocation_bytes - frame_bytes) return fail(c, CSKY_ERR_INVALID, "csky_external_frame_import_fd: bad fd / sizes");
    int rc; if ((rc = bind(c))) return rc;
    csky_external_frame* f = new (std::nothrow) csky_external_frame();
    if (!f) return fail(c, CSKY_ERR_INVALID, "csky_external_frame_import_fd: out of host memory");
    f->device = c->device; f->bytes = frame_bytes;
    hipExternalMemoryHandleDesc md; memset(&md, 0, sizeof md);
    // The runtime takes ownership of the fd it is given on success and says nothing about failure, so it gets a DUPLICATE: on ANY failure the
    // caller still owns opaque_fd (and only it); on success the library closes the caller's fd, as documented.
    const int dfd = dup(opaque_fd);
    if (dfd < 0) { delete f; return fail(c, CSKY_ERR_INVALID, "csky_external_frame_import_fd: dup(fd) failed"); }
    md.type = hipExternalMemoryHandleTypeOpaqueFd; md.handle.fd = dfd; md.size = allocation_bytes;
    hipError_t e = hipImportExternalMemory(&f->mem, &md);
    if (e != hipSuccess) (void)close(dfd);
    else {
        // Who owns the duplicate now is decided HERE, once, and the number is never looked at again (a later descriptor -- of this process, the engine
        // or the Vulkan driver -- may reuse it).  CUDA's convention: the runtime consumed it at import.  ROCm's CLR maps the dma-buf during the call
        // and does not say; a runtime that keeps neither the number nor closes it would leak one descriptor per imported frame.  So: if the number
        // provably still names OUR open file description (same kernel object as the caller's descriptor, which nobody else can have closed), the
        // runtime did not consume it and the mapping no longer needs it: close it now.  "No" or "cannot tell" (no kcmp in this kernel): leave it --
        // leaking a descriptor is the safe side of closing a foreign one.
        if (same_open_file(dfd, opaque_fd) == 1) (void)close(dfd);
    }
    if (e == hipSuccess) {
        hipExternalMemoryBufferDesc bd; memset(&bd, 0, sizeof bd);
        bd.offset = offset; bd.size = frame_bytes;
        e = hipExternalMemoryGetMappedBuffer(&f->d_ptr, f->mem, &bd);
    }
    if (e != hipSuccess) { csky_external_frame_release(f); return fail(c, CSKY_ERR_HIP, "csky_external_frame_import_fd: %s", hipGetErrorString(e)); }
    (void)close(opaque_fd);                                   // success: the library owns the memory object now
    *out = f; *d_ptr = f->d_ptr;
    return CSKY_OK;
}
int csky_external_frame_import_semaphore_fd(csky_ctx* c, csky_external_frame* f, int opaque_fd) {
    if (!c || !f || opaque_fd < 0) return fail(c, CSKY_ERR_INVALID, "csky_external_frame_import_semaphore_fd: bad argument");
    int rc; if ((rc = bind(c))) return rc;
    if (f->sem) { (void)hipDestroyExternalSemaphore(f->sem); f->sem = nullptr; }
    hipExternalSemaphoreHandleDesc sd; memset(&sd, 0, sizeof sd);
    sd.type = hipExternalSemaphoreHandleTypeOpaqueFd; sd.handle.fd = opaque_fd;
    HIPCHK(c, hipImportExternalSemaphore(&f->sem, &sd));
    return CSKY_OK;
}
int csky_external_frame_signal(csky_ctx* c, csky_external_frame* f, void* hip_stream) {
    if (!c || !f || !f->sem) return fail(c, CSKY_ERR_INVALID, "csky_external_frame_signal: no semaphore imported");
    int rc; if ((rc = bind(c))) return rc;
    hipExternalSemaphoreSignalParams sp; memset(&sp, 0, sizeof sp);
    HIPCHK(c, hipSignalExternalSemaphoresAsync(&f->sem, &sp, 1, hip_stream ? (hipStream_t)hip_stream : c->stream));
    return CSKY_OK;
}
// Host-side ordering for runtimes that cannot import a semaphore (ROCm 7.2 on Linux answers hipErrorNotSupported for every handle type,
// profiles/r03/external_semaphore_probe.txt): an event behind the march that the host polls before it lets the engine sample the image.
int csky_external_frame_fence(csky_ctx* c, csky_external_frame* f, void* hip_stream) {
    if (!c || !f) return fail(c, CSKY_ERR_INVALID, "csky_external_frame_fence: NULL argument");
    int rc; if ((rc = bind(c))) return rc;
    if (!f->fence) HIPCHK(c, hipEventCreateWithFlags(&f->fence, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(f->fence, hip_stream ? (hipStream_t)hip_stream : c->stream));
    f->fenced = true;
    return CSKY_OK;
}
int csky_external_frame_ready(csky_ctx* c, csky_external_frame* f) {
    if (!c || !f) return fail(c, CSKY_ERR_INVALID, "csky_external_frame_ready: NULL argument");
    if (!f->fenced) return fail(c, CSKY_ERR_STATE, "csky_external_frame_ready: no fence recorded (csky_external_frame_fence)");
    const hipError_t e = hipEventQuery(f->fence);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return fail(c, CSKY_ERR_HIP, "csky_external_frame_ready: %s", hipGetErrorString(e));
}
int csky_external_frame_wait(csky_ctx* c, csky_external_frame* f) {
    if (!c || !f) return fail(c, CSKY_ERR_INVALID, "csky_external_frame_wait: NULL argument");
    if (!f->fenced) return fail(c, CSKY_ERR_STATE, "csky_external_frame_wait: no fence recorded (csky_external_frame_fence)");
    int rc; if ((rc = bind(c))) return rc;
    HIPCHK(c, hipEventSynchronize(f->fence));
    return CSKY_OK;
}
void csky_external_frame_release(csky_external_frame* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->fence) { (void)hipEventSynchronize(f->fence); (void)hipEventDestroy(f->fence); }   // never unmap memory a march may still be writing
    if (f->sem) (void)hipDestroyExternalSemaphore(f->sem);
    if (f->mem) (void)hipDestroyExternalMemory(f->mem);      // unmaps d_ptr
    // (the duplicate descriptor the runtime was handed was settled at import: csky_external_frame_import_fd)
    delete f;
}

int csky_read_transmittance(csky_ctx* c, uint16_t* out, int* w, int* h) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_read_transmittance: ctx is NULL");
    if (!c->have_trans) return fail(c, CSKY_ERR_STATE, "csky_read_transmittance: LUT not rendered yet");
    int rc; if ((rc = bind(c))) return rc;
    if (w) *w = c->tw; if (h) *h = c->th;
    if (out) { HIPCHK(c, hipStreamSynchronize(c->pro)); HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipMemcpy(out, c->d_trans_h, (size_t)c->tw * c->th * 8, hipMemcpyDeviceToHost)); }
    return CSKY_OK;
}
int csky_read_sky_lut(csky_ctx* c, uint16_t* out, int* w, int* h) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_read_sky_lut: ctx is NULL");
    if (!c->have_sky) return fail(c, CSKY_ERR_STATE, "csky_read_sky_lut: LUT not rendered yet");
    if (!c->sky_in_memory) return fail(c, CSKY_ERR_STATE, "csky_read_sky_lut: the last LUT went to the caller as rows (csky_render_sky_lut_rows_device), this context holds none");
    int rc; if ((rc = bind(c))) return rc;
    if (w) *w = c->sw; if (h) *h = c->sh;
    if (out) {
        HIPCHK(c, hipStreamSynchronize(c->pro));
        for (hipEvent_t ev : c->lut_writers) HIPCHK(c, hipEventSynchronize(ev));       // rows written by the other devices of a csky_multi handle
        HIPCHK(c, hipMemcpy(out, c->d_sky_h, (size_t)c->sw * c->sh * 8, hipMemcpyDeviceToHost));
    }
    return CSKY_OK;
}

int csky_set_kernel_timing(csky_ctx* c, int enabled) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_set_kernel_timing: ctx is NULL");
    int rc; if ((rc = bind(c))) return rc;
    c->kt_on = enabled != 0; c->kt_count = 0;
    if (c->kt_on && c->kt_ev.empty()) {      // the first pool is made HERE, not by the first timed launch: 512 hipEventCreate calls are ~1 ms of host time, 3 % of a 20-frame timed region
        if ((rc = grow_timing_pool(c, 512))) { c->kt_on = false; return rc; }
    }
    return CSKY_OK;
}

int csky_get_kernel_ms(csky_ctx* c, float* total_ms, int* launches) {
    if (!c || !total_ms || !launches) return fail(c, CSKY_ERR_INVALID, "csky_get_kernel_ms: NULL argument");
    int rc; if ((rc = bind(c))) return rc;
    const int n = c->kt_count;                                 // every launch since the last call (the pool grows on demand)
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        HIPCHK(c, hipEventSynchronize(c->kt_ev[(size_t)i * 2 + 1]));
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->kt_ev[(size_t)i * 2], c->kt_ev[(size_t)i * 2 + 1]));
        sum += ms;
    }
    *total_ms = sum; *launches = n; c->kt_count = 0;
    return CSKY_OK;
}

int csky_time_clouds(csky_ctx* c, const csky_cloud_params* p, int tile_w, const csky_bands* bands, int warmup, int iters, float* mean_ms, csky_cloud_stats* stats) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_time_clouds: ctx is NULL");
    if (iters < 1 || warmup < 0 || !mean_ms) return fail(c, CSKY_ERR_INVALID, "csky_time_clouds: bad iters/warmup/mean_ms");
    int rc; if ((rc = bind(c))) return rc;
    if ((rc = check_bands(c, bands, tile_w))) return rc;
    const size_t rows = (size_t)bands->n_bands * bands->band_rows;
    if ((rc = ensure_frame(c, (size_t)tile_w * (rows ? rows : 1)))) return rc;
    const size_t pitch = (size_t)tile_w * 8;
    HIPCHK(c, hipMemsetAsync(c->d_stats, 0, 16, c->stream));
    if ((rc = clouds_dev(c, p, tile_w, bands, c->d_frame, pitch, c->stream, c->d_stats, true))) return rc;   // stats launch (+ frame setup)
    for (int i = 0; i < warmup; i++) if ((rc = clouds_dev(c, p, tile_w, bands, c->d_frame, pitch, c->stream, nullptr, false))) return rc;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < iters; i++) if ((rc = clouds_dev(c, p, tile_w, bands, c->d_frame, pitch, c->stream, nullptr, false))) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *mean_ms = ms / (float)iters;
    unsigned long long st[2] = {0, 0};
    HIPCHK(c, hipMemcpy(st, c->d_stats, 16, hipMemcpyDeviceToHost));
    c->last_stats.rays = (uint64_t)tile_w * rows; c->last_stats.incloud_samples = st[0]; c->last_stats.primary_samples = st[1] * (uint64_t)c->primary_steps;
    if (stats) *stats = c->last_stats;
    return CSKY_OK;
}

static int composite_impl(csky_ctx* c, const csky_composite_params* p, const csky_view* view, const uint16_t* cloud_from, const uint16_t* cloud_to,
                          const uint16_t* sky_from, const uint16_t* sky_to, uint16_t* out);
int csky_composite_sky(csky_ctx* c, const csky_composite_params* p, const uint16_t* cloud_from, const uint16_t* cloud_to, const uint16_t* sky_from,
                       const uint16_t* sky_to, uint16_t* out) {
    return composite_impl(c, p, nullptr, cloud_from, cloud_to, sky_from, sky_to, out);
}
int csky_composite_view(csky_ctx* c, const csky_composite_params* p, const csky_view* view, const uint16_t* cloud_from, const uint16_t* cloud_to,
                        const uint16_t* sky_from, const uint16_t* sky_to, uint16_t* out) {
    if (!view) return fail(c, CSKY_ERR_INVALID, "csky_composite_view: view is NULL");
    if (!(view->fov_y_degrees > 0.0f && view->fov_y_degrees < 180.0f)) return fail(c, CSKY_ERR_INVALID, "csky_composite_view: fov_y_degrees must be in (0, 180)");
    return composite_impl(c, p, view, cloud_from, cloud_to, sky_from, sky_to, out);
}
static int composite_impl(csky_ctx* c, const csky_composite_params* p, const csky_view* view, const uint16_t* cloud_from, const uint16_t* cloud_to,
                          const uint16_t* sky_from, const uint16_t* sky_to, uint16_t* out) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_composite_sky: ctx is NULL");
    if (!p || !cloud_from || !cloud_to || !sky_from || !sky_to || !out) return fail(c, CSKY_ERR_INVALID, "csky_composite_sky: NULL argument");
    if (p->out_w < 1 || p->out_h < 1 || p->cloud_w < 1 || p->cloud_h < 1 || p->sky_w < 1 || p->sky_h < 1 || p->out_w > 16384 || p->out_h > 16384)
        return fail(c, CSKY_ERR_INVALID, "csky_composite_sky: bad image size");
    int rc; if ((rc = bind(c))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->pro));                                                // the transmittance LUT may have been rendered there
    if (!c->have_trans && (rc = render_trans_dev(c, 256, 64, c->stream))) return rc;       // source_transmittance, clouds_material.tres
    const size_t cb = (size_t)p->cloud_w * p->cloud_h * 8, sb = (size_t)p->sky_w * p->sky_h * 8, ob = (size_t)p->out_w * p->out_h * 8;
    const size_t need = 2 * cb + 2 * sb + ob;
    if (c->composite_cap < need) {                            // grow-only scratch: no allocation per call once the sizes have been seen
        if (c->d_composite) { (void)hipFree(c->d_composite); c->d_composite = nullptr; c->composite_cap = 0; }
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_composite), need));
        c->composite_cap = need;
    }
    uint8_t* d = c->d_composite;
    hipError_t e = hipSuccess;
    auto up = [&](size_t off, const void* src, size_t n) { if (e == hipSuccess) e = hipMemcpyAsync(d + off, src, n, hipMemcpyHostToDevice, c->stream); };
    up(0, cloud_from, cb); up(cb, cloud_to, cb); up(2 * cb, sky_from, sb); up(2 * cb + sb, sky_to, sb);
    CompositeArgs a;
    a.cloud_from = reinterpret_cast<const uint16_t*>(d); a.cloud_to = reinterpret_cast<const uint16_t*>(d + cb); a.cw = p->cloud_w; a.ch = p->cloud_h;
    a.sky_from = reinterpret_cast<const uint16_t*>(d + 2 * cb); a.sky_to = reinterpret_cast<const uint16_t*>(d + 2 * cb + sb); a.sw = p->sky_w; a.sh = p->sky_h;
    a.trans = c->d_trans_f; a.tw = c->tw; a.th = c->th;
    a.blend_amount = p->blend_amount; a.sun_disk_scale = p->sun_disk_scale;
    a.sun[0] = p->light_direction[0]; a.sun[1] = p->light_direction[1]; a.sun[2] = p->light_direction[2];
    a.out_w = p->out_w; a.out_h = p->out_h;
    a.view_mode = 0; a.tan_half_fov_y = 1.0f; a.aspect = 1.0f;
    for (int k = 0; k < 9; k++) a.cam[k] = (k % 4 == 0) ? 1.0f : 0.0f;
    if (view) {
        a.view_mode = 1;
        for (int k = 0; k < 9; k++) a.cam[k] = view->basis[k];
        a.tan_half_fov_y = tanf(view->fov_y_degrees * 0.5f * 3.14159265358979323846f / 180.0f);
        a.aspect = (float)p->out_w / (float)p->out_h;
    }
    if (e == hipSuccess) e = launch_composite(a, reinterpret_cast<uint2*>(d + 2 * cb + 2 * sb), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + 2 * cb + 2 * sb, ob, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, CSKY_ERR_HIP, "csky_composite_sky: %s", hipGetErrorString(e));
    return CSKY_OK;
}

int csky_generate_shape_noise_device(csky_ctx* c, uint32_t seed, int n, uint8_t* out_rgba8) { return csky_generate_shape_noise_tuned_device(c, seed, n, nullptr, out_rgba8); }
int csky_generate_shape_noise_tuned_device(csky_ctx* c, uint32_t seed, int n, const csky_shape_noise_params* params, uint8_t* out_rgba8) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_generate_shape_noise_device: ctx is NULL");
    if (!out_rgba8 || n < 8 || n > 512 || (n & (n - 1))) return fail(c, CSKY_ERR_INVALID, "csky_generate_shape_noise_device: n must be a power of two in [8, 512]");
    ShapeNoiseParams P = shape_noise_defaults();
    if (params) {
        if (csky_check_shape_noise_params(params, n)) return fail(c, CSKY_ERR_INVALID, "csky_generate_shape_noise_tuned_device: %s", csky_assets_last_error());
        memcpy(&P, params, sizeof P);
    }
    int rc; if ((rc = bind(c))) return rc;
    const size_t bytes = (size_t)n * n * n * 4;
    uint32_t* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), bytes));
    hipError_t e = launch_shape_noise(seed, n, P, d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_rgba8, d, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, CSKY_ERR_HIP, "csky_generate_shape_noise_device: %s", hipGetErrorString(e));
    return CSKY_OK;
}

int csky_get_cloud_stats(csky_ctx* c, csky_cloud_stats* stats) {
    if (!c || !stats) return fail(c, CSKY_ERR_INVALID, "csky_get_cloud_stats: NULL argument");
    *stats = c->last_stats; return CSKY_OK;
}


int csky_copy_sky_lut_device(csky_ctx* c, void* d_out, void* hip_stream) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_copy_sky_lut_device: ctx is NULL");
    if (!d_out) return fail(c, CSKY_ERR_INVALID, "csky_copy_sky_lut_device: d_out is NULL");
    if (!c->have_sky) return fail(c, CSKY_ERR_STATE, "csky_copy_sky_lut_device: LUT not rendered yet");
    if (!c->sky_in_memory) return fail(c, CSKY_ERR_STATE, "csky_copy_sky_lut_device: the last LUT went to the caller as rows (csky_render_sky_lut_rows_device), this context holds none");
    int rc; if ((rc = bind(c))) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    // the copy runs on the prologue stream right behind the LUT's render (a later render goes to the other ring slot and, like every
    // writer of a slot, is queued behind this reader on the same stream); the caller's stream then waits for it
    for (hipEvent_t ev : c->lut_writers) HIPCHK(c, hipStreamWaitEvent(c->pro, ev, 0));   // rows written by the other devices of a csky_multi handle
    HIPCHK(c, hipMemcpyAsync(d_out, c->d_sky_h, (size_t)c->sw * c->sh * 8, hipMemcpyDeviceToDevice, c->pro));
    HIPCHK(c, hipEventRecord(c->ev_copy, c->pro));
    HIPCHK(c, hipStreamWaitEvent(s, c->ev_copy, 0));
    return CSKY_OK;
}

int csky_interleave_bands_device(csky_ctx* c, const void* d_gathered, size_t member_stride_bytes, int members, size_t band_bytes, int total_bands, void* d_frame, void* hip_stream) {
    if (!c) return fail(nullptr, CSKY_ERR_INVALID, "csky_interleave_bands_device: ctx is NULL");
    if (!d_gathered || !d_frame || members < 1 || total_bands < 0) return fail(c, CSKY_ERR_INVALID, "csky_interleave_bands_device: NULL argument or bad counts");
    if (band_bytes % 16 || member_stride_bytes % 16 || ((uintptr_t)d_gathered & 15) || ((uintptr_t)d_frame & 15))
        return fail(c, CSKY_ERR_INVALID, "csky_interleave_bands_device: band size, member stride and both pointers must be multiples of 16 bytes");
    const size_t local = ((size_t)total_bands + members - 1) / members;
    if (local * band_bytes > member_stride_bytes) return fail(c, CSKY_ERR_INVALID, "csky_interleave_bands_device: a member's %zu bands of %zu bytes do not fit its stride of %zu", local, band_bytes, member_stride_bytes);
    int rc; if ((rc = bind(c))) return rc;
    HIPCHK(c, launch_interleave_bands(d_gathered, member_stride_bytes, members, band_bytes, total_bands, d_frame, hip_stream ? (hipStream_t)hip_stream : c->stream));
    return CSKY_OK;
}

// ---- multi-GPU: n contexts, one frame on the first device written by peer stores (cloudsky.h) ----------------------------
constexpr int MULTI_SLOTS = 8;            // frames in flight over all groups (csky_multi_set_frames_in_flight x csky_multi_set_groups)
struct csky_multi {
    std::vector<csky_ctx*> ctx;
    std::vector<hipEvent_t> ev_done[MULTI_SLOTS];   // [frame slot][device]: its march (and staged copy) of that frame has finished
    hipEvent_t ev_begin[MULTI_SLOTS] = {};          // on the first device: the consumer stream's position when the frame was requested
    std::vector<hipStream_t> side[RING - 1];        // frames in flight: the streams of a device's 2nd..4th frame in flight (the 1st: the context's own)
    int fif = 1, groups = 1;              // frames in flight PER GROUP, frame groups (csky_multi_set_groups)
    unsigned long long frame_no = 0;
    bool staged = false;                  // CSKY_MULTI_STAGED=1 / csky_multi_set_staged: local band buffer + peer copy instead of in-place peer stores
    std::vector<uint2*> d_stage[RING];    // [per-device frame slot][device]: compact band buffer on that device (staged form)
    std::vector<size_t> stage_px[RING];
    uint2* d_frame = nullptr; size_t frame_px = 0;   // host-buffer form: internal frame on the first device
    std::vector<hipEvent_t> ev_lut;       // [device]: its rows of the sky LUT have been stored into the first device's LUT (csky_multi_render_sky_lut)
    hipEvent_t ev_lut_begin = nullptr;    // on the first device's prologue stream: the readers of the LUT slot about to be rewritten are behind this
    // preconditions + instrumentation (round 6): which devices can store into the first device's memory, and -- while csky_multi_set_timing is on --
    // timing events around every device's march and its staged peer copy of the LAST frame enqueued
    std::vector<int> peer_ok;             // [device]: 1 = hipDeviceCanAccessPeer(device, first) and it was enabled (1 for the first device itself)
    bool all_peer = true;                 // false: some device has no peer access -> staged copies (the runtime bounces them) + the whole LUT on the first device
    bool timing = false;
    std::vector<hipEvent_t> tm[3];        // [0] before the march, [1] after it, [2] after the staged copy; [device]
    std::vector<int> tm_valid;            // [device]: 0 none, 1 march only, 2 march + copy
    char err[512] = {0};
    char warn[512] = {0};
};
namespace {
int mfail(csky_multi* m, int code, const char* fmt, ...) {
    char* dst = m ? m->err : g_err;
    va_list ap; va_start(ap, fmt); vsnprintf(dst, 512, fmt, ap); va_end(ap);
    return code;
}
int mpass(csky_multi* m, int i, int rc) {      // propagate a per-device error text
    if (rc) snprintf(m->err, sizeof m->err, "device %d (index %d): %s", m->ctx[i]->device, i, m->ctx[i]->err);
    return rc;
}
}  // namespace

int csky_multi_create(csky_multi** out, const int* device_ids, int n) {
    if (!out) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_create: out is NULL");
    *out = nullptr;
    if (!device_ids || n < 1 || n > 64) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_create: need 1..64 device ids");
    csky_multi* m = new (std::nothrow) csky_multi();
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_create: out of host memory");
    for (int i = 0; i < n; i++) {
        csky_ctx* c = nullptr;
        const int rc = csky_create(&c, device_ids[i]);
        if (rc) { csky_multi_destroy(m); return rc; }           // g_err already holds csky_create's text
        m->ctx.push_back(c);
    }
    auto bail = [&](int code, const char* what, hipError_t e) { mfail(nullptr, code, "csky_multi_create: %s: %s", what, hipGetErrorString(e)); csky_multi_destroy(m); return code; };
    const int d0 = device_ids[0];
    hipError_t e;
    for (int i = 0; i < n; i++) {
        const int di = device_ids[i];
        if ((e = hipSetDevice(di)) != hipSuccess) return bail(CSKY_ERR_HIP, "hipSetDevice", e);
        for (int sl = 0; sl < MULTI_SLOTS; sl++) {
            hipEvent_t ev = nullptr;
            if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail(CSKY_ERR_HIP, "hipEventCreate", e);
            m->ev_done[sl].push_back(ev);
        }
        for (int k = 0; k < RING - 1; k++) {
            hipStream_t st = nullptr;
            if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return bail(CSKY_ERR_HIP, "hipStreamCreate", e);
            m->side[k].push_back(st);
        }
        for (int k = 0; k < RING; k++) { m->d_stage[k].push_back(nullptr); m->stage_px[k].push_back(0); }
        { hipEvent_t ev = nullptr; if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail(CSKY_ERR_HIP, "hipEventCreate", e); m->ev_lut.push_back(ev); }
        for (int k = 0; k < 3; k++) m->tm[k].push_back(nullptr);
        m->tm_valid.push_back(0);
        int ok = 1;
        if (di != d0) {                                          // the march on device di stores into the frame on d0: xGMI peer access
            int can = 0;
            if ((e = hipDeviceCanAccessPeer(&can, di, d0)) != hipSuccess) return bail(CSKY_ERR_HIP, "hipDeviceCanAccessPeer", e);
            if (can) {
                e = hipDeviceEnablePeerAccess(d0, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
                (void)hipGetLastError();
            }
            ok = can;
        }
        // test hook: pretend device INDEX i (> 0) has no peer access -- also when it IS the first device's GPU, which is all a single-GPU box can offer
        if (const char* fk = getenv("CSKY_MULTI_FAKE_NO_PEER")) if (i > 0 && atoi(fk) == i) ok = 0;
        m->peer_ok.push_back(ok);
        if (!ok) {
            // No peer access between this device and the first one (no xGMI / PCIe P2P route, IOMMU policy, a container without the devices' links): the
            // in-place form cannot work for it.  Not fatal: the whole handle falls back to the STAGED form -- every device renders into a local band
            // buffer and hipMemcpy2DAsync moves the bands (the runtime bounces them through host memory where it must) -- and the first device renders the
            // whole sky LUT itself.  Slower; the caller is told (csky_multi_last_warning), and csky_multi_get_stats says which devices.
            const size_t at = strlen(m->warn);
            snprintf(m->warn + at, sizeof m->warn - at, "%sdevice %d (index %d) has no peer access to device %d", at ? "; " : "csky_multi_create: ", di, i, d0);
            m->all_peer = false;
        }
    }
    if (!m->all_peer) {
        const size_t at = strlen(m->warn);
        snprintf(m->warn + at, sizeof m->warn - at, ": falling back to staged band copies and a whole sky LUT on the first device (slower than in-place xGMI stores)");
        m->staged = true;
    }
    if ((e = hipSetDevice(d0)) != hipSuccess) return bail(CSKY_ERR_HIP, "hipSetDevice", e);
    for (int sl = 0; sl < MULTI_SLOTS; sl++)
        if ((e = hipEventCreateWithFlags(&m->ev_begin[sl], hipEventDisableTiming)) != hipSuccess) return bail(CSKY_ERR_HIP, "hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&m->ev_lut_begin, hipEventDisableTiming)) != hipSuccess) return bail(CSKY_ERR_HIP, "hipEventCreate", e);
    if (const char* se = getenv("CSKY_MULTI_STAGED")) m->staged = (atoi(se) != 0) || !m->all_peer;   // A/B switch for the driver's 8-GPU node (never off without peer access)
    *out = m;
    return CSKY_OK;
}

void csky_multi_destroy(csky_multi* m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); i++) {
        (void)hipSetDevice(m->ctx[i]->device);
        (void)hipDeviceSynchronize();
        for (int sl = 0; sl < MULTI_SLOTS; sl++) if (i < m->ev_done[sl].size() && m->ev_done[sl][i]) (void)hipEventDestroy(m->ev_done[sl][i]);
        for (int k = 0; k < RING - 1; k++) if (i < m->side[k].size() && m->side[k][i]) (void)hipStreamDestroy(m->side[k][i]);
        for (int k = 0; k < RING; k++) if (i < m->d_stage[k].size() && m->d_stage[k][i]) (void)hipFree(m->d_stage[k][i]);
        if (i < m->ev_lut.size() && m->ev_lut[i]) (void)hipEventDestroy(m->ev_lut[i]);
        for (int k = 0; k < 3; k++) if (i < m->tm[k].size() && m->tm[k][i]) (void)hipEventDestroy(m->tm[k][i]);
    }
    if (!m->ctx.empty()) {
        (void)hipSetDevice(m->ctx[0]->device);
        for (int sl = 0; sl < MULTI_SLOTS; sl++) if (m->ev_begin[sl]) (void)hipEventDestroy(m->ev_begin[sl]);
        if (m->ev_lut_begin) (void)hipEventDestroy(m->ev_lut_begin);
        m->ctx[0]->lut_writers.clear();
        if (m->d_frame) (void)hipFree(m->d_frame);
    }
    for (csky_ctx* c : m->ctx) csky_destroy(c);
    delete m;
}

int csky_multi_device_count(const csky_multi* m) { return m ? (int)m->ctx.size() : 0; }
csky_ctx* csky_multi_ctx(csky_multi* m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }
const char* csky_multi_last_error(const csky_multi* m) { return m ? m->err : g_err; }
const char* csky_multi_last_warning(const csky_multi* m) { return m ? m->warn : ""; }
int csky_multi_set_timing(csky_multi* m, int enabled) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_set_timing: handle is NULL");
    const int rc = csky_multi_sync(m); if (rc) return rc;
    if (enabled)
        for (size_t i = 0; i < m->ctx.size(); i++) {
            if ((hipSetDevice(m->ctx[i]->device)) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_set_timing: hipSetDevice failed");
            for (int k = 0; k < 3; k++) if (!m->tm[k][i] && hipEventCreate(&m->tm[k][i]) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_set_timing: hipEventCreate failed");
        }
    m->timing = enabled != 0;
    for (auto& v : m->tm_valid) v = 0;
    return CSKY_OK;
}
int csky_multi_get_stats(csky_multi* m, csky_multi_stats* out) {
    if (!m || !out) return mfail(m, CSKY_ERR_INVALID, "csky_multi_get_stats: NULL argument");
    memset(out, 0, sizeof *out);
    const int rc = csky_multi_sync(m); if (rc) return rc;       // the events of the last frame must have completed
    const int n = (int)m->ctx.size();
    out->n_devices = n; out->staged = m->staged ? 1 : 0; out->all_peer = m->all_peer ? 1 : 0; out->groups = m->groups; out->frames_in_flight = m->fif;
    out->timing = m->timing ? 1 : 0;
    for (int i = 0; i < n && i < CSKY_MULTI_STATS_MAX; i++) {
        out->device_id[i] = m->ctx[i]->device; out->peer_access[i] = m->peer_ok[i];
        out->march_ms[i] = out->copy_ms[i] = -1.0f;
        if (m->tm_valid[i] >= 1) {
            if (hipSetDevice(m->ctx[i]->device) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_get_stats: hipSetDevice failed");
            float ms = 0.0f;
            // (the first device's share runs on the CALLER's consumer stream, which csky_multi_sync does not own: wait for the timing events themselves)
            if (hipEventSynchronize(m->tm[m->tm_valid[i] >= 2 ? 2 : 1][i]) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_get_stats: hipEventSynchronize failed");
            if (hipEventElapsedTime(&ms, m->tm[0][i], m->tm[1][i]) == hipSuccess) out->march_ms[i] = ms; else (void)hipGetLastError();
            if (m->tm_valid[i] >= 2) { if (hipEventElapsedTime(&ms, m->tm[1][i], m->tm[2][i]) == hipSuccess) out->copy_ms[i] = ms; else (void)hipGetLastError(); }
            else out->copy_ms[i] = 0.0f;                         // in-place form: the stores ARE the march
        }
    }
    return CSKY_OK;
}

int csky_multi_set_noise(csky_multi* m, const uint8_t* large, const uint8_t* small, const uint8_t* weather) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_set_noise: handle is NULL");
    for (size_t i = 0; i < m->ctx.size(); i++) { const int rc = csky_set_noise(m->ctx[i], large, small, weather); if (rc) return mpass(m, (int)i, rc); }
    return CSKY_OK;
}
int csky_multi_set_noise_mips(csky_multi* m, const uint8_t* large_chain, const uint8_t* small_chain, const uint8_t* weather) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_set_noise_mips: handle is NULL");
    for (size_t i = 0; i < m->ctx.size(); i++) { const int rc = csky_set_noise_mips(m->ctx[i], large_chain, small_chain, weather); if (rc) return mpass(m, (int)i, rc); }
    return CSKY_OK;
}
int csky_multi_set_march(csky_multi* m, int primary_steps, int light_steps) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_set_march: handle is NULL");
    for (size_t i = 0; i < m->ctx.size(); i++) { const int rc = csky_set_march(m->ctx[i], primary_steps, light_steps); if (rc) return mpass(m, (int)i, rc); }
    return CSKY_OK;
}
int csky_multi_set_frames_in_flight(csky_multi* m, int frames) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_set_frames_in_flight: handle is NULL");
    if (frames < 1 || frames > RING) return mfail(m, CSKY_ERR_INVALID, "csky_multi_set_frames_in_flight: 1 .. 8 per frame group (the per-device rings are eight deep)");
    if (frames * m->groups > MULTI_SLOTS) return mfail(m, CSKY_ERR_INVALID, "csky_multi_set_frames_in_flight: frames x groups must be <= %d", MULTI_SLOTS);
    for (size_t i = 0; i < m->ctx.size(); i++) { const int rc = csky_set_frames_in_flight(m->ctx[i], frames); if (rc) return mpass(m, (int)i, rc); }
    m->fif = frames;
    return CSKY_OK;
}
int csky_multi_set_groups(csky_multi* m, int groups) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_set_groups: handle is NULL");
    const int n = (int)m->ctx.size();
    if (groups < 1 || groups > n || n % groups) return mfail(m, CSKY_ERR_INVALID, "csky_multi_set_groups: the group count must divide the %d devices", n);
    if (groups * m->fif > MULTI_SLOTS) return mfail(m, CSKY_ERR_INVALID, "csky_multi_set_groups: frames in flight x groups must be <= %d", MULTI_SLOTS);
    const int rc = csky_multi_sync(m); if (rc) return rc;       // frames of the old partition may be in flight
    m->groups = groups; m->frame_no = 0;
    return CSKY_OK;
}
int csky_multi_set_staged(csky_multi* m, int staged) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_set_staged: handle is NULL");
    const int rc = csky_multi_sync(m); if (rc) return rc;
    if (!staged && !m->all_peer) return mfail(m, CSKY_ERR_STATE, "csky_multi_set_staged: the in-place form needs peer access from every device to the first (%s)", m->warn);
    m->staged = staged != 0;
    return CSKY_OK;
}
int csky_multi_render_sky_lut(csky_multi* m, const csky_sky_params* p) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_render_sky_lut: handle is NULL");
    if (!p) return mfail(m, CSKY_ERR_INVALID, "csky_multi_render_sky_lut: params is NULL");
    const int w = (int)p->texture_size[0], h = (int)p->texture_size[1];
    if (w < 1 || h < 1 || w > 8192 || h > 8192) return mfail(m, CSKY_ERR_INVALID, "csky_multi_render_sky_lut: texture_size out of range");
    // sky_lut.gd:43-52 renders the LUT once per frame (cloud_sky.gd:187); n devices rendering n whole copies would each spend 33 us of a chip on it,
    // 12 % of a 1/8 frame share.  Device i renders rows i, i + n, ... and stores them, like its bands, straight into the LUT on the first device
    // (the copy a consumer reads: csky_read_sky_lut / csky_copy_sky_lut_device on csky_multi_ctx(m, 0) wait for every writer).  No device's frame
    // set-up reads that copy: each renders the <= 12 texels it filters itself (frame_setup_taps_kernel), for the sun recorded here -- on EVERY
    // device, whatever the group layout (a caller with a static sun renders the LUT once and then frames on all groups).
    const int n = (int)m->ctx.size();
    csky_ctx* c0 = m->ctx[0];
    int rc;
    if (c0->d_sky_h && (c0->sw != w || c0->sh != h))            // a size change re-allocates the first device's LUT slots: no device may still be storing rows into them
        for (int i = 0; i < n; i++) { if ((rc = bind(m->ctx[i]))) return mpass(m, i, rc); if (hipStreamSynchronize(m->ctx[i]->pro) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_sky_lut: hipStreamSynchronize failed"); }
    if ((rc = bind(c0))) return mpass(m, 0, rc);
    if ((rc = ensure_sky(c0, w, h))) return mpass(m, 0, rc);
    const int k = (c0->have_sky && c0->sky_in_memory) ? c0->sky_cur ^ 1 : c0->sky_cur;     // the other ring slot, as in csky_render_sky_lut_device
    // the readers of slot k (device copies of the LUT before last) sit on the first device's prologue stream: every writer queues behind them
    if (hipEventRecord(m->ev_lut_begin, c0->pro) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_sky_lut: hipEventRecord failed");
    for (int i = 0; i < n; i++) {
        csky_ctx* c = m->ctx[i];
        if ((rc = bind(c))) return mpass(m, i, rc);
        if (!c->have_trans && (rc = render_trans_dev(c, 256, 64, c->pro))) return mpass(m, i, rc);   // transmittance_lut.gd:6 default size
        hipError_t e = i ? hipStreamWaitEvent(c->pro, m->ev_lut_begin, 0) : hipSuccess;
        // rows i, i + n, ... stored into the first device's LUT -- or, when some device cannot reach that memory, every row by the first device itself
        if (e == hipSuccess && (m->all_peer || i == 0))
            e = launch_sky_lut_rows(w, h, m->all_peer ? i : 0, m->all_peer ? n : 1, p->sun_direction, c->d_trans_f, c->tw, c->th, reinterpret_cast<uint2*>(c0->sky_h_ring[k]), c0->sky_f_ring[k], c->pro);
        if (e == hipSuccess) e = hipEventRecord(m->ev_lut[i], c->pro);
        if (e != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_sky_lut: device index %d: %s", i, hipGetErrorString(e));
        for (int q = 0; q < 3; q++) c->sky_sun[q] = p->sun_direction[q];
        c->psw = w; c->psh = h; c->sky_partial = true; c->have_sky = true;
        if (i) { c->sky_in_memory = false; c->lut_writers.clear(); }
    }
    c0->sky_cur = k; c0->d_sky_h = c0->sky_h_ring[k]; c0->d_sky_f = c0->sky_f_ring[k]; c0->sky_in_memory = true;
    c0->lut_writers.assign(m->ev_lut.begin() + 1, m->ev_lut.end());      // (its own rows are on its prologue stream, ahead of any reader)
    return CSKY_OK;
}

int csky_multi_render_clouds_device(csky_multi* m, const csky_cloud_params* p, int tile_w, int tile_h, void* d_out, size_t pitch, void* hip_stream) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_render_clouds_device: handle is NULL");
    if (!d_out || !p) return mfail(m, CSKY_ERR_INVALID, "csky_multi_render_clouds_device: NULL argument");
    if (tile_w < 1 || tile_h < 8 || (tile_h & 7)) return mfail(m, CSKY_ERR_INVALID, "csky_multi_render_clouds_device: tile_h must be a positive multiple of 8 (bands are 8 rows)");
    if (pitch % 8 || pitch < (size_t)tile_w * 8) return mfail(m, CSKY_ERR_INVALID, "csky_multi_render_clouds_device: row pitch must be a multiple of 8 and >= tile_w*8");
    const int n_all = (int)m->ctx.size(), total = tile_h / 8;
    // Frame groups (csky_multi_set_groups): consecutive frames go to the G groups in turn; the n/G devices of a group split the frame's bands.
    // Slot = frame number mod (groups x frames in flight): its events, and on every device of the group the stream / ring position
    // slot / groups of that device's frames in flight.
    const int G = m->groups, per = n_all / G, slots = G * m->fif;
    const int slot = (int)(m->frame_no % (unsigned long long)slots), grp = slot % G, dslot = slot / G;   // (frame_no advances only when the frame was enqueued: a failed call must not shift the slot / group rotation)
    csky_ctx* c0 = m->ctx[0];
    int rc; if ((rc = bind(c0))) return mpass(m, 0, rc);
    hipStream_t consumer = hip_stream ? (hipStream_t)hip_stream : c0->stream;
    // no device may store into the frame before the consumer's earlier work on it (reads of the previous frame) is done
    if (hipEventRecord(m->ev_begin[slot], consumer) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds_device: hipEventRecord failed");
    for (int k = 0; k < per; k++) {
        const int i = grp * per + k;
        csky_ctx* c = m->ctx[i];
        if ((rc = bind(c))) return mpass(m, i, rc);
        const int nb = k < total ? (total - k + per - 1) / per : 0;
        if (nb == 0) continue;
        hipStream_t s = (i == 0) ? consumer : (dslot ? m->side[dslot - 1][i] : c->stream);
        if (i != 0 && hipStreamWaitEvent(s, m->ev_begin[slot], 0) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds_device: hipStreamWaitEvent failed");
        const csky_bands b = {8, k, per, nb};
        const bool tmg = m->timing && m->tm[0][i];
        if (tmg) { m->tm_valid[i] = 0; if (hipEventRecord(m->tm[0][i], s) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds_device: hipEventRecord failed"); }
        if (m->staged && i != 0) {
            // Fallback for nodes where fine-grained remote stores from inside the march stall: the device renders its bands into a compact local
            // buffer and one strided peer copy (a band = 8 rows, n bands apart in the frame) moves them over xGMI behind the march.
            const size_t need = (size_t)nb * 8 * tile_w;
            if (m->stage_px[dslot][i] < need) {
                if (hipStreamSynchronize(s) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds_device: hipStreamSynchronize failed");
                if (m->d_stage[dslot][i]) { (void)hipFree(m->d_stage[dslot][i]); m->d_stage[dslot][i] = nullptr; m->stage_px[dslot][i] = 0; }
                if (hipMalloc(reinterpret_cast<void**>(&m->d_stage[dslot][i]), need * 8) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds_device: hipMalloc of the staging buffer failed");
                m->stage_px[dslot][i] = need;
            }
            uint2* st = m->d_stage[dslot][i];
            if ((rc = clouds_dev(c, p, tile_w, &b, st, (size_t)tile_w * 8, s, nullptr, true, /*out_full=*/false))) return mpass(m, i, rc);
            if (tmg && hipEventRecord(m->tm[1][i], s) == hipSuccess) m->tm_valid[i] = 1;
            const size_t band_bytes = (size_t)8 * tile_w * 8;
            char* dst0 = (char*)d_out + (size_t)k * 8 * pitch;
            hipError_t e;
            if (pitch == (size_t)tile_w * 8) e = hipMemcpy2DAsync(dst0, (size_t)per * 8 * pitch, st, band_bytes, band_bytes, nb, hipMemcpyDeviceToDevice, s);
            else {
                e = hipSuccess;
                for (int bnd = 0; bnd < nb && e == hipSuccess; bnd++)
                    e = hipMemcpy2DAsync(dst0 + (size_t)bnd * per * 8 * pitch, pitch, (char*)st + (size_t)bnd * band_bytes, (size_t)tile_w * 8, (size_t)tile_w * 8, 8, hipMemcpyDeviceToDevice, s);
            }
            if (e != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds_device: peer copy failed: %s", hipGetErrorString(e));
            if (tmg && m->tm_valid[i] == 1 && hipEventRecord(m->tm[2][i], s) == hipSuccess) m->tm_valid[i] = 2;
        } else {
            if ((rc = clouds_dev(c, p, tile_w, &b, (uint2*)d_out, pitch, s, nullptr, true, /*out_full=*/true))) return mpass(m, i, rc);
            if (tmg && hipEventRecord(m->tm[1][i], s) == hipSuccess) m->tm_valid[i] = 1;
        }
        if (i != 0 && hipEventRecord(m->ev_done[slot][i], s) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds_device: hipEventRecord failed");
    }
    if ((rc = bind(c0))) return mpass(m, 0, rc);
    for (int k = 0; k < per; k++) {
        const int i = grp * per + k;
        if (i == 0 || k >= total) continue;
        if (hipStreamWaitEvent(consumer, m->ev_done[slot][i], 0) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds_device: hipStreamWaitEvent failed");
    }
    m->frame_no++;
    return CSKY_OK;
}

int csky_multi_render_clouds(csky_multi* m, const csky_cloud_params* p, int tile_w, int tile_h, uint16_t* out, size_t pitch) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_render_clouds: handle is NULL");
    if (tile_w < 1 || tile_h < 8 || (tile_h & 7)) return mfail(m, CSKY_ERR_INVALID, "csky_multi_render_clouds: tile_h must be a positive multiple of 8 (bands are 8 rows; the single-device csky_render_clouds takes ragged tiles)");
    if (out && pitch < (size_t)tile_w * 8) return mfail(m, CSKY_ERR_INVALID, "csky_multi_render_clouds: row_pitch_bytes < tile_w*8");
    csky_ctx* c0 = m->ctx[0];
    int rc; if ((rc = bind(c0))) return mpass(m, 0, rc);
    const size_t px = (size_t)tile_w * tile_h;
    if (m->frame_px < px) {
        if (hipStreamSynchronize(c0->stream) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds: hipStreamSynchronize failed");
        if (m->d_frame) { (void)hipFree(m->d_frame); m->d_frame = nullptr; m->frame_px = 0; }
        if (hipMalloc(reinterpret_cast<void**>(&m->d_frame), px * 8) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds: hipMalloc failed");
        m->frame_px = px;
    }
    if ((rc = csky_multi_render_clouds_device(m, p, tile_w, tile_h, m->d_frame, (size_t)tile_w * 8, nullptr))) return rc;
    if ((rc = bind(c0))) return mpass(m, 0, rc);
    if (out && hipMemcpy2DAsync(out, pitch, m->d_frame, (size_t)tile_w * 8, (size_t)tile_w * 8, tile_h, hipMemcpyDeviceToHost, c0->stream) != hipSuccess)
        return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds: copy to host failed");
    if (hipStreamSynchronize(c0->stream) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_render_clouds: hipStreamSynchronize failed");
    return CSKY_OK;
}

// asynchronous host form over the multi-device handle: the first context's pinned ring and streams serve as consumer streams
int csky_multi_set_host_ring(csky_multi* m, int slots) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_set_host_ring: handle is NULL");
    if (slots < 1 || slots > HOST_RING || slots > m->groups * RING || slots % m->groups) return mfail(m, CSKY_ERR_INVALID, "csky_multi_set_host_ring: 1 .. 8 frames, a multiple of the group count, at most 8 per group");
    csky_ctx* c0 = m->ctx[0];
    for (auto& hs : c0->hring) if (hs.busy) return mfail(m, CSKY_ERR_STATE, "csky_multi_set_host_ring: collect the outstanding tickets first");
    const int rc = csky_multi_set_frames_in_flight(m, slots / m->groups); if (rc) return rc;
    c0->hslots = slots;
    return CSKY_OK;
}
int csky_multi_submit_clouds(csky_multi* m, const csky_cloud_params* p, int tile_w, int tile_h, int64_t* ticket) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_submit_clouds: handle is NULL");
    if (!ticket || !p) return mfail(m, CSKY_ERR_INVALID, "csky_multi_submit_clouds: NULL argument");
    if (tile_w < 1 || tile_h < 8 || (tile_h & 7) || tile_w > 16384 || tile_h > 16384) return mfail(m, CSKY_ERR_INVALID, "csky_multi_submit_clouds: tile_h must be a positive multiple of 8, sizes <= 16384");
    csky_ctx* c0 = m->ctx[0];
    if (c0->hslots != m->groups * m->fif) { const int rc = csky_multi_set_host_ring(m, m->groups * m->fif); if (rc) return rc; }
    int rc; if ((rc = bind(c0))) return mpass(m, 0, rc);
    csky_ctx::HostSlot& hs = c0->hring[c0->next_ticket % c0->hslots];
    if (hs.busy) return mfail(m, CSKY_ERR_STATE, "csky_multi_submit_clouds: %d frames are already in flight; collect ticket %lld first", c0->hslots, hs.ticket);
    const size_t px = (size_t)tile_w * tile_h;
    if ((rc = host_slot_prepare(c0, hs, px, true))) return mpass(m, 0, rc);
    if ((rc = csky_multi_render_clouds_device(m, p, tile_w, tile_h, hs.d, (size_t)tile_w * 8, hs.s))) return rc;
    if ((rc = bind(c0))) return mpass(m, 0, rc);
    if (hipMemcpyAsync(hs.h, hs.d, px * 8, hipMemcpyDeviceToHost, hs.s) != hipSuccess || hipEventRecord(hs.done, hs.s) != hipSuccess)
        return mfail(m, CSKY_ERR_HIP, "csky_multi_submit_clouds: copy to the pinned frame failed");
    hs.busy = true; hs.ticket = c0->next_ticket; hs.w = tile_w; hs.hh = tile_h;
    *ticket = c0->next_ticket++;
    return CSKY_OK;
}
int csky_multi_collect(csky_multi* m, int64_t ticket, const uint16_t** frame, size_t* bytes) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_collect: handle is NULL");
    return mpass(m, 0, csky_collect(m->ctx[0], ticket, frame, bytes));
}

int csky_multi_sync(csky_multi* m) {
    if (!m) return mfail(nullptr, CSKY_ERR_INVALID, "csky_multi_sync: handle is NULL");
    for (size_t i = 0; i < m->ctx.size(); i++) {
        const int rc = csky_sync(m->ctx[i]); if (rc) return mpass(m, (int)i, rc);
        for (int k = 0; k < RING - 1; k++)                     // the streams of a device's 2nd..4th frame in flight
            if (hipStreamSynchronize(m->side[k][i]) != hipSuccess) return mfail(m, CSKY_ERR_HIP, "csky_multi_sync: hipStreamSynchronize failed on device index %d", (int)i);
    }
    return CSKY_OK;
}

}  // extern "C"
