/*
 * cloudsky_gdextension.c -- the thin GDExtension shim over libcloudsky's C ABI (include/cloudsky.h).
 *
 * It registers ONE class, `CloudSkyHIP` (extends RefCounted), whose methods are the dispatch sites of the reference's
 * GDScript drivers, so that cloud_sky.gd / sky_lut.gd / transmittance_lut.gd keep their resources, textures and
 * materials and only swap the RenderingDevice dispatch for a call + `rd.texture_update()` (INTEGRATION.md shows the
 * GDScript diff; the textures are created with TEXTURE_USAGE_CAN_UPDATE_BIT already: cloud_sky.gd:376, sky_lut.gd:91,
 * transmittance_lut.gd:44):
 *
 *   create(device_id: int) -> int                                   cloud_sky.gd:355-408 `_initialize_compute_code`
 *   set_noise(large, small, weather: PackedByteArray) -> int        cloud_sky.gd:298-341 `_create_noise_uniform_set` (level 0, or every mip level back to back)
 *   set_march(primary_steps, light_steps: int) -> int               clouds.glsl:228 / :186 (literals in the reference)
 *   render_transmittance(pc: PackedFloat32Array) -> PackedByteArray transmittance_lut.gd:66-77   (pc = its 4-float push constant)
 *   render_sky_lut(pc: PackedFloat32Array) -> PackedByteArray       sky_lut.gd:122-148           (pc = its 8-float push constant)
 *   render_clouds(pc: PackedFloat32Array, tile_w, tile_h: int) -> PackedByteArray
 *                                                                   cloud_sky.gd:234-248         (pc = `_fill_push_constant()`, 28 floats)
 *   get_status() -> int                                             0 or the CSKY_ERR_* code of the last call
 *   get_last_error() -> String
 *   get_last_warning() -> String                                    "" or the caveat of the last call that succeeded with one (create_multi's staged fallback, ...)
 * Throughput path (round 3): the blocking render_clouds() costs march + 16 MiB device-to-host copy per frame, one frame at a time.
 *   create_multi(device_ids: PackedInt32Array) -> int               the GPUs of the node behind this one object (csky_multi_*): every device
 *                                                                   renders its bands of each frame straight into the frame on the first one
 *   set_noise_mips(large_chain, small_chain, weather) -> int        explicit form of set_noise() with all mip levels back to back (the importer's own chains)
 *   set_frames(slots: int) -> int                                   frames kept in flight by submit/collect (1..8; default 2)
 *   submit_clouds(pc: PackedFloat32Array, tile_w, tile_h: int) -> int   enqueue march + copy into a pinned ring slot, return a ticket (>= 0) at once, < 0 = error
 *   collect(ticket: int) -> PackedByteArray                         wait for that frame, return its bytes for rd.texture_update()
 *   is_ready(ticket: int) -> int                                    1 ready, 0 in flight, < 0 error
 * cloud_sky.gd's loop already draws with the texture finished in an EARLIER pass (:137-148), so collect(ticket of the previous update) right
 * before submit_clouds(this update) drops in without changing what is on screen when.
 * Zero-copy path (round 3): the march writes into memory ANOTHER API allocated -- the engine's VkImage, exported as a POSIX fd with
 * VK_KHR_external_memory_fd (gdext/unverified/zero_copy_vulkan.c holds that half, compile-guarded: no Vulkan headers / no engine in this image) -- so
 * "returns the same TextureRD" needs no host hop at all:
 *   import_frame_fd(fd: int, layout: PackedInt32Array) -> int       layout = [allocation_bytes, offset_bytes, row_pitch_bytes, width, height]; returns a slot (0..3)
 *                                                                   or < 0; the library owns the fd on success (csky_external_frame_import_fd)
 *   render_clouds_into(slot: int, pc: PackedFloat32Array) -> int    enqueue the march of the whole width x height tile into that memory + a fence; returns at once
 *   frame_ready(slot: int) -> int                                   1 = the last march into that slot is complete (safe to sample), 0 = in flight, < 0 error
 *   release_frame(slot: int) -> int                                 waits for the slot's fence and drops the library's mapping (the exporter's memory is untouched)
 * Exercised in tests/gdext_mock_host.c with an allocation from HIP's virtual-memory API exported as a dma-buf fd standing in for the VkImage.
 *
 * The render methods return the image as tightly packed RGBA16F bytes (what texture_update takes); on error they
 * return an empty array and get_status() / get_last_error() say why.  Nothing here computes anything: every method is
 * argument marshalling around one csky_* call.
 *
 * Build: gdext/Makefile.  Against a real engine define CSKY_HAVE_GODOT_HEADERS and put Godot's own
 * gdextension_interface.h on the include path; in this repository it is compiled against gdextension_min.h and run
 * against tests/gdext_mock_host.c (Godot is on neither box).
 */
#ifdef CSKY_HAVE_GODOT_HEADERS
#include <gdextension_interface.h>
#else
#include "gdextension_min.h"
#endif
#include <string.h>
#include "../include/cloudsky.h"

/* builtin-method lookup (PackedByteArray.size / .resize) and default constructor: declared here because the minimal
 * header keeps to plain typedefs */
typedef void (*csky_builtin_method)(GDExtensionTypePtr p_base, const GDExtensionConstTypePtr *p_args, GDExtensionTypePtr r_return, int p_argument_count);
typedef csky_builtin_method (*csky_get_builtin_method)(GDExtensionVariantType p_type, GDExtensionConstStringNamePtr p_method, GDExtensionInt p_hash);
typedef void (*csky_ptr_constructor)(GDExtensionTypePtr p_base, const GDExtensionConstTypePtr *p_args);
typedef csky_ptr_constructor (*csky_get_ptr_constructor)(GDExtensionVariantType p_type, int32_t p_constructor);

/* method hashes from Godot 4.2's extension_api.json (builtin_classes / PackedByteArray, PackedFloat32Array) */
#define CSKY_HASH_PACKED_SIZE 3173160232LL
#define CSKY_HASH_PACKED_RESIZE 848867239LL

#define CSKY_OPAQUE 16 /* sizeof(PackedByteArray) == sizeof(PackedFloat32Array) == 16 on 64-bit builds */
typedef struct { uint8_t opaque[CSKY_OPAQUE]; } csky_packed;
typedef struct { uint8_t opaque[8]; } csky_name;   /* StringName */
typedef struct { uint8_t opaque[8]; } csky_string; /* String */
typedef struct { uint8_t opaque[24]; } csky_variant;

static struct {
    GDExtensionClassLibraryPtr library;
    GDExtensionInterfaceMemAlloc mem_alloc;
    GDExtensionInterfaceMemFree mem_free;
    GDExtensionInterfaceStringNameNewWithLatin1Chars string_name_new;
    GDExtensionInterfaceStringNewWithUtf8Chars string_new;
    GDExtensionInterfacePackedByteArrayOperatorIndex pba_index;
    GDExtensionInterfacePackedByteArrayOperatorIndexConst pba_index_const;
    GDExtensionInterfacePackedFloat32ArrayOperatorIndexConst pfa_index_const;
    GDExtensionInterfacePackedInt32ArrayOperatorIndexConst pia_index_const;
    GDExtensionInterfaceClassdbConstructObject construct_object;
    GDExtensionInterfaceObjectSetInstance object_set_instance;
    GDExtensionInterfaceClassdbRegisterExtensionClass2 register_class;
    GDExtensionInterfaceClassdbRegisterExtensionClassMethod register_method;
    GDExtensionInterfaceClassdbUnregisterExtensionClass unregister_class;
    GDExtensionInterfaceGetVariantToTypeConstructor to_type;
    GDExtensionInterfaceGetVariantFromTypeConstructor from_type;
    GDExtensionInterfaceVariantGetPtrDestructor get_destructor;
    csky_ptr_constructor pba_default_ctor;
    csky_builtin_method pba_size, pba_resize, pfa_size, pia_size;
    GDExtensionPtrDestructor pba_destroy, pfa_destroy, pia_destroy;
    csky_name class_name, parent_name;
} G;

#define CSKY_EXT_SLOTS 4   /* the reference keeps three cloud textures (cloud_sky.gd:368-378) */
typedef struct {
    csky_ctx *ctx;        /* the context every method talks to: the object's own, or the first device's context of `multi` */
    csky_multi *multi;    /* create_multi(): the devices of the node behind this object (owns ctx then) */
    struct { csky_external_frame *frame; void *d_ptr; int w, h; size_t pitch; } ext[CSKY_EXT_SLOTS];   /* import_frame_fd() */
    int status;
    char err[512];
} CloudSkyHIP;

/* ---- small helpers ---------------------------------------------------------------------------------------------------- */
static GDExtensionInt packed_size(csky_builtin_method size_fn, GDExtensionConstTypePtr arr) {
    GDExtensionInt n = 0;
    size_fn((GDExtensionTypePtr)arr, NULL, &n, 0);
    return n;
}
/* *r_out = PackedByteArray(); r_out.resize(bytes); returns its writable data pointer, NULL if bytes == 0 OR the resize failed (the caller
 * must tell the two apart).  A ptrcall return slot holds an INITIALISED value of the return type (the engine's own PtrToArg<Packed*Array>::encode
 * ASSIGNS into it, core/variant/method_ptrcall.h; GDScript and godot-cpp pass a default-constructed value), so this is an assignment: the old
 * value is destroyed first, otherwise a caller that passes a non-empty array would leak it. */
static uint8_t *packed_byte_array_new(csky_packed *r_out, GDExtensionInt bytes) {
    GDExtensionInt arg = bytes, ret = 0;
    GDExtensionConstTypePtr args[1];
    G.pba_destroy(r_out);
    G.pba_default_ctor(r_out, NULL);
    if (bytes <= 0) return NULL;
    args[0] = &arg;
    G.pba_resize(r_out, args, &ret, 1);
    if (ret != 0 || packed_size(G.pba_size, r_out) != bytes) return NULL;       /* resize() returns an Error; OK == 0 */
    return G.pba_index(r_out, 0);
}
/* LUT / tile sizes come from push-constant FLOATS the script controls: NaN, negative or huge values must never reach an integer cast or
 * an allocation.  Returns 1 and the sizes when both are finite integers in [1, 8192] (the library's own limit). */
static int lut_size_ok(const float ts[2], GDExtensionInt *w, GDExtensionInt *h) {
    if (!(ts[0] >= 1.0f && ts[0] <= 8192.0f && ts[1] >= 1.0f && ts[1] <= 8192.0f)) return 0;   /* false for NaN too */
    *w = (GDExtensionInt)ts[0]; *h = (GDExtensionInt)ts[1];
    return 1;
}
static int fail(CloudSkyHIP *self, int code, const char *text) {
    self->status = code;
    strncpy(self->err, text ? text : "", sizeof self->err - 1);
    self->err[sizeof self->err - 1] = 0;
    return code;
}
static int pass(CloudSkyHIP *self, int rc) { /* record the library's own error text */
    if (rc != CSKY_OK) return fail(self, rc, csky_last_error(self->ctx));
    self->status = CSKY_OK; self->err[0] = 0;
    return rc;
}
static const float *float_args(CloudSkyHIP *self, GDExtensionConstTypePtr arr, GDExtensionInt want, const char *what) {
    if (packed_size(G.pfa_size, arr) != want) { fail(self, CSKY_ERR_INVALID, what); return NULL; }
    return G.pfa_index_const(arr, 0);
}

static void release(CloudSkyHIP *self) {
    int i;
    for (i = 0; i < CSKY_EXT_SLOTS; i++)
        if (self->ext[i].frame) { csky_external_frame_release(self->ext[i].frame); self->ext[i].frame = NULL; }    /* waits for its fence */
    if (self->multi) { csky_multi_destroy(self->multi); self->multi = NULL; self->ctx = NULL; }
    if (self->ctx) { csky_destroy(self->ctx); self->ctx = NULL; }
}
static int mpass(CloudSkyHIP *self, int rc) { /* the same for calls through the multi-device handle */
    if (rc != CSKY_OK) return fail(self, rc, csky_multi_last_error(self->multi));
    self->status = CSKY_OK; self->err[0] = 0;
    return rc;
}

/* ---- the methods (ptrcall form: p_args[i] points to the native value) --------------------------------------------------- */
static void m_create(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    int rc;
    release(self);
    rc = csky_create(&self->ctx, (int)*(const GDExtensionInt *)a[0]);
    if (rc != CSKY_OK) fail(self, rc, csky_last_error(NULL)); else pass(self, rc);
    *(GDExtensionInt *)r = rc;
}
static void m_create_multi(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    int ids[64], i, rc;
    const GDExtensionInt n = packed_size(G.pia_size, a[0]);
    release(self);
    if (n < 1 || n > 64) { *(GDExtensionInt *)r = fail(self, CSKY_ERR_INVALID, "create_multi: 1 .. 64 device ids"); return; }
    for (i = 0; i < (int)n; i++) ids[i] = (int)*G.pia_index_const(a[0], i);
    rc = csky_multi_create(&self->multi, ids, (int)n);
    if (rc != CSKY_OK) { fail(self, rc, csky_multi_last_error(NULL)); *(GDExtensionInt *)r = rc; return; }
    self->ctx = csky_multi_ctx(self->multi, 0);      /* LUT reads, status texts; owned by `multi` */
    *(GDExtensionInt *)r = pass(self, CSKY_OK);
}
static void m_set_noise(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    int rc;
    if (!self->ctx) rc = fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called");
    else if (packed_size(G.pba_size, a[2]) != 512 * 512 * 3)
        rc = fail(self, CSKY_ERR_INVALID, "set_noise: the weather map must be 512^2 RGB8");
    else if (packed_size(G.pba_size, a[0]) == 128 * 128 * 128 * 4 && packed_size(G.pba_size, a[1]) == 32 * 32 * 32 * 3)
        rc = self->multi ? mpass(self, csky_multi_set_noise(self->multi, G.pba_index_const(a[0], 0), G.pba_index_const(a[1], 0), G.pba_index_const(a[2], 0)))
                         : pass(self, csky_set_noise(self->ctx, G.pba_index_const(a[0], 0), G.pba_index_const(a[1], 0), G.pba_index_const(a[2], 0)));
    /* all mip levels back to back (what Texture3D.get_data() holds after Image.decompress()): the importer's own chains are bound as they are */
    else if ((size_t)packed_size(G.pba_size, a[0]) == csky_mip_offset(128, 8, 4) && (size_t)packed_size(G.pba_size, a[1]) == csky_mip_offset(32, 6, 3))
        rc = self->multi ? mpass(self, csky_multi_set_noise_mips(self->multi, G.pba_index_const(a[0], 0), G.pba_index_const(a[1], 0), G.pba_index_const(a[2], 0)))
                         : pass(self, csky_set_noise_mips(self->ctx, G.pba_index_const(a[0], 0), G.pba_index_const(a[1], 0), G.pba_index_const(a[2], 0)));
    else
        rc = fail(self, CSKY_ERR_INVALID, "set_noise: expected 128^3 RGBA8 and 32^3 RGB8 byte arrays, level 0 only or all mip levels back to back");
    *(GDExtensionInt *)r = rc;
}
/* the explicit form: ONLY the full chains are accepted (a host that decompressed the importer's .ctex3d files, perlworlnoise.tga.import:24) */
static void m_set_noise_mips(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    int rc;
    if (!self->ctx) rc = fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called");
    else if (packed_size(G.pba_size, a[2]) != 512 * 512 * 3 || (size_t)packed_size(G.pba_size, a[0]) != csky_mip_offset(128, 8, 4) ||
             (size_t)packed_size(G.pba_size, a[1]) != csky_mip_offset(32, 6, 3))
        rc = fail(self, CSKY_ERR_INVALID, "set_noise_mips: expected the 8-level 128^3 RGBA8 chain, the 6-level 32^3 RGB8 chain and the 512^2 RGB8 weather map");
    else
        rc = self->multi ? mpass(self, csky_multi_set_noise_mips(self->multi, G.pba_index_const(a[0], 0), G.pba_index_const(a[1], 0), G.pba_index_const(a[2], 0)))
                         : pass(self, csky_set_noise_mips(self->ctx, G.pba_index_const(a[0], 0), G.pba_index_const(a[1], 0), G.pba_index_const(a[2], 0)));
    *(GDExtensionInt *)r = rc;
}
static void m_set_march(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    const int ps = (int)*(const GDExtensionInt *)a[0], ls = (int)*(const GDExtensionInt *)a[1];
    *(GDExtensionInt *)r = !self->ctx ? fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called")
                         : self->multi ? mpass(self, csky_multi_set_march(self->multi, ps, ls)) : pass(self, csky_set_march(self->ctx, ps, ls));
}
static void m_render_transmittance(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    csky_transmittance_params p;
    const float *pc;
    GDExtensionInt w, h;
    uint8_t *dst;
    if (!self->ctx) { fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called"); packed_byte_array_new((csky_packed *)r, 0); return; }
    pc = float_args(self, a[0], 4, "render_transmittance: push constant must be 4 floats (transmittance-lut.glsl:12-15)");
    if (!pc) { packed_byte_array_new((csky_packed *)r, 0); return; }
    memcpy(&p, pc, sizeof p);
    if (!lut_size_ok(p.texture_size, &w, &h)) { fail(self, CSKY_ERR_INVALID, "render_transmittance: texture_size must be finite and in [1, 8192]"); packed_byte_array_new((csky_packed *)r, 0); return; }
    dst = packed_byte_array_new((csky_packed *)r, w * h * 8);
    if (!dst) { fail(self, CSKY_ERR_INVALID, "render_transmittance: could not allocate the result array"); packed_byte_array_new((csky_packed *)r, 0); return; }
    if (pass(self, csky_render_transmittance(self->ctx, &p, (uint16_t *)dst)) != CSKY_OK) { packed_byte_array_new((csky_packed *)r, 0); }
}
static void m_render_sky_lut(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    csky_sky_params p;
    const float *pc;
    GDExtensionInt w, h;
    uint8_t *dst;
    if (!self->ctx) { fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called"); packed_byte_array_new((csky_packed *)r, 0); return; }
    pc = float_args(self, a[0], 8, "render_sky_lut: push constant must be 8 floats (sky-lut.glsl:12-18)");
    if (!pc) { packed_byte_array_new((csky_packed *)r, 0); return; }
    memcpy(&p, pc, sizeof p);
    if (!lut_size_ok(p.texture_size, &w, &h)) { fail(self, CSKY_ERR_INVALID, "render_sky_lut: texture_size must be finite and in [1, 8192]"); packed_byte_array_new((csky_packed *)r, 0); return; }
    dst = packed_byte_array_new((csky_packed *)r, w * h * 8);
    if (!dst) { fail(self, CSKY_ERR_INVALID, "render_sky_lut: could not allocate the result array"); packed_byte_array_new((csky_packed *)r, 0); return; }
    /* with several devices every one of them needs the LUT for its bands; the bytes come from the first */
    if (self->multi && mpass(self, csky_multi_render_sky_lut(self->multi, &p)) != CSKY_OK) { packed_byte_array_new((csky_packed *)r, 0); return; }
    if (pass(self, self->multi ? csky_read_sky_lut(self->ctx, (uint16_t *)dst, NULL, NULL) : csky_render_sky_lut(self->ctx, &p, (uint16_t *)dst)) != CSKY_OK) { packed_byte_array_new((csky_packed *)r, 0); }
}
static void m_render_clouds(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    csky_cloud_params p;
    const float *pc;
    const GDExtensionInt w = *(const GDExtensionInt *)a[1], h = *(const GDExtensionInt *)a[2];
    uint8_t *dst;
    if (!self->ctx) { fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called"); packed_byte_array_new((csky_packed *)r, 0); return; }
    pc = float_args(self, a[0], 28, "render_clouds: push constant must be the 28 floats of _fill_push_constant() (clouds.glsl:18-40)");
    if (!pc || w < 1 || h < 1 || w > 16384 || h > 16384) {
        if (pc) fail(self, CSKY_ERR_INVALID, "render_clouds: tile size out of range");
        packed_byte_array_new((csky_packed *)r, 0);
        return;
    }
    memcpy(&p, pc, sizeof p);
    dst = packed_byte_array_new((csky_packed *)r, w * h * 8);
    if (!dst) { fail(self, CSKY_ERR_INVALID, "render_clouds: could not allocate the result array"); packed_byte_array_new((csky_packed *)r, 0); return; }
    if ((self->multi ? mpass(self, csky_multi_render_clouds(self->multi, &p, (int)w, (int)h, (uint16_t *)dst, (size_t)w * 8))
                     : pass(self, csky_render_clouds(self->ctx, &p, (int)w, (int)h, (uint16_t *)dst, (size_t)w * 8))) != CSKY_OK) { packed_byte_array_new((csky_packed *)r, 0); }
}
/* ---- throughput path: frames in flight over the library's pinned ring (csky_submit_clouds / csky_collect) ------------------------------- */
static void m_set_frames(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    const int n = (int)*(const GDExtensionInt *)a[0];
    *(GDExtensionInt *)r = !self->ctx ? fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called")
                         : self->multi ? mpass(self, csky_multi_set_host_ring(self->multi, n)) : pass(self, csky_set_host_ring(self->ctx, n));
}
static void m_submit_clouds(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    csky_cloud_params p;
    const float *pc;
    const GDExtensionInt w = *(const GDExtensionInt *)a[1], h = *(const GDExtensionInt *)a[2];
    int64_t ticket = -1;
    int rc;
    if (!self->ctx) { *(GDExtensionInt *)r = fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called"); return; }
    pc = float_args(self, a[0], 28, "submit_clouds: push constant must be the 28 floats of _fill_push_constant() (clouds.glsl:18-40)");
    if (!pc) { *(GDExtensionInt *)r = CSKY_ERR_INVALID; return; }
    if (w < 1 || h < 1 || w > 16384 || h > 16384) { *(GDExtensionInt *)r = fail(self, CSKY_ERR_INVALID, "submit_clouds: tile size out of range"); return; }
    memcpy(&p, pc, sizeof p);
    rc = self->multi ? mpass(self, csky_multi_submit_clouds(self->multi, &p, (int)w, (int)h, &ticket)) : pass(self, csky_submit_clouds(self->ctx, &p, (int)w, (int)h, &ticket));
    *(GDExtensionInt *)r = rc == CSKY_OK ? (GDExtensionInt)ticket : (GDExtensionInt)rc;
}
static void m_collect(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    const uint16_t *frame = NULL;
    size_t bytes = 0;
    uint8_t *dst;
    if (!self->ctx) { fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called"); packed_byte_array_new((csky_packed *)r, 0); return; }
    if (pass(self, csky_collect(self->ctx, (int64_t)*(const GDExtensionInt *)a[0], &frame, &bytes)) != CSKY_OK) { packed_byte_array_new((csky_packed *)r, 0); return; }
    dst = packed_byte_array_new((csky_packed *)r, (GDExtensionInt)bytes);
    if (!dst) { fail(self, CSKY_ERR_INVALID, "collect: could not allocate the result array"); packed_byte_array_new((csky_packed *)r, 0); return; }
    memcpy(dst, frame, bytes);           /* pinned ring slot -> the array rd.texture_update() takes; the slot is free for the next submit */
}
static void m_is_ready(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    int rc;
    if (!self->ctx) { *(GDExtensionInt *)r = fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called"); return; }
    rc = csky_poll(self->ctx, (int64_t)*(const GDExtensionInt *)a[0]);
    if (rc < 0) fail(self, rc, csky_last_error(self->ctx)); else { self->status = CSKY_OK; self->err[0] = 0; }
    *(GDExtensionInt *)r = rc;
}
/* ---- zero-copy path: frames that live in memory another API allocated ------------------------------------------------------- */
static void m_import_frame_fd(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    const GDExtensionInt fd = *(const GDExtensionInt *)a[0];
    const int32_t *l;
    int slot, rc;
    if (!self->ctx) { *(GDExtensionInt *)r = fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called"); return; }
    if (packed_size(G.pia_size, a[1]) != 5) { *(GDExtensionInt *)r = fail(self, CSKY_ERR_INVALID, "import_frame_fd: layout = [allocation_bytes, offset_bytes, row_pitch_bytes, width, height]"); return; }
    l = G.pia_index_const(a[1], 0);
    if (fd < 0 || fd > 0x7fffffff || l[0] < 1 || l[1] < 0 || l[3] < 1 || l[4] < 1 || l[3] > 16384 || l[4] > 16384 || (int64_t)l[2] < (int64_t)l[3] * 8 ||
        (int64_t)l[1] + (int64_t)l[2] * l[4] > (int64_t)l[0]) { *(GDExtensionInt *)r = fail(self, CSKY_ERR_INVALID, "import_frame_fd: the frame does not fit the allocation"); return; }
    for (slot = 0; slot < CSKY_EXT_SLOTS && self->ext[slot].frame; slot++) {}
    if (slot == CSKY_EXT_SLOTS) { *(GDExtensionInt *)r = fail(self, CSKY_ERR_STATE, "import_frame_fd: all 4 slots are in use (release_frame)"); return; }
    rc = pass(self, csky_external_frame_import_fd(self->ctx, (int)fd, (size_t)l[0], (size_t)l[1], (size_t)l[2] * (size_t)l[4], &self->ext[slot].frame, &self->ext[slot].d_ptr));
    if (rc != CSKY_OK) { self->ext[slot].frame = NULL; *(GDExtensionInt *)r = rc; return; }
    self->ext[slot].pitch = (size_t)l[2]; self->ext[slot].w = l[3]; self->ext[slot].h = l[4];
    *(GDExtensionInt *)r = slot;
}
static int ext_slot(CloudSkyHIP *self, GDExtensionInt slot) {
    if (!self->ctx) return fail(self, CSKY_ERR_STATE, "CloudSkyHIP: create() has not been called");
    if (slot < 0 || slot >= CSKY_EXT_SLOTS || !self->ext[slot].frame) return fail(self, CSKY_ERR_INVALID, "no imported frame in that slot (import_frame_fd)");
    return CSKY_OK;
}
static void m_render_clouds_into(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    const GDExtensionInt slot = *(const GDExtensionInt *)a[0];
    csky_cloud_params p;
    const float *pc;
    int rc;
    if ((rc = ext_slot(self, slot)) != CSKY_OK) { *(GDExtensionInt *)r = rc; return; }
    pc = float_args(self, a[1], 28, "render_clouds_into: push constant must be the 28 floats of _fill_push_constant() (clouds.glsl:18-40)");
    if (!pc) { *(GDExtensionInt *)r = CSKY_ERR_INVALID; return; }
    memcpy(&p, pc, sizeof p);
    if (self->multi) {
        /* An imported allocation is mapped on the first device only (hipExternalMemoryGetMappedBuffer); peer stores of the other devices into
         * it are not guaranteed and have never run on a multi-GPU box, and the fence below is recorded on the first context's stream while a
         * multi handle rotates streams: refused until it has been exercised.  create() + render_clouds_into is the tested form. */
        *(GDExtensionInt *)r = fail(self, CSKY_ERR_STATE, "render_clouds_into: not available on a create_multi() object (imported memory is mapped on one device only)");
        return;
    } else {
        const csky_bands whole = {self->ext[slot].h, 0, 1, 1};
        rc = pass(self, csky_render_clouds_device(self->ctx, &p, self->ext[slot].w, &whole, self->ext[slot].d_ptr, self->ext[slot].pitch, NULL));
    }
    if (rc == CSKY_OK) rc = pass(self, csky_external_frame_fence(self->ctx, self->ext[slot].frame, NULL));   /* NULL = the same stream the march was ordered on */
    *(GDExtensionInt *)r = rc;
}
static void m_frame_ready(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    const GDExtensionInt slot = *(const GDExtensionInt *)a[0];
    int rc;
    if ((rc = ext_slot(self, slot)) != CSKY_OK) { *(GDExtensionInt *)r = rc; return; }
    rc = csky_external_frame_ready(self->ctx, self->ext[slot].frame);
    if (rc < 0) fail(self, rc, csky_last_error(self->ctx)); else { self->status = CSKY_OK; self->err[0] = 0; }
    *(GDExtensionInt *)r = rc;
}
static void m_release_frame(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst; (void)ud;
    const GDExtensionInt slot = *(const GDExtensionInt *)a[0];
    int rc;
    if ((rc = ext_slot(self, slot)) != CSKY_OK) { *(GDExtensionInt *)r = rc; return; }
    csky_external_frame_release(self->ext[slot].frame);
    self->ext[slot].frame = NULL; self->ext[slot].d_ptr = NULL;
    self->status = CSKY_OK; self->err[0] = 0;
    *(GDExtensionInt *)r = CSKY_OK;
}
static void m_get_status(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    (void)ud; (void)a;
    *(GDExtensionInt *)r = ((CloudSkyHIP *)inst)->status;
}
static void m_get_last_error(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    (void)ud; (void)a;
    G.string_new(r, ((CloudSkyHIP *)inst)->err);
}
/* "" or the caveat of the last call that succeeded with one: csky_multi_create's fallback (a device without peer access: staged copies), else the context's
 * csky_last_warning (textures that do not fit fp16 cells, hardware queues) */
static void m_get_last_warning(void *ud, GDExtensionClassInstancePtr inst, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r) {
    CloudSkyHIP *self = (CloudSkyHIP *)inst;
    const char *w = "";
    (void)ud; (void)a;
    if (self->multi && csky_multi_last_warning(self->multi)[0]) w = csky_multi_last_warning(self->multi);
    else if (self->ctx) w = csky_last_warning(self->ctx);
    G.string_new(r, w);
}

/* ---- method table + the generic Variant-call trampoline ----------------------------------------------------------------- */
typedef struct {
    const char *name;
    GDExtensionClassMethodPtrCall ptrcall;
    int argc;
    GDExtensionVariantType ret;
    GDExtensionVariantType args[3];
    const char *arg_names[3];
} csky_method;

static const csky_method METHODS[] = {
    {"create", m_create, 1, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_INT}, {"device_id"}},
    {"set_noise", m_set_noise, 3, GDEXTENSION_VARIANT_TYPE_INT,
     {GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY}, {"large_rgba8", "small_rgb8", "weather_rgb8"}},
    {"set_march", m_set_march, 2, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_INT, GDEXTENSION_VARIANT_TYPE_INT}, {"primary_steps", "light_steps"}},
    {"render_transmittance", m_render_transmittance, 1, GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, {GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY}, {"push_constant"}},
    {"render_sky_lut", m_render_sky_lut, 1, GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, {GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY}, {"push_constant"}},
    {"render_clouds", m_render_clouds, 3, GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY,
     {GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY, GDEXTENSION_VARIANT_TYPE_INT, GDEXTENSION_VARIANT_TYPE_INT}, {"push_constant", "tile_w", "tile_h"}},
    {"create_multi", m_create_multi, 1, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_PACKED_INT32_ARRAY}, {"device_ids"}},
    {"set_noise_mips", m_set_noise_mips, 3, GDEXTENSION_VARIANT_TYPE_INT,
     {GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY}, {"large_chain_rgba8", "small_chain_rgb8", "weather_rgb8"}},
    {"set_frames", m_set_frames, 1, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_INT}, {"slots"}},
    {"submit_clouds", m_submit_clouds, 3, GDEXTENSION_VARIANT_TYPE_INT,
     {GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY, GDEXTENSION_VARIANT_TYPE_INT, GDEXTENSION_VARIANT_TYPE_INT}, {"push_constant", "tile_w", "tile_h"}},
    {"collect", m_collect, 1, GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, {GDEXTENSION_VARIANT_TYPE_INT}, {"ticket"}},
    {"is_ready", m_is_ready, 1, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_INT}, {"ticket"}},
    {"import_frame_fd", m_import_frame_fd, 2, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_INT, GDEXTENSION_VARIANT_TYPE_PACKED_INT32_ARRAY}, {"fd", "layout"}},
    {"render_clouds_into", m_render_clouds_into, 2, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_INT, GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY}, {"slot", "push_constant"}},
    {"frame_ready", m_frame_ready, 1, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_INT}, {"slot"}},
    {"release_frame", m_release_frame, 1, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_INT}, {"slot"}},
    {"get_status", m_get_status, 0, GDEXTENSION_VARIANT_TYPE_INT, {GDEXTENSION_VARIANT_TYPE_NIL}, {0}},
    {"get_last_error", m_get_last_error, 0, GDEXTENSION_VARIANT_TYPE_STRING, {GDEXTENSION_VARIANT_TYPE_NIL}, {0}},
    {"get_last_warning", m_get_last_warning, 0, GDEXTENSION_VARIANT_TYPE_STRING, {GDEXTENSION_VARIANT_TYPE_NIL}, {0}},
};
#define N_METHODS ((int)(sizeof METHODS / sizeof METHODS[0]))

/* Variant call: unpack every argument to its native type, run the ptrcall body, pack the result (what godot-cpp's generated
 * bindings do).  method_userdata = the table entry. */
static void call_trampoline(void *method_userdata, GDExtensionClassInstancePtr inst, const GDExtensionConstVariantPtr *p_args, GDExtensionInt argc,
                            GDExtensionVariantPtr r_return, GDExtensionCallError *r_error) {
    const csky_method *m = (const csky_method *)method_userdata;
    csky_packed native[3];                       /* large enough for every argument type used (int64 or a 16-byte packed array) */
    GDExtensionConstTypePtr argp[3];
    csky_packed ret;                             /* int64, String (8 bytes) or PackedByteArray (16 bytes) */
    int i;
    if (argc != m->argc) {
        r_error->error = argc < m->argc ? GDEXTENSION_CALL_ERROR_TOO_FEW_ARGUMENTS : GDEXTENSION_CALL_ERROR_TOO_MANY_ARGUMENTS;
        r_error->argument = m->argc; r_error->expected = m->argc;
        return;
    }
    for (i = 0; i < m->argc; i++) {
        memset(&native[i], 0, sizeof native[i]);
        G.to_type(m->args[i])(&native[i], (GDExtensionVariantPtr)p_args[i]);
        argp[i] = &native[i];
    }
    memset(&ret, 0, sizeof ret);
    m->ptrcall(NULL, inst, argp, &ret);
    G.from_type(m->ret)(r_return, &ret);
    for (i = 0; i < m->argc; i++) {
        if (m->args[i] == GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY) G.pba_destroy(&native[i]);
        else if (m->args[i] == GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY) G.pfa_destroy(&native[i]);
        else if (m->args[i] == GDEXTENSION_VARIANT_TYPE_PACKED_INT32_ARRAY) G.pia_destroy(&native[i]);
    }
    if (m->ret == GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY) G.pba_destroy(&ret);
    else if (m->ret == GDEXTENSION_VARIANT_TYPE_STRING) { GDExtensionPtrDestructor d = G.get_destructor(GDEXTENSION_VARIANT_TYPE_STRING); if (d) d(&ret); }
    r_error->error = GDEXTENSION_CALL_OK;
}

/* ---- class plumbing ---------------------------------------------------------------------------------------------------- */
static GDExtensionObjectPtr create_instance(void *class_userdata) {
    GDExtensionObjectPtr obj = G.construct_object(&G.parent_name);
    CloudSkyHIP *self = (CloudSkyHIP *)G.mem_alloc(sizeof(CloudSkyHIP));
    (void)class_userdata;
    memset(self, 0, sizeof *self);
    G.object_set_instance(obj, &G.class_name, self);
    return obj;
}
static void free_instance(void *class_userdata, GDExtensionClassInstancePtr inst) {   /* cloud_sky.gd:193-212 cleanup / NOTIFICATION_PREDELETE */
    CloudSkyHIP *self = (CloudSkyHIP *)inst;
    (void)class_userdata;
    if (!self) return;
    release(self);
    G.mem_free(self);
}

static void initialize_level(void *userdata, GDExtensionInitializationLevel level) {
    GDExtensionClassCreationInfo2 ci;
    int i, k;
    (void)userdata;
    if (level != GDEXTENSION_INITIALIZATION_SCENE) return;
    G.string_name_new(&G.class_name, "CloudSkyHIP", 1);
    G.string_name_new(&G.parent_name, "RefCounted", 1);
    memset(&ci, 0, sizeof ci);
    ci.is_exposed = 1;
    ci.create_instance_func = create_instance;
    ci.free_instance_func = free_instance;
    G.register_class(G.library, &G.class_name, &G.parent_name, &ci);
    for (i = 0; i < N_METHODS; i++) {
        const csky_method *m = &METHODS[i];
        csky_name mname, anames[3], empty;
        GDExtensionPropertyInfo ret_info, arg_info[3];
        GDExtensionClassMethodArgumentMetadata meta[3] = {GDEXTENSION_METHOD_ARGUMENT_METADATA_NONE, GDEXTENSION_METHOD_ARGUMENT_METADATA_NONE, GDEXTENSION_METHOD_ARGUMENT_METADATA_NONE};
        GDExtensionClassMethodInfo mi;
        csky_string no_hint;
        G.string_name_new(&mname, m->name, 1);
        G.string_name_new(&empty, "", 1);
        G.string_new(&no_hint, "");
        memset(&ret_info, 0, sizeof ret_info);
        ret_info.type = m->ret; ret_info.name = &empty; ret_info.class_name = &empty; ret_info.hint_string = &no_hint; ret_info.usage = 6; /* PROPERTY_USAGE_DEFAULT */
        for (k = 0; k < m->argc; k++) {
            G.string_name_new(&anames[k], m->arg_names[k], 1);
            memset(&arg_info[k], 0, sizeof arg_info[k]);
            arg_info[k].type = m->args[k]; arg_info[k].name = &anames[k]; arg_info[k].class_name = &empty; arg_info[k].hint_string = &no_hint; arg_info[k].usage = 6;
        }
        memset(&mi, 0, sizeof mi);
        mi.name = &mname;
        mi.method_userdata = (void *)m;
        mi.call_func = call_trampoline;
        mi.ptrcall_func = m->ptrcall;
        mi.method_flags = GDEXTENSION_METHOD_FLAGS_DEFAULT;
        mi.has_return_value = 1;
        mi.return_value_info = &ret_info;
        mi.argument_count = (uint32_t)m->argc;
        mi.arguments_info = arg_info;
        mi.arguments_metadata = meta;
        G.register_method(G.library, &G.class_name, &mi);
    }
}
static void deinitialize_level(void *userdata, GDExtensionInitializationLevel level) {
    (void)userdata;
    if (level == GDEXTENSION_INITIALIZATION_SCENE && G.unregister_class) G.unregister_class(G.library, &G.class_name);
}

#define LOAD(field, type, name) do { G.field = (type)get_proc(name); if (!G.field) return 0; } while (0)

/* entry_symbol of gdext/cloudsky.gdextension */
GDExtensionBool csky_gdextension_init(GDExtensionInterfaceGetProcAddress get_proc, GDExtensionClassLibraryPtr library, GDExtensionInitialization *r_init) {
    csky_get_builtin_method get_builtin;
    csky_get_ptr_constructor get_ctor;
    csky_name n_size, n_resize;
    if (!get_proc || !r_init) return 0;
    memset(&G, 0, sizeof G);
    G.library = library;
    LOAD(mem_alloc, GDExtensionInterfaceMemAlloc, "mem_alloc");
    LOAD(mem_free, GDExtensionInterfaceMemFree, "mem_free");
    LOAD(string_name_new, GDExtensionInterfaceStringNameNewWithLatin1Chars, "string_name_new_with_latin1_chars");
    LOAD(string_new, GDExtensionInterfaceStringNewWithUtf8Chars, "string_new_with_utf8_chars");
    LOAD(pba_index, GDExtensionInterfacePackedByteArrayOperatorIndex, "packed_byte_array_operator_index");
    LOAD(pba_index_const, GDExtensionInterfacePackedByteArrayOperatorIndexConst, "packed_byte_array_operator_index_const");
    LOAD(pfa_index_const, GDExtensionInterfacePackedFloat32ArrayOperatorIndexConst, "packed_float32_array_operator_index_const");
    LOAD(pia_index_const, GDExtensionInterfacePackedInt32ArrayOperatorIndexConst, "packed_int32_array_operator_index_const");
    LOAD(construct_object, GDExtensionInterfaceClassdbConstructObject, "classdb_construct_object");
    LOAD(object_set_instance, GDExtensionInterfaceObjectSetInstance, "object_set_instance");
    LOAD(register_class, GDExtensionInterfaceClassdbRegisterExtensionClass2, "classdb_register_extension_class2");
    LOAD(register_method, GDExtensionInterfaceClassdbRegisterExtensionClassMethod, "classdb_register_extension_class_method");
    LOAD(unregister_class, GDExtensionInterfaceClassdbUnregisterExtensionClass, "classdb_unregister_extension_class");
    LOAD(to_type, GDExtensionInterfaceGetVariantToTypeConstructor, "get_variant_to_type_constructor");
    LOAD(from_type, GDExtensionInterfaceGetVariantFromTypeConstructor, "get_variant_from_type_constructor");
    LOAD(get_destructor, GDExtensionInterfaceVariantGetPtrDestructor, "variant_get_ptr_destructor");
    get_builtin = (csky_get_builtin_method)get_proc("variant_get_ptr_builtin_method");
    get_ctor = (csky_get_ptr_constructor)get_proc("variant_get_ptr_constructor");
    if (!get_builtin || !get_ctor) return 0;
    G.string_name_new(&n_size, "size", 1);
    G.string_name_new(&n_resize, "resize", 1);
    G.pba_size = get_builtin(GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, &n_size, CSKY_HASH_PACKED_SIZE);
    G.pfa_size = get_builtin(GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY, &n_size, CSKY_HASH_PACKED_SIZE);
    G.pia_size = get_builtin(GDEXTENSION_VARIANT_TYPE_PACKED_INT32_ARRAY, &n_size, CSKY_HASH_PACKED_SIZE);
    G.pba_resize = get_builtin(GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, &n_resize, CSKY_HASH_PACKED_RESIZE);
    G.pba_default_ctor = get_ctor(GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY, 0);
    G.pba_destroy = G.get_destructor(GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY);
    G.pfa_destroy = G.get_destructor(GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY);
    G.pia_destroy = G.get_destructor(GDEXTENSION_VARIANT_TYPE_PACKED_INT32_ARRAY);
    if (!G.pba_size || !G.pfa_size || !G.pia_size || !G.pba_resize || !G.pba_default_ctor || !G.pba_destroy || !G.pfa_destroy || !G.pia_destroy) return 0;
    r_init->minimum_initialization_level = GDEXTENSION_INITIALIZATION_SCENE;
    r_init->userdata = NULL;
    r_init->initialize = initialize_level;
    r_init->deinitialize = deinitialize_level;
    return 1;
}
