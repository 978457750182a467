/* A mock GDExtension host: just enough of Godot's extension interface to load gdext/libcloudsky_gdext.so the way the engine
 * would (entry symbol -> initialize(SCENE) -> class + method registration), instantiate `CloudSkyHIP` and call its methods
 * through BOTH binding forms the engine uses (ptrcall and Variant call).  TEST INFRASTRUCTURE: Godot is on neither box.
 *   gdext_mock_host <libcloudsky_gdext.so>                          registration + error paths (no GPU needed)
 *   gdext_mock_host <libcloudsky_gdext.so> <asset_dir> <fixture>    + create / set_noise / LUTs / a 64x32 cloud frame vs the fixture
 * Mock object model: StringName = const char*, String = char*, Packed*Array = {data, size}, Variant = {type, payload}. */
#include <dlfcn.h>
#include <unistd.h>
#include "../gdext/gdextension_min.h"
#include "c_test_util.h"

typedef struct { uint8_t *data; int64_t size; } MockPacked;                 /* 16 bytes, like the engine's */
typedef struct { int32_t type; int32_t pad; union { int64_t i; MockPacked arr; char *s; } u; } MockVariant;   /* 24 bytes */

typedef void (*builtin_method)(GDExtensionTypePtr, const GDExtensionConstTypePtr *, GDExtensionTypePtr, int);
typedef void (*ptr_constructor)(GDExtensionTypePtr, const GDExtensionConstTypePtr *);

static struct {
    char class_name[64], parent[64];
    GDExtensionClassCreationInfo2 ci;
    struct { char name[64]; GDExtensionClassMethodInfo mi; } methods[32];
    int n_methods;
    void *instance;
} R;

/* ---- interface functions ------------------------------------------------------------------------------------------------- */
static void *i_mem_alloc(size_t n) { return malloc(n); }
static void i_mem_free(void *p) { free(p); }
static void i_string_name_new(GDExtensionStringNamePtr dst, const char *s, GDExtensionBool is_static) { (void)is_static; *(const char **)dst = s; }
static void i_string_new(GDExtensionStringPtr dst, const char *s) { *(char **)dst = strdup(s); }
static uint8_t *i_pba_index(GDExtensionTypePtr self, GDExtensionInt i) { return ((MockPacked *)self)->data + i; }
static const uint8_t *i_pba_index_const(GDExtensionConstTypePtr self, GDExtensionInt i) { return ((const MockPacked *)self)->data + i; }
static const float *i_pfa_index_const(GDExtensionConstTypePtr self, GDExtensionInt i) { return (const float *)((const MockPacked *)self)->data + i; }
static const int32_t *i_pia_index_const(GDExtensionConstTypePtr self, GDExtensionInt i) { return (const int32_t *)((const MockPacked *)self)->data + i; }
static GDExtensionObjectPtr i_construct_object(GDExtensionConstStringNamePtr name) { (void)name; return malloc(8); }
static void i_object_set_instance(GDExtensionObjectPtr o, GDExtensionConstStringNamePtr cls, GDExtensionClassInstancePtr inst) { (void)o; (void)cls; R.instance = inst; }
static void i_register_class(GDExtensionClassLibraryPtr lib, GDExtensionConstStringNamePtr name, GDExtensionConstStringNamePtr parent, const GDExtensionClassCreationInfo2 *ci) {
    (void)lib;
    snprintf(R.class_name, sizeof R.class_name, "%s", *(const char *const *)name);
    snprintf(R.parent, sizeof R.parent, "%s", *(const char *const *)parent);
    R.ci = *ci;
}
static void i_register_method(GDExtensionClassLibraryPtr lib, GDExtensionConstStringNamePtr cls, const GDExtensionClassMethodInfo *mi) {
    (void)lib; (void)cls;
    snprintf(R.methods[R.n_methods].name, 64, "%s", *(const char *const *)mi->name);
    R.methods[R.n_methods].mi = *mi;
    R.n_methods++;
}
static void i_unregister_class(GDExtensionClassLibraryPtr lib, GDExtensionConstStringNamePtr name) { (void)lib; (void)name; R.class_name[0] = 0; }
/* Variant <-> native */
static void v2t_int(GDExtensionTypePtr dst, GDExtensionVariantPtr v) { *(int64_t *)dst = ((MockVariant *)v)->u.i; }
static void v2t_packed(GDExtensionTypePtr dst, GDExtensionVariantPtr v) {   /* the engine shares the buffer; the mock copies it */
    const MockPacked *s = &((MockVariant *)v)->u.arr; MockPacked *d = (MockPacked *)dst;
    d->size = s->size; d->data = (uint8_t *)malloc((size_t)(s->size ? s->size : 1) * 4); memcpy(d->data, s->data, (size_t)s->size * (((MockVariant *)v)->type == GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY ? 1 : 4));
}
static GDExtensionTypeFromVariantConstructorFunc i_to_type(GDExtensionVariantType t) { return t == GDEXTENSION_VARIANT_TYPE_INT ? v2t_int : v2t_packed; }
static void t2v_int(GDExtensionVariantPtr v, GDExtensionTypePtr src) { MockVariant *m = (MockVariant *)v; m->type = GDEXTENSION_VARIANT_TYPE_INT; m->u.i = *(int64_t *)src; }
static void t2v_pba(GDExtensionVariantPtr v, GDExtensionTypePtr src) {
    MockVariant *m = (MockVariant *)v; const MockPacked *s = (const MockPacked *)src;
    m->type = GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY; m->u.arr.size = s->size; m->u.arr.data = (uint8_t *)malloc((size_t)(s->size ? s->size : 1)); memcpy(m->u.arr.data, s->data, (size_t)s->size);
}
static void t2v_str(GDExtensionVariantPtr v, GDExtensionTypePtr src) { MockVariant *m = (MockVariant *)v; m->type = GDEXTENSION_VARIANT_TYPE_STRING; m->u.s = strdup(*(char **)src); }
static GDExtensionVariantFromTypeConstructorFunc i_from_type(GDExtensionVariantType t) {
    return t == GDEXTENSION_VARIANT_TYPE_INT ? t2v_int : (t == GDEXTENSION_VARIANT_TYPE_STRING ? t2v_str : t2v_pba);
}
static void d_packed(GDExtensionTypePtr p) { free(((MockPacked *)p)->data); ((MockPacked *)p)->data = NULL; ((MockPacked *)p)->size = 0; }
static void d_string(GDExtensionTypePtr p) { free(*(char **)p); *(char **)p = NULL; }
static GDExtensionPtrDestructor i_get_destructor(GDExtensionVariantType t) { return t == GDEXTENSION_VARIANT_TYPE_STRING ? d_string : d_packed; }
static void b_size(GDExtensionTypePtr self, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r, int n) { (void)a; (void)n; *(int64_t *)r = ((MockPacked *)self)->size; }
static void b_resize(GDExtensionTypePtr self, const GDExtensionConstTypePtr *a, GDExtensionTypePtr r, int n) {
    MockPacked *p = (MockPacked *)self; (void)n;
    p->size = *(const int64_t *)a[0]; p->data = (uint8_t *)realloc(p->data, (size_t)(p->size ? p->size : 1)); *(int64_t *)r = 0;
}
static builtin_method i_get_builtin(GDExtensionVariantType t, GDExtensionConstStringNamePtr name, GDExtensionInt hash) {
    const char *n = *(const char *const *)name; (void)t;
    if (!strcmp(n, "size") && hash == 3173160232LL) return b_size;
    if (!strcmp(n, "resize") && hash == 848867239LL) return b_resize;
    return NULL;
}
static void c_default(GDExtensionTypePtr self, const GDExtensionConstTypePtr *a) { (void)a; memset(self, 0, sizeof(MockPacked)); }
static ptr_constructor i_get_ctor(GDExtensionVariantType t, int32_t idx) { (void)t; return idx == 0 ? c_default : NULL; }

static GDExtensionInterfaceFunctionPtr get_proc(const char *name) {
#define F(n, f) if (!strcmp(name, n)) return (GDExtensionInterfaceFunctionPtr)f
    F("mem_alloc", i_mem_alloc); F("mem_free", i_mem_free);
    F("string_name_new_with_latin1_chars", i_string_name_new); F("string_new_with_utf8_chars", i_string_new);
    F("packed_byte_array_operator_index", i_pba_index); F("packed_byte_array_operator_index_const", i_pba_index_const);
    F("packed_float32_array_operator_index_const", i_pfa_index_const); F("packed_int32_array_operator_index_const", i_pia_index_const);
    F("classdb_construct_object", i_construct_object); F("object_set_instance", i_object_set_instance);
    F("classdb_register_extension_class2", i_register_class); F("classdb_register_extension_class_method", i_register_method);
    F("classdb_unregister_extension_class", i_unregister_class);
    F("get_variant_to_type_constructor", i_to_type); F("get_variant_from_type_constructor", i_from_type);
    F("variant_get_ptr_destructor", i_get_destructor); F("variant_get_ptr_builtin_method", i_get_builtin);
    F("variant_get_ptr_constructor", i_get_ctor);
#undef F
    return NULL;
}

static const GDExtensionClassMethodInfo *method(const char *name) {
    int i;
    for (i = 0; i < R.n_methods; i++) if (!strcmp(R.methods[i].name, name)) return &R.methods[i].mi;
    return NULL;
}
static int64_t ptrcall_int(const char *name, const GDExtensionConstTypePtr *args) {
    int64_t r = -999;
    method(name)->ptrcall_func(method(name)->method_userdata, R.instance, args, &r);
    return r;
}
static MockPacked ptrcall_bytes(const char *name, const GDExtensionConstTypePtr *args) {
    MockPacked r = {NULL, 0};                      /* the return slot holds an initialised (empty) PackedByteArray, like the engine's */
    r.data = (uint8_t *)malloc(16); r.size = 16;   /* ... here even a NON-empty one: the shim must assign (destroy the old value), not construct over it */
    method(name)->ptrcall_func(method(name)->method_userdata, R.instance, args, &r);
    return r;
}

/* ---- a foreign allocator for the zero-copy path: HIP's virtual-memory API standing in for the engine's VkDeviceMemory ------------------
 * (hipMemCreate + hipMemExportToShareableHandle: a dma-buf fd, the kind of object VK_KHR_external_memory_fd hands out on amdgpu; the
 * declarations below are the handful of hip_runtime_api.h entries used, looked up at run time so that this C99 file needs no HIP headers) */
typedef struct { int type; int id; } MockMemLocation;
typedef struct { int type; int requestedHandleType; MockMemLocation location; void *win32HandleMetaData; struct { unsigned char c, g; unsigned short u; } allocFlags; } MockMemAllocationProp;
typedef struct { MockMemLocation location; int flags; } MockMemAccessDesc;
typedef struct { void *hip; void *handle; void *ptr; size_t size; int fd;
                 int (*memcpy_)(void *, const void *, size_t, int); int (*unmap)(void *, size_t); int (*addr_free)(void *, size_t); int (*release)(void *); } ForeignAlloc;
static int foreign_alloc(ForeignAlloc *f, size_t bytes) {
    MockMemAllocationProp prop; MockMemAccessDesc acc; size_t gran = 0;
    int (*granularity)(size_t *, const void *, int), (*create)(void **, size_t, const void *, unsigned long long), (*export_)(void *, void *, int, unsigned long long);
    int (*reserve)(void **, size_t, size_t, void *, unsigned long long), (*map)(void *, size_t, size_t, void *, unsigned long long), (*set_access)(void *, size_t, const void *, size_t);
    memset(f, 0, sizeof *f); f->fd = -1;
    f->hip = dlopen("libamdhip64.so", RTLD_NOW);
    if (!f->hip) return -1;
    *(void **)&granularity = dlsym(f->hip, "hipMemGetAllocationGranularity"); *(void **)&create = dlsym(f->hip, "hipMemCreate");
    *(void **)&export_ = dlsym(f->hip, "hipMemExportToShareableHandle"); *(void **)&reserve = dlsym(f->hip, "hipMemAddressReserve");
    *(void **)&map = dlsym(f->hip, "hipMemMap"); *(void **)&set_access = dlsym(f->hip, "hipMemSetAccess");
    *(void **)&f->memcpy_ = dlsym(f->hip, "hipMemcpy"); *(void **)&f->unmap = dlsym(f->hip, "hipMemUnmap");
    *(void **)&f->addr_free = dlsym(f->hip, "hipMemAddressFree"); *(void **)&f->release = dlsym(f->hip, "hipMemRelease");
    if (!granularity || !create || !export_ || !reserve || !map || !set_access || !f->memcpy_ || !f->unmap || !f->addr_free || !f->release) return -2;
    memset(&prop, 0, sizeof prop); prop.type = 1; prop.requestedHandleType = 1; prop.location.type = 1; prop.location.id = 0;   /* pinned, POSIX fd, device 0 */
    if (granularity(&gran, &prop, 0) != 0 || gran == 0) return -3;
    f->size = (bytes + gran - 1) / gran * gran;
    if (create(&f->handle, f->size, &prop, 0) != 0) return -4;
    if (export_(&f->fd, f->handle, 1, 0) != 0) return -5;
    if (reserve(&f->ptr, f->size, 0, NULL, 0) != 0 || map(f->ptr, f->size, 0, f->handle, 0) != 0) return -6;
    memset(&acc, 0, sizeof acc); acc.location.type = 1; acc.location.id = 0; acc.flags = 3;
    if (set_access(f->ptr, f->size, &acc, 1) != 0) return -7;
    return 0;
}
static void foreign_free(ForeignAlloc *f) { if (f->ptr) { f->unmap(f->ptr, f->size); f->addr_free(f->ptr, f->size); } if (f->handle) f->release(f->handle); }

int main(int argc, char **argv) {
    GDExtensionInitialization init;
    GDExtensionInitializationFunction entry;
    void *so;
    static const char *expect[] = {"create", "set_noise", "set_march", "render_transmittance", "render_sky_lut", "render_clouds", "get_status", "get_last_error",
                                   "create_multi", "set_noise_mips", "set_frames", "submit_clouds", "collect", "is_ready",
                                   "import_frame_fd", "render_clouds_into", "frame_ready", "release_frame", "get_last_warning"};
    static const int expect_argc[] = {1, 3, 2, 1, 1, 3, 0, 0, 1, 3, 1, 3, 1, 1, 2, 2, 1, 1, 0};
    enum { N_EXPECT = 19 };
    int i;
    if (argc < 2) return 2;
    so = dlopen(argv[1], RTLD_NOW);
    if (!so) { fprintf(stderr, "%s\n", dlerror()); return 3; }
    entry = (GDExtensionInitializationFunction)dlsym(so, "csky_gdextension_init");     /* entry_symbol of gdext/cloudsky.gdextension */
    if (!entry) return 4;
    memset(&init, 0, sizeof init);
    if (!entry(get_proc, (GDExtensionClassLibraryPtr)&R, &init)) return 5;
    if (init.minimum_initialization_level != GDEXTENSION_INITIALIZATION_SCENE || !init.initialize || !init.deinitialize) return 6;
    init.initialize(init.userdata, GDEXTENSION_INITIALIZATION_CORE);
    if (R.class_name[0]) return 7;                                  /* nothing may register before the SCENE level */
    init.initialize(init.userdata, GDEXTENSION_INITIALIZATION_SCENE);
    if (strcmp(R.class_name, "CloudSkyHIP") || strcmp(R.parent, "RefCounted") || !R.ci.create_instance_func || !R.ci.free_instance_func || !R.ci.is_exposed) return 8;
    if (R.n_methods != N_EXPECT) return 9;
    for (i = 0; i < N_EXPECT; i++) {
        const GDExtensionClassMethodInfo *mi = method(expect[i]);
        if (!mi || (int)mi->argument_count != expect_argc[i] || !mi->ptrcall_func || !mi->call_func || !mi->has_return_value) return 10;
    }
    free(R.ci.create_instance_func(R.ci.class_userdata));
    if (!R.instance) return 20;
    {   /* error paths before create(): ERR_STATE, empty image; through ptrcall and through the Variant trampoline */
        float pcf[28]; MockPacked pc = {(uint8_t *)pcf, 28}; int64_t w = 64, h = 32;
        GDExtensionConstTypePtr a[3] = {&pc, &w, &h};
        MockPacked img;
        MockVariant va[3], vr; const GDExtensionConstVariantPtr vp[3] = {&va[0], &va[1], &va[2]}; GDExtensionCallError ce;
        ctu_default_push_constant(pcf, 64.0f, 32.0f);
        img = ptrcall_bytes("render_clouds", a);
        if (img.size != 0 || ptrcall_int("get_status", NULL) != CSKY_ERR_STATE) return 21;
        memset(va, 0, sizeof va); memset(&vr, 0, sizeof vr);
        va[0].type = GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY; va[0].u.arr = pc;
        va[1].type = va[2].type = GDEXTENSION_VARIANT_TYPE_INT; va[1].u.i = 64; va[2].u.i = 32;
        ce.error = GDEXTENSION_CALL_ERROR_INVALID_METHOD;
        method("render_clouds")->call_func(method("render_clouds")->method_userdata, R.instance, vp, 3, &vr, &ce);
        if (ce.error != GDEXTENSION_CALL_OK || vr.type != GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY || vr.u.arr.size != 0) return 22;
        method("render_clouds")->call_func(method("render_clouds")->method_userdata, R.instance, vp, 2, &vr, &ce);
        if (ce.error != GDEXTENSION_CALL_ERROR_TOO_FEW_ARGUMENTS) return 23;
        memset(&vr, 0, sizeof vr);
        method("get_last_error")->call_func(method("get_last_error")->method_userdata, R.instance, NULL, 0, &vr, &ce);
        if (vr.type != GDEXTENSION_VARIANT_TYPE_STRING || !strstr(vr.u.s, "create()")) return 24;
        memset(&vr, 0, sizeof vr);
        method("get_last_warning")->call_func(method("get_last_warning")->method_userdata, R.instance, NULL, 0, &vr, &ce);
        if (ce.error != GDEXTENSION_CALL_OK || vr.type != GDEXTENSION_VARIANT_TYPE_STRING || vr.u.s[0] != 0) return 27;      /* no context yet: "" */
        if (ptrcall_int("submit_clouds", a) != CSKY_ERR_STATE) return 25;             /* the asynchronous form before create() */
        { int64_t t = 0; GDExtensionConstTypePtr ta[1] = {&t}; if (ptrcall_bytes("collect", ta).size != 0 || ptrcall_int("is_ready", ta) != CSKY_ERR_STATE) return 26; }
    }
    if (argc >= 4) {   /* the whole chain on the GPU through the shim */
        uint8_t *large, *small, *weather;
        uint16_t *ref = (uint16_t *)ctu_read_file(argv[3], (size_t)64 * 32 * 8);
        int64_t dev = 0, prim = 128, light = 6, w = 64, h = 32;
        float pcf[28], tpc[4] = {256, 64, 0, 0}, spc[8] = {200, 100, 0, 0, 0, 0, 0, 0};
        MockPacked pl, ps, pw, pc = {(uint8_t *)pcf, 28}, ptc = {(uint8_t *)tpc, 4}, psc = {(uint8_t *)spc, 8}, img, lut;
        GDExtensionConstTypePtr a[3];
        int bad;
        if (!ref || ctu_default_noise(argv[2], &large, &small, &weather) != 0) return 30;
        a[0] = &dev;
        if (ptrcall_int("create", a) != CSKY_OK) return 31;                           /* exit 31 = no usable GPU */
        pl.data = large; pl.size = 128 * 128 * 128 * 4; ps.data = small; ps.size = 32 * 32 * 32 * 3; pw.data = weather; pw.size = 512 * 512 * 3;
        a[0] = &pl; a[1] = &ps; a[2] = &pw;
        if (ptrcall_int("set_noise", a) != CSKY_OK) return 32;
        ps.size -= 1;
        if (ptrcall_int("set_noise", a) != CSKY_ERR_INVALID) return 33;               /* wrong array size is refused, not read */
        {   /* every mip level back to back (the importer's chains): routed to csky_set_noise_mips; here the library's own box chains, so the frame below is unchanged */
            const size_t nl = csky_mip_offset(128, 8, 4), ns = csky_mip_offset(32, 6, 3);
            uint8_t *lc = (uint8_t *)calloc(nl, 1), *sc = (uint8_t *)calloc(ns, 1);
            int rc_chain;
            if (!lc || !sc) return 30;
            memcpy(lc, large, (size_t)128 * 128 * 128 * 4); memcpy(sc, small, (size_t)32 * 32 * 32 * 3);
            if (csky_build_mips(lc, 128, 4, 8) != CSKY_OK || csky_build_mips(sc, 32, 3, 6) != CSKY_OK) return 30;
            pl.data = lc; pl.size = (int64_t)nl; ps.data = sc; ps.size = (int64_t)ns;
            rc_chain = ptrcall_int("set_noise", a);
            free(lc); free(sc);
            if (rc_chain != CSKY_OK) return 39;
        }
        a[0] = &prim; a[1] = &light;
        if (ptrcall_int("set_march", a) != CSKY_OK) return 34;
        a[0] = &ptc; lut = ptrcall_bytes("render_transmittance", a);
        if (lut.size != 256 * 64 * 8) return 35;
        ctu_default_push_constant(pcf, 64.0f, 32.0f);
        spc[4] = pcf[16]; spc[5] = pcf[17]; spc[6] = pcf[18];                          /* sky_lut.gd:123-132 */
        a[0] = &psc; lut = ptrcall_bytes("render_sky_lut", a);
        if (lut.size != 200 * 100 * 8) return 36;
        a[0] = &pc; a[1] = &w; a[2] = &h;
        img = ptrcall_bytes("render_clouds", a);
        if (img.size != 64 * 32 * 8 || ptrcall_int("get_status", NULL) != CSKY_OK) return 37;
        bad = ctu_bad_pixels((const uint16_t *)img.data, ref, 64 * 32);
        printf("shim frame vs fixture: %d pixels beyond 2 fp16 ulp\n", bad);
        if (bad < 0 || bad > 2) return 38;
        {   /* the same frame through the Variant call: byte-identical */
            MockVariant va[3], vr; const GDExtensionConstVariantPtr vp[3] = {&va[0], &va[1], &va[2]}; GDExtensionCallError ce;
            memset(va, 0, sizeof va); memset(&vr, 0, sizeof vr);
            va[0].type = GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY; va[0].u.arr = pc;
            va[1].type = va[2].type = GDEXTENSION_VARIANT_TYPE_INT; va[1].u.i = 64; va[2].u.i = 32;
            method("render_clouds")->call_func(method("render_clouds")->method_userdata, R.instance, vp, 3, &vr, &ce);
            if (ce.error != GDEXTENSION_CALL_OK || vr.u.arr.size != img.size || memcmp(vr.u.arr.data, img.data, (size_t)img.size)) return 39;
        }
        {   /* LUT sizes are script-controlled FLOATS: NaN / huge / negative must be refused before any cast or allocation */
            static const float bad_sizes[][2] = {{0.0f / 0.0f, 64}, {1e30f, 1e30f}, {-5, 64}, {256, 0}, {1.0f / 0.0f, 1}};
            size_t k;
            for (k = 0; k < sizeof bad_sizes / sizeof bad_sizes[0]; k++) {
                float tb[4] = {bad_sizes[k][0], bad_sizes[k][1], 0, 0}, sb[8] = {bad_sizes[k][0], bad_sizes[k][1], 0, 0, 0, 1, 0, 0};
                MockPacked ptb = {(uint8_t *)tb, 4}, psb = {(uint8_t *)sb, 8};
                a[0] = &ptb; if (ptrcall_bytes("render_transmittance", a).size != 0 || ptrcall_int("get_status", NULL) != CSKY_ERR_INVALID) return 41;
                a[0] = &psb; if (ptrcall_bytes("render_sky_lut", a).size != 0 || ptrcall_int("get_status", NULL) != CSKY_ERR_INVALID) return 42;
            }
        }
        {   /* the throughput path: two tickets outstanding over the pinned ring, collected out of order; frames byte-identical to the blocking call */
            int64_t slots = 2, t0, t1, t2, q;
            GDExtensionConstTypePtr ta[1];
            MockPacked f0, f1;
            ta[0] = &slots; if (ptrcall_int("set_frames", ta) != CSKY_OK) return 43;
            a[0] = &pc; a[1] = &w; a[2] = &h;
            t0 = ptrcall_int("submit_clouds", a); t1 = ptrcall_int("submit_clouds", a);
            if (t0 != 0 || t1 != 1) return 44;
            t2 = ptrcall_int("submit_clouds", a);                                      /* a third frame while two are outstanding: refused, nothing lost */
            if (t2 != CSKY_ERR_STATE) return 45;
            ta[0] = &t1; f1 = ptrcall_bytes("collect", ta);
            ta[0] = &t0; q = ptrcall_int("is_ready", ta); if (q != 0 && q != 1) return 46;
            f0 = ptrcall_bytes("collect", ta);
            if (f0.size != img.size || f1.size != img.size || memcmp(f0.data, img.data, (size_t)img.size) || memcmp(f1.data, img.data, (size_t)img.size)) return 47;
            if (ptrcall_bytes("collect", ta).size != 0 || ptrcall_int("get_status", NULL) != CSKY_ERR_STATE) return 48;    /* collected already */
            t2 = ptrcall_int("submit_clouds", a); if (t2 != 2) return 49;             /* tickets keep counting; the ring slot is reused */
            ta[0] = &t2; f0 = ptrcall_bytes("collect", ta);
            if (f0.size != img.size || memcmp(f0.data, img.data, (size_t)img.size)) return 50;
        }
        {   /* the zero-copy path: the march writes into memory a FOREIGN allocator owns (exported as a dma-buf fd, imported at an offset with a
             * padded row pitch); read back through the exporter's own mapping: byte-identical to the blocking call, padding untouched */
            ForeignAlloc fa;
            const int32_t pitch = 64 * 8 + 128, offset = 4096;
            int32_t layout[5];
            MockPacked pl5 = {(uint8_t *)layout, 5};
            int64_t fdv, slot, q;
            GDExtensionConstTypePtr za[2];
            const int frc = foreign_alloc(&fa, (size_t)offset + (size_t)pitch * 32);
            if (frc == 0) {
                uint8_t *back = (uint8_t *)malloc((size_t)pitch * 32);
                int y, same = 1, pad_ok = 1;
                layout[0] = (int32_t)fa.size; layout[1] = offset; layout[2] = pitch; layout[3] = 64; layout[4] = 32;
                fdv = dup(fa.fd); za[0] = &fdv; za[1] = &pl5;
                slot = ptrcall_int("import_frame_fd", za);
                if (slot != 0) { fprintf(stderr, "import_frame_fd: %lld\n", (long long)slot); return 60; }
                layout[0] = 1000; fdv = 0;                                           /* a frame that does not fit its allocation is refused before the fd is touched */
                if (ptrcall_int("import_frame_fd", za) != CSKY_ERR_INVALID) return 61;
                q = 3; za[0] = &q; if (ptrcall_int("frame_ready", za) != CSKY_ERR_INVALID) return 62;            /* empty slot */
                za[0] = &slot; if (ptrcall_int("frame_ready", za) != CSKY_ERR_STATE) return 63;                 /* nothing marched into it yet */
                {   /* fill the window with a pattern through the exporter's mapping, then march */
                    uint8_t *pat = (uint8_t *)malloc((size_t)pitch * 32); memset(pat, 0xA7, (size_t)pitch * 32);
                    if (fa.memcpy_((uint8_t *)fa.ptr + offset, pat, (size_t)pitch * 32, 1) != 0) return 64;      /* hipMemcpyHostToDevice */
                    free(pat);
                }
                za[0] = &slot; za[1] = &pc;
                if (ptrcall_int("render_clouds_into", za) != CSKY_OK) return 65;
                do { q = ptrcall_int("frame_ready", za); } while (q == 0);
                if (q != 1) return 66;
                if (fa.memcpy_(back, (uint8_t *)fa.ptr + offset, (size_t)pitch * 32, 2) != 0) return 67;         /* hipMemcpyDeviceToHost, the EXPORTER's address */
                for (y = 0; y < 32; y++) {
                    int x;
                    if (memcmp(back + (size_t)y * pitch, img.data + (size_t)y * 64 * 8, 64 * 8)) same = 0;
                    for (x = 64 * 8; x < pitch; x++) if (back[(size_t)y * pitch + x] != 0xA7) pad_ok = 0;
                }
                printf("zero-copy frame in the foreign allocation vs the blocking call: %s, row padding untouched: %s\n", same ? "identical" : "DIFFERENT", pad_ok ? "yes" : "NO");
                if (!same || !pad_ok) return 68;
                if (ptrcall_int("release_frame", za) != CSKY_OK || ptrcall_int("release_frame", za) != CSKY_ERR_INVALID) return 69;
                free(back);
                foreign_free(&fa);
            } else {
                printf("zero-copy scenario skipped: no exportable allocation from this runtime (%d)\n", frc);
            }
        }
        {   /* create_multi: two contexts behind the object (both on device 0 here); explicit set_noise_mips; blocking and asynchronous frames */
            int32_t ids[2] = {0, 0};
            MockPacked pid = {(uint8_t *)ids, 2}, f;
            const size_t nl = csky_mip_offset(128, 8, 4), ns = csky_mip_offset(32, 6, 3);
            uint8_t *lc = (uint8_t *)calloc(nl, 1), *sc = (uint8_t *)calloc(ns, 1);
            int64_t t;
            GDExtensionConstTypePtr ta[1];
            a[0] = &pid; if (ptrcall_int("create_multi", a) != CSKY_OK) return 51;
            memcpy(lc, large, (size_t)128 * 128 * 128 * 4); memcpy(sc, small, (size_t)32 * 32 * 32 * 3);
            if (csky_build_mips(lc, 128, 4, 8) != CSKY_OK || csky_build_mips(sc, 32, 3, 6) != CSKY_OK) return 30;
            pl.data = lc; pl.size = (int64_t)nl; ps.data = sc; ps.size = (int64_t)ns;
            a[0] = &pl; a[1] = &ps; a[2] = &pw;
            if (ptrcall_int("set_noise_mips", a) != CSKY_OK) return 52;
            pl.size -= 4; if (ptrcall_int("set_noise_mips", a) != CSKY_ERR_INVALID) return 53; pl.size += 4;
            a[0] = &prim; a[1] = &light; if (ptrcall_int("set_march", a) != CSKY_OK) return 54;
            a[0] = &psc; lut = ptrcall_bytes("render_sky_lut", a); if (lut.size != 200 * 100 * 8) return 55;
            a[0] = &pc; a[1] = &w; a[2] = &h;
            f = ptrcall_bytes("render_clouds", a);
            bad = (f.size == 64 * 32 * 8) ? ctu_bad_pixels((const uint16_t *)f.data, ref, 64 * 32) : -1;
            if (bad < 0 || bad > 2) return 56;                                        /* (a 64x32 tile over two devices runs as ray segments: equal to rounding) */
            t = ptrcall_int("submit_clouds", a); if (t < 0) return 57;
            ta[0] = &t; f = ptrcall_bytes("collect", ta);
            bad = (f.size == 64 * 32 * 8) ? ctu_bad_pixels((const uint16_t *)f.data, ref, 64 * 32) : -1;
            if (bad < 0 || bad > 2) return 58;
            free(lc); free(sc);
        }
    }
    R.ci.free_instance_func(R.ci.class_userdata, R.instance);
    init.deinitialize(init.userdata, GDEXTENSION_INITIALIZATION_SCENE);
    if (R.class_name[0]) return 40;
    printf("gdext mock host ok\n");
    return 0;
}
