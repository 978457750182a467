/*
 * cloudsky_oracle.c -- TEST INFRASTRUCTURE ONLY (see cloudsky_oracle.h).  PARITY UNPINNED.
 *
 * Scalar fp32 restatement, one C function per GLSL function, of
 *   /root/reference/cloud_sky/clouds.glsl            (cited below as C:line)
 *   /root/reference/cloud_sky/sky-lut.glsl           (S:line)
 *   /root/reference/cloud_sky/transmittance-lut.glsl (T:line)
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  Every arithmetic
 * expression keeps the GLSL's left-to-right association; all maths is float (never double)
 * except compile-time constant folding, which glslang performs in double before narrowing.
 * Round 6: bit-identical, on every fixture, to the reference's own shader text executed under
 * oracle/glsl_exec (tests/test_oracle_glslexec.py; the one discrepancy that run found -- two constant
 * expressions of the compositor evaluated in float here -- is fixed below, G:74 / G:82).
 *
 * Third-party behaviour that is NOT in the reference tree (Godot Engine >= 4.2 / Vulkan):
 *   - sampler filtering: restated from the Vulkan spec texel-coordinate rules (u = s*size,
 *     i0 = floor(u-0.5), weights = fract(u-0.5), REPEAT = integer mod, CLAMP_TO_EDGE = integer
 *     clamp), exact fp32 weights, nested lerp x then y then z;
 *   - integer LOD with a LINEAR mip filter selects exactly one level, clamped to [0, levels-1];
 *   - rgba16f imageStore: round-to-nearest-even (implementation defined in Vulkan);
 *   - 3-D mip chain: 2x2x2 box, (sum+4)>>3;
 *   - BC7 compression of the inputs is NOT modelled (uncompressed UNORM8 texels).
 */
#include "cloudsky_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small helpers */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 divs3(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 normalize3(v3 a) { return divs3(a, length3(a)); }      /* GLSL: x / length(x) */
static inline v3 mix3(v3 a, v3 b, float t) { return add3(muls3(a, 1.0f - t), muls3(b, t)); }

static inline v4 V4(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }
static inline v4 add4(v4 a, v4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline v4 sub4(v4 a, v4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline v4 mul4(v4 a, v4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline v4 div4(v4 a, v4 b) { return V4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
static inline v4 muls4(v4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline v4 exp4(v4 a) { return V4(expf(a.x), expf(a.y), expf(a.z), expf(a.w)); }
static inline v4 maxs4(v4 a, float s) { return V4(fmaxf(a.x, s), fmaxf(a.y, s), fmaxf(a.z, s), fmaxf(a.w, s)); }

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); } /* NaN -> lo */
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline float fractf(float x) { return x - floorf(x); }
static inline float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline float smoothstepf(float e0, float e1, float x) {
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

/* ------------------------------------------------------------------ half conversion */
uint16_t csko_f2h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t em = x & 0x7fffffffu;
    if (em >= 0x7f800000u) return (uint16_t)(sign | (em > 0x7f800000u ? 0x7e00u : 0x7c00u)); /* nan / inf */
    if (em >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);        /* rounds to >= 65520 -> inf      */
    if (em < 0x33000001u) return (uint16_t)sign;                     /* <= 2^-25 -> 0 (RNE tie->even)   */
    int e = (int)(em >> 23) - 127;
    uint32_t m = (em & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                   /* denormal half needs more shift   */
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t h = (e < -14) ? q : (((uint32_t)(e + 15) << 10) + (q - 0x400u)); /* carry flows into exponent */
    return (uint16_t)(sign | h);
}
float csko_h2f(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int s = 0; while (!(m & 0x400u)) { m <<= 1; s++; } x = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ffu) << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}

/* ------------------------------------------------------------------ mips */
size_t csko_mip_offset(int n, int level, int ch) {
    size_t off = 0;
    for (int l = 0; l < level; l++) { size_t m = (size_t)(n >> l); off += m * m * m * (size_t)ch; }
    return off;
}
size_t csko_mip_total(int n, int levels, int ch) { return csko_mip_offset(n, levels, ch); }

void csko_build_mips(uint8_t *vol, int n, int ch, int levels) {
    for (int l = 1; l < levels; l++) {
        const uint8_t *src = vol + csko_mip_offset(n, l - 1, ch);
        uint8_t *dst = vol + csko_mip_offset(n, l, ch);
        int ns = n >> (l - 1), nd = n >> l;
        for (int z = 0; z < nd; z++) for (int y = 0; y < nd; y++) for (int x = 0; x < nd; x++)
            for (int c = 0; c < ch; c++) {
                unsigned s = 0;
                for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++)
                    s += src[((((size_t)(2 * z + dz)) * ns + (2 * y + dy)) * ns + (2 * x + dx)) * ch + c];
                dst[(((size_t)z * nd + y) * nd + x) * ch + c] = (uint8_t)((s + 4u) >> 3);
            }
    }
}

/* ------------------------------------------------------------------ samplers */
static inline int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
static inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }
static inline float lerpf(float a, float b, float f) { return a + (b - a) * f; }

/* REPEAT, LINEAR 3-D tap at one integer mip level; nch channels of UNORM8 -> float (texel/255). */
static void sample3d_repeat(const uint8_t *base, int n0, int levels, int ch, float lod, v3 s, float out[4]) {
    int level = (int)floorf(clampf(lod, 0.0f, (float)(levels - 1)));
    int n = n0 >> level;
    const uint8_t *t = base + csko_mip_offset(n0, level, ch);
    float ux = s.x * (float)n - 0.5f, uy = s.y * (float)n - 0.5f, uz = s.z * (float)n - 0.5f;
    float fx0 = floorf(ux), fy0 = floorf(uy), fz0 = floorf(uz);
    float ax = ux - fx0, ay = uy - fy0, az = uz - fz0;
    int x0 = wrapi((int)fx0, n), y0 = wrapi((int)fy0, n), z0 = wrapi((int)fz0, n);
    int x1 = (x0 + 1 == n) ? 0 : x0 + 1, y1 = (y0 + 1 == n) ? 0 : y0 + 1, z1 = (z0 + 1 == n) ? 0 : z0 + 1;
    for (int c = 0; c < ch; c++) {
#define TX(X, Y, Z) ((float)t[((((size_t)(Z)) * n + (Y)) * n + (X)) * ch + c] / 255.0f)
        float c00 = lerpf(TX(x0, y0, z0), TX(x1, y0, z0), ax);
        float c10 = lerpf(TX(x0, y1, z0), TX(x1, y1, z0), ax);
        float c01 = lerpf(TX(x0, y0, z1), TX(x1, y0, z1), ax);
        float c11 = lerpf(TX(x0, y1, z1), TX(x1, y1, z1), ax);
#undef TX
        out[c] = lerpf(lerpf(c00, c10, ay), lerpf(c01, c11, ay), az);
    }
}

/* REPEAT, LINEAR 2-D tap, RGB8 (weather, C:174; sampler cloud_sky.gd:301-309), LOD 0 */
static v3 sample_weather(const uint8_t *t, float sx, float sy) {
    const int n = 512;
    float ux = sx * 512.0f - 0.5f, uy = sy * 512.0f - 0.5f;
    float fx0 = floorf(ux), fy0 = floorf(uy);
    float ax = ux - fx0, ay = uy - fy0;
    int x0 = wrapi((int)fx0, n), y0 = wrapi((int)fy0, n);
    int x1 = (x0 + 1) & 511, y1 = (y0 + 1) & 511;
    float o[3];
    for (int c = 0; c < 3; c++) {
#define TX(X, Y) ((float)t[(((size_t)(Y)) * n + (X)) * 3 + c] / 255.0f)
        o[c] = lerpf(lerpf(TX(x0, y0), TX(x1, y0), ax), lerpf(TX(x0, y1), TX(x1, y1), ax), ay);
#undef TX
    }
    return V3(o[0], o[1], o[2]);
}

/* CLAMP_TO_EDGE, LINEAR 2-D tap on an RGBA16F image (sky LUT: cloud_sky.gd:381-390; transmittance: sky_lut.gd:62-68) */
static v4 sample_rgba16f_clamp(const uint16_t *t, int w, int h, float sx, float sy) {
    float ux = sx * (float)w - 0.5f, uy = sy * (float)h - 0.5f;
    float fx0 = floorf(ux), fy0 = floorf(uy);
    float ax = ux - fx0, ay = uy - fy0;
    int x0 = clampi((int)fx0, 0, w - 1), x1 = clampi((int)fx0 + 1, 0, w - 1);
    int y0 = clampi((int)fy0, 0, h - 1), y1 = clampi((int)fy0 + 1, 0, h - 1);
    float o[4];
    for (int c = 0; c < 4; c++) {
#define TX(X, Y) csko_h2f(t[(((size_t)(Y)) * w + (X)) * 4 + c])
        o[c] = lerpf(lerpf(TX(x0, y0), TX(x1, y0), ax), lerpf(TX(x0, y1), TX(x1, y1), ax), ay);
#undef TX
    }
    return V4(o[0], o[1], o[2], o[3]);
}

/* ====================================================================================
 * Atmosphere helpers shared by transmittance-lut.glsl (T:45-145) and sky-lut.glsl (S:44-202)
 * (the two shaders carry identical copies).  Units: km.
 * ==================================================================================== */
#define EARTH_RADIUS 6371.0f                         /* T:50  S:58 */
#define ATMOSPHERE_THICKNESS 100.0f                  /* T:51  S:59 */
#define ATMOSPHERE_RADIUS 6471.0f                    /* T:52  S:60 */
static const float sun_spectral_irradiance[4] = {1.679f, 1.828f, 1.986f, 1.307f};                 /* S:67 */
static const float molecular_scattering_coefficient_base[4] = {6.605e-3f, 1.067e-2f, 1.842e-2f, 3.156e-2f}; /* T:60 S:71 */
/* T:64,67 / S:75,78: ozone_absorption_cross_section(*1e-4f) * ozone_mean_monthly_dobson(350): constant folded (double) */
static const float ozone_cross_times_dobson[4] = {(float)(3.472e-21 * 1e-4 * 350.0), (float)(3.914e-21 * 1e-4 * 350.0),
                                                  (float)(1.349e-21 * 1e-4 * 350.0), (float)(11.03e-23 * 1e-4 * 350.0)};
static const float aerosol_absorption_cross_section[4] = {2.8722e-24f, 4.6168e-24f, 7.9706e-24f, 1.3578e-23f}; /* T:74 S:85 */
static const float aerosol_scattering_cross_section[4] = {1.5908e-22f, 1.7711e-22f, 2.0942e-22f, 2.4033e-22f}; /* T:75 S:86 */
#define aerosol_base_density 1.3681e20f              /* T:76 S:87 */
#define aerosol_height_scale 0.73f                   /* T:78 S:89 */
static const float aerosol_background_divided_by_base_density = (float)(2e6 / 1.3681e20);        /* T:80 S:91 */

/* T:89-98 / S:100-109 */
static float ray_sphere_intersection(v3 ro, v3 rd, float radius) {
    float b = dot3(ro, rd);
    float c = dot3(ro, ro) - radius * radius;
    if (c > 0.0f && b > 0.0f) return -1.0f;
    float d = b * b - c;
    if (d < 0.0f) return -1.0f;
    if (d > b * b) return (-b + sqrtf(d));
    return (-b - sqrtf(d));
}

/* T:104-107 / S:132-135 */
static v4 get_molecular_scattering_coefficient(float h) {
    float e = expf(-0.07771971f * powf(h, 1.16364243f));
    const float *b = molecular_scattering_coefficient_base;
    return V4(b[0] * e, b[1] * e, b[2] * e, b[3] * e);
}
/* T:113-119 / S:170-176 */
static v4 get_molecular_absorption_coefficient(float h) {
    h += 1e-4f;
    float t = logf(h) - 3.22261f;
    float density = 3.78547397e20f * (1.0f / h) * expf(-t * t * 5.55555555f);
    const float *o = ozone_cross_times_dobson;
    return V4(o[0] * density, o[1] * density, o[2] * density, o[3] * density);
}
/* T:121-125 / S:178-182 */
static float get_aerosol_density(float h) {
    return aerosol_base_density * (expf(-h / aerosol_height_scale) + aerosol_background_divided_by_base_density);
}
/* T:131-145 / S:188-202 */
static void get_atmosphere_collision_coefficients(float h, v4 *aerosol_absorption, v4 *aerosol_scattering,
                                                  v4 *molecular_absorption, v4 *molecular_scattering, v4 *extinction) {
    h = fmaxf(h, 0.0f);
    float ad = get_aerosol_density(h);
    const float *a = aerosol_absorption_cross_section, *s = aerosol_scattering_cross_section;
    *aerosol_absorption = V4(a[0] * ad, a[1] * ad, a[2] * ad, a[3] * ad);
    *aerosol_scattering = V4(s[0] * ad, s[1] * ad, s[2] * ad, s[3] * ad);
    *molecular_absorption = get_molecular_absorption_coefficient(h);
    *molecular_scattering = get_molecular_scattering_coefficient(h);
    *extinction = add4(add4(add4(*aerosol_absorption, *aerosol_scattering), *molecular_absorption), *molecular_scattering);
}

/* ==================================================================================== transmittance-lut.glsl */
#define TRANSMITTANCE_STEPS 40                        /* T:45 */
/* T:157-196 main() */
void csko_transmittance_lut(int w, int h, uint16_t *out) {
    for (int py = 0; py < h; py++) for (int px = 0; px < w; px++) {
        float uvx = (float)px / (float)w, uvy = (float)py / (float)h;                       /* T:162 */
        float sun_cos_theta = uvx * 2.0f - 1.0f;                                             /* T:164 */
        v3 sun_dir = V3(-sqrtf(1.0f - sun_cos_theta * sun_cos_theta), 0.0f, sun_cos_theta);  /* T:165 */
        float distance_to_earth_center = mixf(EARTH_RADIUS, ATMOSPHERE_RADIUS, uvy);         /* T:167 */
        v3 ray_origin = V3(0.0f, 0.0f, distance_to_earth_center);                            /* T:168 */
        float t_d = ray_sphere_intersection(ray_origin, sun_dir, ATMOSPHERE_RADIUS);         /* T:170 */
        float dt = t_d / (float)TRANSMITTANCE_STEPS;                                         /* T:171 */
        v4 result = V4(0, 0, 0, 0);
        for (int i = 0; i < TRANSMITTANCE_STEPS; ++i) {                                      /* T:175 */
            float t = ((float)i + 0.5f) * dt;
            v3 x_t = add3(ray_origin, muls3(sun_dir, t));
            float altitude = length3(x_t) - EARTH_RADIUS;                                    /* T:179 */
            v4 aa, as, ma, ms, ext;
            get_atmosphere_collision_coefficients(altitude, &aa, &as, &ma, &ms, &ext);
            result = add4(result, muls4(ext, dt));                                           /* T:190 */
        }
        v4 tr = exp4(V4(-result.x, -result.y, -result.z, -result.w));                        /* T:193 */
        uint16_t *o = out + ((size_t)py * w + px) * 4;
        o[0] = csko_f2h(tr.x); o[1] = csko_f2h(tr.y); o[2] = csko_f2h(tr.z); o[3] = csko_f2h(tr.w); /* T:195 */
    }
}

/* ==================================================================================== sky-lut.glsl */
#define S_PI 3.14159265358979323846                    /* S:44 (double; folded then narrowed) */
#define S_INV_PI 0.31830988618379067154                /* S:45 */
#define IN_SCATTERING_STEPS 30                         /* S:53 */
#define EYE_DISTANCE_TO_EARTH_CENTER 6371.5f           /* S:61-62 */
static const float PHASE_ISOTROPIC = (float)(0.25 * S_INV_PI);                 /* S:46-47 */
static const float RAYLEIGH_PHASE_SCALE = (float)((3.0 / 16.0) * S_INV_PI);    /* S:48 */
static const float S_g = 0.8f;                                                 /* S:49 */

typedef struct { const uint16_t *t; int w, h; } lut2d;

/* S:114-117 */
static float molecular_phase_function(float c) { return RAYLEIGH_PHASE_SCALE * (1.0f + c * c); }
/* S:122-126 */
static float aerosol_phase_function(float c) {
    const float gg = (float)(0.8 * 0.8);
    float den = (float)(1.0 + 0.8 * 0.8) + (float)(2.0 * 0.8) * c;
    (void)S_g;
    return (float)(0.25 * S_INV_PI) * (1.0f - gg) / (den * sqrtf(den));
}
/* S:137-142 */
static v4 transmittance_from_lut(lut2d lut, float cos_theta, float normalized_altitude) {
    float u = clampf(cos_theta * 0.5f + 0.5f, 0.0f, 1.0f);
    float v = clampf(normalized_altitude, 0.0f, 1.0f);
    return sample_rgba16f_clamp(lut.t, lut.w, lut.h, u, v);
}
/* S:144-164 */
static v4 get_multiple_scattering(lut2d lut, float cos_theta, float normalized_height, float d) {
    float omega = (float)(2.0 * S_PI) * (1.0f - sqrtf(d * d - EARTH_RADIUS * EARTH_RADIUS) / d);
    v4 T_to_ground = transmittance_from_lut(lut, cos_theta, 0.0f);
    v4 T_ground_to_sample = div4(transmittance_from_lut(lut, 1.0f, 0.0f), transmittance_from_lut(lut, 1.0f, normalized_height));
    const float albedo_over_pi = (float)(0.3 / S_PI);                            /* GROUND_ALBEDO / PI, S:63,157 */
    v4 L_ground = muls4(mul4(mul4(muls4(V4(1, 1, 1, 1), PHASE_ISOTROPIC * omega * albedo_over_pi), T_to_ground), T_ground_to_sample), cos_theta);
    float f = 1.0f / (1.0f + 5.0f * expf(-17.92f * cos_theta));
    v4 L_ms = V4((float)(0.02 * 0.217) * f, (float)(0.02 * 0.347) * f, (float)(0.02 * 0.594) * f, (float)(0.02 * 1.0) * f);
    return add4(L_ms, L_ground);
}
/* S:207-217: mat4x3 M, column-major: 4 columns of 3 */
static const float M_cols[4][3] = {
    {137.672389239975f, -8.632904716299537f, -1.7181567391931372f},
    {32.549094028629234f, 91.29801417199785f, -12.005406444382531f},
    {-38.91428392614275f, 34.31665471469816f, 29.89044807197628f},
    {8.572844237945445f, -11.103384660054624f, 117.47585277566478f}};
static v3 linear_srgb_from_spectral_samples(v4 L) {
    float o[3];
    for (int r = 0; r < 3; r++) o[r] = M_cols[0][r] * L.x + M_cols[1][r] * L.y + M_cols[2][r] * L.z + M_cols[3][r] * L.w;
    return V3(o[0], o[1], o[2]);
}
/* S:219-276 */
static v4 compute_inscattering(lut2d lut, const float sun[3], v3 ray_origin, v3 ray_dir, float t_d, v4 *transmittance) {
    v3 sun_dir = V3(-sun[0], -sun[2], sun[1]);                                   /* S:221-223 (.xzy, negate x,y) */
    float cos_theta = dot3(V3(-ray_dir.x, -ray_dir.y, -ray_dir.z), sun_dir);     /* S:224 */
    float molecular_phase = molecular_phase_function(cos_theta);
    float aerosol_phase = aerosol_phase_function(cos_theta);
    float dt = t_d / (float)IN_SCATTERING_STEPS;                                 /* S:229 */
    v4 L = V4(0, 0, 0, 0);
    *transmittance = V4(1, 1, 1, 1);
    const v4 irr = V4(sun_spectral_irradiance[0], sun_spectral_irradiance[1], sun_spectral_irradiance[2], sun_spectral_irradiance[3]);
    for (int i = 0; i < IN_SCATTERING_STEPS; ++i) {                              /* S:234 */
        float t = ((float)i + 0.5f) * dt;
        v3 x_t = add3(ray_origin, muls3(ray_dir, t));
        float distance_to_earth_center = length3(x_t);
        v3 zenith_dir = divs3(x_t, distance_to_earth_center);
        float altitude = distance_to_earth_center - EARTH_RADIUS;
        float normalized_altitude = altitude / ATMOSPHERE_THICKNESS;
        float sample_cos_theta = dot3(zenith_dir, sun_dir);                      /* S:243 */
        v4 aa, as, ma, msc, ext;
        get_atmosphere_collision_coefficients(altitude, &aa, &as, &ma, &msc, &ext);
        v4 transmittance_to_sun = transmittance_from_lut(lut, sample_cos_theta, normalized_altitude); /* S:254 */
        v4 ms = get_multiple_scattering(lut, sample_cos_theta, normalized_altitude, distance_to_earth_center);
        v4 S = mul4(irr, add4(mul4(msc, add4(muls4(transmittance_to_sun, molecular_phase), ms)),
                              mul4(as, add4(muls4(transmittance_to_sun, aerosol_phase), ms))));       /* S:261-263 */
        v4 step_transmittance = exp4(muls4(ext, -dt));                           /* S:265 */
        v4 S_int = div4(sub4(S, mul4(S, step_transmittance)), maxs4(ext, 1e-7f)); /* S:270 */
        L = add4(L, mul4(*transmittance, S_int));
        *transmittance = mul4(*transmittance, step_transmittance);
    }
    return L;
}
/* S:278-315 main().  (The `>` bounds test at S:281 lets 4 out-of-range rows through on the GPU where the
 * image store discards them; a w x h loop is the observable behaviour.) */
void csko_sky_lut(int w, int h, const float sun_dir[3], const uint16_t *trans, int tw, int th, uint16_t *out) {
    lut2d lut = {trans, tw, th};
    for (int py = 0; py < h; py++) for (int px = 0; px < w; px++) {
        float uvx = (float)px / (float)w, uvy = (float)py / (float)h;            /* S:284 */
        float azimuth = (float)(2.0 * S_PI) * uvx;                               /* S:286 */
        float l = uvy * 2.0f - 1.0f;                                             /* S:290 */
        float elev = l * l * signf(l) * (float)S_PI * 0.5f;                      /* S:291 */
        v3 ray_dir = V3(cosf(elev) * cosf(azimuth), cosf(elev) * sinf(azimuth), sinf(elev)); /* S:293-295 */
        v3 ray_origin = V3(0.0f, 0.0f, EYE_DISTANCE_TO_EARTH_CENTER);
        float atmos_dist = ray_sphere_intersection(ray_origin, ray_dir, ATMOSPHERE_RADIUS);
        float ground_dist = ray_sphere_intersection(ray_origin, ray_dir, EARTH_RADIUS);
        float t_d = (ground_dist < 0.0f) ? atmos_dist : ground_dist;             /* S:303-309 */
        v4 tr;
        v4 L = compute_inscattering(lut, sun_dir, ray_origin, ray_dir, t_d, &tr);
        v3 c = linear_srgb_from_spectral_samples(L);
        uint16_t *o = out + ((size_t)py * w + px) * 4;
        o[0] = csko_f2h(c.x); o[1] = csko_f2h(c.y); o[2] = csko_f2h(c.z); o[3] = csko_f2h(1.0f); /* S:313 */
    }
}

/* ==================================================================================== clouds.glsl */
#define g_radius 6000000.0f                           /* C:43 */
#define sky_b_radius 6001500.0f                       /* C:44 */
#define sky_t_radius 6004000.0f                       /* C:45 */
#define C_PI 3.141592f                                /* C:47 (truncated literal, kept) */

typedef struct {                                      /* C:18-40, offsets in floats */
    float texture_size[2], update_position[2], cloud_pos[2], detailed_pos[2], weather_pos[2], pad1[2];
    float ground_color[4], LIGHT_DIRECTION[3], LIGHT_ENERGY, LIGHT_COLOR[3], time, pad2, density, cloud_coverage, time_offset;
} cloud_params;

typedef struct { const csko_textures *tex; const cloud_params *pc; const uint16_t *sky; int sw, sh; int light_steps; } cloud_ctx;

/* C:49-57 */
static v3 getValFromSkyLUT(const cloud_ctx *c, v3 rayDir) {
    float phi = atan2f(rayDir.z, rayDir.x);
    float theta = asinf(rayDir.y);
    float uvx = (phi / C_PI * 0.5f + 0.5f);
    float uvy = sqrtf(fabsf(theta) / (C_PI * 0.5f)) * signf(theta) * 0.5f + 0.5f;
    v4 t = sample_rgba16f_clamp(c->sky, c->sw, c->sh, uvx, uvy);
    return V3(t.x, t.y, t.z);
}
/* C:60-64 */
static float hash3(v3 p) {
    p = V3(fractf(p.x * 0.3183099f + 0.1f), fractf(p.y * 0.3183099f + 0.1f), fractf(p.z * 0.3183099f + 0.1f));
    p = muls3(p, 17.0f);
    return fractf(p.x * p.y * p.z * (p.x + p.y + p.z));
}
/* C:67-69 */
static float remap(float v, float omin, float omax, float nmin, float nmax) {
    return nmin + (((v - omin) / (omax - omin)) * (nmax - nmin));
}
/* C:72-75 */
static float henyey_greenstein(float cos_theta, float g) {
    const float k = 0.0795774715459f;
    return k * (1.0f - g * g) / (powf(1.0f + g * g - 2.0f * g * cos_theta, 1.5f));
}
/* C:77-80 */
static float GetHeightFractionForPoint(float inPosition) {
    float height_fraction = (inPosition - sky_b_radius) / (sky_t_radius - sky_b_radius);
    return clampf(height_fraction, 0.0f, 1.0f);
}
/* C:82-90 */
static v4 mixGradients(float cloudType) {
    const v4 STRATUS = {0.02f, 0.05f, 0.09f, 0.11f}, STRATOCUMULUS = {0.02f, 0.2f, 0.48f, 0.625f}, CUMULUS = {0.01f, 0.0625f, 0.78f, 1.0f};
    float stratus = 1.0f - clampf(cloudType * 2.0f, 0.0f, 1.0f);
    float stratocumulus = 1.0f - fabsf(cloudType - 0.5f) * 2.0f;
    float cumulus = clampf(cloudType - 0.5f, 0.0f, 1.0f) * 2.0f;
    return add4(add4(muls4(STRATUS, stratus), muls4(STRATOCUMULUS, stratocumulus)), muls4(CUMULUS, cumulus));
}
/* C:92-95 */
static float densityHeightGradient(float heightFrac, float cloudType) {
    v4 g = mixGradients(cloudType);
    return smoothstepf(g.x, g.y, heightFrac) - smoothstepf(g.z, g.w, heightFrac);
}
/* C:97-105 */
static float intersectSphere(v3 pos, v3 dir, float r) {
    float a = dot3(dir, dir);
    float b = 2.0f * dot3(dir, pos);
    float c = dot3(pos, pos) - (r * r);
    float d = sqrtf((b * b) - 4.0f * a * c);
    float p = -b - d;
    float p2 = -b + d;
    return fmaxf(p, p2) / (2.0f * a);
}
/* C:109-137 */
static float density(const cloud_ctx *c, v3 pip, v3 weather, float mip) {
    const cloud_params *P = c->pc;
    v3 p = pip;
    float height_fraction = GetHeightFractionForPoint(length3(p));
    p.x += 20.0f * P->cloud_pos[0] * 0.6f;                                       /* C:114 (p.xz += vec2) */
    p.z += 20.0f * P->cloud_pos[1] * 0.6f;
    float n[4];
    sample3d_repeat(c->tex->large_rgba8, 128, c->tex->large_levels, 4, mip - 2.0f, muls3(p, 0.00008f), n); /* C:117 */
    float fbm = n[1] * 0.625f + n[2] * 0.25f + n[3] * 0.125f;                    /* C:118 */
    float g = densityHeightGradient(height_fraction, weather.x);                 /* C:121 */
    float base_cloud = remap(n[0], -(1.0f - fbm), 1.0f, 0.0f, 1.0f);             /* C:122 */
    float weather_coverage = P->cloud_coverage * weather.z;                      /* C:123 */
    base_cloud = remap(base_cloud * g, 1.0f - (weather_coverage), 1.0f, 0.0f, 1.0f); /* C:124 */
    base_cloud *= weather_coverage;                                              /* C:125 */
    p.x -= P->detailed_pos[0] * 40.0f;                                           /* C:128 */
    p.z -= P->detailed_pos[1] * 40.0f;
    p.y -= P->time * 40.0f;                                                      /* C:129 */
    float hn[4];
    sample3d_repeat(c->tex->small_rgb8, 32, c->tex->small_levels, 3, mip, muls3(p, 0.001f), hn); /* C:132 */
    float hfbm = hn[0] * 0.625f + hn[1] * 0.25f + hn[2] * 0.125f;                /* C:133 */
    hfbm = mixf(hfbm, 1.0f - hfbm, clampf(height_fraction * 4.0f, 0.0f, 1.0f));  /* C:134 */
    base_cloud = remap(base_cloud, hfbm * 0.4f * height_fraction, 1.0f, 0.0f, 1.0f); /* C:135 */
    return powf(clampf(base_cloud, 0.0f, 1.0f), (1.0f - height_fraction) * 0.8f + 0.5f); /* C:136 */
}

static const float RANDOM_VECTORS[6][3] = {                                      /* C:140 */
    {0.38051305f, 0.92453449f, -0.02111345f}, {-0.50625799f, -0.03590792f, -0.86163418f},
    {-0.32509218f, -0.94557439f, 0.01428793f}, {0.09026238f, -0.27376545f, 0.95755165f},
    {0.28128598f, 0.42443639f, -0.86065785f}, {-0.16852403f, 0.14748697f, 0.97460106f}};

/* C:139-215.  `end` (unused by the shader) is dropped.  light_steps generalises the literal 6 at C:186. */
static v4 march(const cloud_ctx *c, v3 pos, v3 dir, int depth, uint64_t *incloud) {
    const cloud_params *P = c->pc;
    float ss = length3(dir);                                                     /* C:143 */
    dir = normalize3(dir);                                                       /* C:144 */
    v3 p = add3(pos, muls3(muls3(dir, hash3(muls3(pos, 10.0f))), ss));           /* C:145 */
    const float t_dist = sky_t_radius - sky_b_radius;                            /* C:148 */
    float lss = (t_dist / 64.0f);                                                /* C:149 */
    v3 ldir = normalize3(V3(P->LIGHT_DIRECTION[0], P->LIGHT_DIRECTION[1], P->LIGHT_DIRECTION[2])); /* C:150 */
    float t = 1.0f, T = 1.0f, alpha = 0.0f;
    v3 L = V3(0, 0, 0);
    float costheta = dot3(ldir, dir);                                            /* C:158 */
    float phase = fmaxf(fmaxf(henyey_greenstein(costheta, 0.6f), henyey_greenstein(costheta, (0.4f - 1.4f * ldir.y))),
                        henyey_greenstein(costheta, -0.2f));                     /* C:160 */
    v3 LD = V3(P->LIGHT_DIRECTION[0], P->LIGHT_DIRECTION[1], P->LIGHT_DIRECTION[2]);
    v3 atmosphere_sun = mul3(muls3(muls3(getValFromSkyLUT(c, LD), 0.1f), P->LIGHT_ENERGY),
                             V3(P->LIGHT_COLOR[0], P->LIGHT_COLOR[1], P->LIGHT_COLOR[2]));           /* C:163 */
    v3 atmosphere_ambient = muls3(getValFromSkyLUT(c, normalize3(V3(1.0f, 1.0f, 0.0f))), 0.05f);   /* C:164 */
    { float l = length3(atmosphere_ambient); atmosphere_ambient = mix3(atmosphere_ambient, V3(l, l, l), 0.5f); } /* C:165 */
    v3 atmosphere_ground = muls3(muls3(getValFromSkyLUT(c, normalize3(V3(1.0f, -1.0f, 0.0f))), 5.0f), 0.05f); /* C:166 */
    { float l = length3(atmosphere_ground);
      atmosphere_ground = mix3(atmosphere_ground, mul3(V3(P->ground_color[0], P->ground_color[1], P->ground_color[2]), V3(l, l, l)), 0.5f); } /* C:167 */
    const float weather_scale = 0.00006f;                                        /* C:169 */
    float wpx = P->weather_pos[0], wpy = P->weather_pos[1];                      /* C:170 */

    for (int i = 0; i < depth; i++) {                                            /* C:172 */
        p = add3(p, muls3(dir, ss));                                             /* C:173 */
        v3 weather_sample = sample_weather(c->tex->weather_rgb8, p.x * weather_scale + 0.5f + wpx, p.z * weather_scale + 0.5f + wpy); /* C:174 */
        float height_fraction = GetHeightFractionForPoint(length3(p));           /* C:175 */
        t = density(c, p, weather_sample, 0.0f);                                 /* C:177 */
        float dt = expf(-P->density * t * ss);                                   /* C:178 */
        v3 lp = p;
        float lt = 1.0f, cd = 0.0f;
        if (t > 0.0f) {                                                          /* C:184 */
            (*incloud)++;
            float lheight_fraction = 0.0f;
            for (int j = 0; j < c->light_steps; j++) {                           /* C:186 */
                v3 rv = V3(RANDOM_VECTORS[j][0], RANDOM_VECTORS[j][1], RANDOM_VECTORS[j][2]);
                lp = add3(lp, muls3(add3(ldir, muls3(rv, (float)j)), lss));      /* C:187 */
                lheight_fraction = GetHeightFractionForPoint(length3(lp));       /* C:188 */
                v3 lweather = sample_weather(c->tex->weather_rgb8, lp.x * weather_scale + 0.5f + wpx, lp.z * weather_scale + 0.5f + wpy); /* C:189 */
                lt = density(c, lp, lweather, (float)j);                         /* C:190 */
                cd += lt;                                                        /* C:191 */
            }
            lp = add3(p, muls3(muls3(ldir, 18.0f), lss));                        /* C:195 */
            lheight_fraction = GetHeightFractionForPoint(length3(lp));           /* C:196 */
            v3 lweather = sample_weather(c->tex->weather_rgb8, lp.x * weather_scale + 0.5f, lp.z * weather_scale + 0.5f); /* C:197 (no weather_pos) */
            lt = powf(density(c, lp, lweather, 5.0f), (1.0f - lheight_fraction) * 0.8f + 0.5f); /* C:198 */
            cd += lt;                                                            /* C:199 */
            float beers = expf(-P->density * cd * lss * 3.0f);                   /* C:202 */
            float powder_sugar_effect = 1.0f - expf(-P->density * cd * lss * 3.0f * 2.0f); /* C:203 */
            float beers_total = 2.0f * beers * powder_sugar_effect;              /* C:204 */
            v3 ambient = mix3(atmosphere_ground, atmosphere_ambient, smoothstepf(0.0f, 1.0f, height_fraction)); /* C:206 */
            alpha += (1.0f - dt) * (1.0f - alpha);                               /* C:207 */
            v3 radiance = muls3(add3(ambient, muls3(muls3(atmosphere_sun, beers_total), phase)), t); /* C:208: (beers_total*atmosphere_sun)*phase */
            float inv = fmaxf(0.0000001f, t);
            L = add3(L, divs3(muls3(sub3(radiance, muls3(radiance, dt)), T), inv)); /* C:209: T*(r - r*dt)/max(1e-7,t) */
            T *= dt;                                                             /* C:210 */
        }
        (void)lt;
    }
    alpha = clampf(alpha, 0.0f, 1.0f);                                           /* C:213 */
    return V4(L.x, L.y, L.z, alpha);
}

/* C:239-244 */
static void oct_wrap(float vx, float vy, float *ox, float *oy) {
    float sx = vx >= 0.0f ? 1.0f : -1.0f, sy = vy >= 0.0f ? 1.0f : -1.0f;
    *ox = (1.0f - fabsf(vy)) * sx; *oy = (1.0f - fabsf(vx)) * sy;
}
/* C:248-256 */
static v3 oct_to_vec3(float ex, float ey) {
    v3 n;
    n.x = (ex - ey);
    n.y = (ex + ey) - 1.0f;
    n.z = 1.0f - fabsf(n.x) - fabsf(n.y);
    if (!(n.z >= 0.0f)) { float a, b; oct_wrap(n.x, n.y, &a, &b); n.x = a; n.y = b; }
    return normalize3(n);
}
/* C:218-237 */
static v4 sky(const cloud_ctx *c, v3 dir, int steps_i, uint64_t *incloud, int *marched) {
    v4 col = V4(0, 0, 0, 0);
    if (dir.y > 0.0f) {
        v3 camPos = V3(0.0f, g_radius, 0.0f);
        v3 start = add3(camPos, muls3(dir, intersectSphere(camPos, dir, sky_b_radius)));
        v3 end = add3(camPos, muls3(dir, intersectSphere(camPos, dir, sky_t_radius)));
        float shelldist = (length3(sub3(end, start)));
        float steps = (float)steps_i;                                            /* C:228 (128.0) */
        v3 raystep = divs3(muls3(dir, shelldist), steps);                        /* C:230 */
        col = march(c, start, raystep, (int)steps, incloud);                     /* C:231 */
        *marched = 1;
    }
    return col;
}
static v3 pixel_dir(const cloud_params *P, int gx, int gy) {
    int px = gx + (int)P->update_position[0], py = gy + (int)P->update_position[1];      /* C:260 */
    float uvx = (float)px / P->texture_size[0], uvy = (float)py / P->texture_size[1];    /* C:261 */
    v3 o = oct_to_vec3(uvx, uvy);
    return V3(o.x, o.z, o.y);                                                            /* C:262 (.xzy) */
}

/* C:258-266 main() over a pixel rectangle */
void csko_clouds(const csko_textures *tex, const float params[28], int primary_steps, int light_steps,
                 const uint16_t *sky_lut, int sw, int sh, int gx0, int gy0, int w, int h,
                 uint16_t *out, size_t pitch, int nthreads, csko_stats *stats) {
    cloud_params P; memcpy(&P, params, sizeof(P));
    cloud_ctx c = {tex, &P, sky_lut, sw, sh, light_steps};
    uint64_t incloud_total = 0, marched_total = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : incloud_total, marched_total)
#endif
    for (int ry = 0; ry < h; ry++) {
        for (int rx = 0; rx < w; rx++) {
            int gx = gx0 + rx, gy = gy0 + ry;
            v3 dir = pixel_dir(&P, gx, gy);
            uint64_t ic = 0; int m = 0;
            v4 col = sky(&c, dir, primary_steps, &ic, &m);
            incloud_total += ic; marched_total += (uint64_t)m;
            uint16_t *o = (uint16_t *)((char *)out + (size_t)ry * pitch) + (size_t)rx * 4;
            o[0] = csko_f2h(col.x); o[1] = csko_f2h(col.y); o[2] = csko_f2h(col.z); o[3] = csko_f2h(col.w); /* C:264 */
        }
    }
    (void)nthreads;
    if (stats) {
        stats->rays = (uint64_t)w * (uint64_t)h; stats->rays_marched = marched_total;
        stats->primary_samples = marched_total * (uint64_t)primary_steps; stats->incloud_samples = incloud_total;
    }
}


/* Same as csko_clouds but over the band set of the product's csky_bands (compact output rows): used by the
 * cpu_baseline leg of bench.py to time a bounded, evenly spread sample of the frame on all host cores
 * (OpenMP over (row, 64-column chunk) work items, dynamic schedule). */
void csko_clouds_bands(const csko_textures *tex, const float params[28], int primary_steps, int light_steps,
                       const uint16_t *sky_lut, int sw, int sh, int tile_w, int band_rows, int first_band, int band_stride,
                       int n_bands, uint16_t *out, int nthreads, csko_stats *stats) {
    cloud_params P; memcpy(&P, params, sizeof(P));
    cloud_ctx c = {tex, &P, sky_lut, sw, sh, light_steps};
    uint64_t incloud_total = 0, marched_total = 0;
    const int rows = n_bands * band_rows, chunks = (tile_w + 63) / 64;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : incloud_total, marched_total)
#endif
    for (int item = 0; item < rows * chunks; item++) {
        const int lr = item / chunks, ch = item - lr * chunks;
        const int band = lr / band_rows, rib = lr - band * band_rows;
        const int gy = (first_band + band * band_stride) * band_rows + rib;
        const int x1 = (ch * 64 + 64 < tile_w) ? ch * 64 + 64 : tile_w;
        for (int gx = ch * 64; gx < x1; gx++) {
            v3 dir = pixel_dir(&P, gx, gy);
            uint64_t ic = 0; int m = 0;
            v4 col = sky(&c, dir, primary_steps, &ic, &m);
            incloud_total += ic; marched_total += (uint64_t)m;
            uint16_t *o = out + ((size_t)lr * tile_w + gx) * 4;
            o[0] = csko_f2h(col.x); o[1] = csko_f2h(col.y); o[2] = csko_f2h(col.z); o[3] = csko_f2h(col.w);
        }
    }
    (void)nthreads;
    if (stats) {
        stats->rays = (uint64_t)tile_w * (uint64_t)rows; stats->rays_marched = marched_total;
        stats->primary_samples = marched_total * (uint64_t)primary_steps; stats->incloud_samples = incloud_total;
    }
}

/* ==================================================================================== clouds.gdshader (G:line)
 * "Next" row 1: the raster sky() pass that composites the cloud texture over the atmosphere.  EYEDIR is
 * supplied by Godot per screen pixel; this restatement evaluates sky() for the pixels of an equirectangular panorama:
 * u = (i+0.5)/W -> azimuth (2u-1)*pi, v = (j+0.5)/H -> elevation (0.5-v)*pi, EYEDIR = (cos e cos a, sin e, cos e sin a)
 * (build-side mapping, y up like Godot).  G_PI is the Godot shading language's built-in PI (full precision). */
#define G_PI 3.14159265358979323846f
static v4 tap16_clamp(const uint16_t *t, int w, int h, float sx, float sy) { return sample_rgba16f_clamp(t, w, h, sx, sy); }

/* G:22-32 vec3_to_oct */
static void vec3_to_oct(v3 e, float *ox, float *oy) {
    float d = fabsf(e.x) + fabsf(e.y) + fabsf(e.z);
    e = divs3(e, d);
    if (!(e.z >= 0.0f)) { float a, b; oct_wrap(e.x, e.y, &a, &b); e.x = a; e.y = b; }
    float ny = e.y * 0.5f + 0.5f;
    float nx = e.x * 0.5f + ny;
    ny = e.x * -0.5f + ny;
    *ox = nx; *oy = ny;
}
typedef struct { const uint16_t *cf, *ct; int cw, ch; const uint16_t *sf, *st; int sw, sh; const uint16_t *tr; int tw, th;
                 float blend, sun_disk_scale; v3 sun; } comp_ctx;
/* G:34-45 */
static v3 g_getValFromSkyLUT(const comp_ctx *c, v3 rayDir) {
    float phi = atan2f(rayDir.z, rayDir.x);
    float theta = asinf(rayDir.y);
    float ux = (phi / G_PI * 0.5f + 0.5f);
    float uy = sqrtf(fabsf(theta) / (G_PI * 0.5f)) * signf(theta) * 0.5f + 0.5f;
    v4 a = tap16_clamp(c->sf, c->sw, c->sh, ux, uy), b = tap16_clamp(c->st, c->sw, c->sh, ux, uy);
    v3 m = mix3(V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), c->blend);
    return divs3(m, 50.0f);
}
/* G:48-59 */
static v3 sunWithBloom(const comp_ctx *c, v3 rayDir, v3 sunDir) {
    float sunSolidAngle = c->sun_disk_scale * 0.53f * G_PI / 180.0f;
    float minSunCosTheta = cosf(sunSolidAngle);
    float cosTheta = dot3(rayDir, sunDir);
    if (cosTheta >= minSunCosTheta) return V3(1.0f, 1.0f, 1.0f);
    float offset = minSunCosTheta - cosTheta;
    float gaussianBloom = expf(-offset * 50000.0f) * 0.5f;
    float invBloom = 1.0f / (0.02f + offset * 300.0f) * 0.01f;
    float s = gaussianBloom + invBloom;
    return V3(s, s, s);
}
/* G:61-70 */
static float rayIntersectSphere(v3 ro, v3 rd, float rad) {
    float b = dot3(ro, rd);
    float c = dot3(ro, ro) - rad * rad;
    if (c > 0.0f && b > 0.0f) return -1.0f;
    float discr = b * b - c;
    if (discr < 0.0f) return -1.0f;
    if (discr > b * b) return (-b + sqrtf(discr));
    return -b - sqrtf(discr);
}
#define groundRadiusMM 6.360f        /* G:72 */
#define atmosphereRadiusMM 6.460f    /* G:73 */
/* G:77-85 (tLUTRes == bufferRes, G:75,101) */
static v3 getValFromTLUT(const comp_ctx *c, v3 pos, v3 sunDir) {
    float height = length3(pos);
    v3 up = divs3(pos, height);
    float sunCosZenithAngle = dot3(up, sunDir);
    float ux = 256.0f * clampf(0.5f + 0.5f * sunCosZenithAngle, 0.0f, 1.0f);
    /* atmosphereRadiusMM - groundRadiusMM is a constant expression: glslang folds it in double (0.1), it is NOT 6.46f - 6.36f = 0.0999999 (found by executing
     * the shader text, oracle/glsl_exec: 3 halfs of the demo-scene panorama were 1 ulp off) */
    float uy = 64.0f * fmaxf(0.0f, fminf(1.0f, (height - groundRadiusMM) / (float)(6.460 - 6.360)));
    ux /= 256.0f; uy /= 64.0f;
    v4 t = tap16_clamp(c->tr, c->tw, c->th, ux, uy);
    return V3(t.x, t.y, t.z);
}
static float smoothstep1(float e0, float e1, float x) { return smoothstepf(e0, e1, x); }
/* G:87-103 */
static v3 get_atmo(const comp_ctx *c, v3 dir) {
    v3 col = g_getValFromSkyLUT(c, dir);
    v3 sunLum = sunWithBloom(c, dir, c->sun);
    sunLum = V3(smoothstep1(0.002f, 1.0f, sunLum.x), smoothstep1(0.002f, 1.0f, sunLum.y), smoothstep1(0.002f, 1.0f, sunLum.z));
    const v3 viewPos = V3(0.0f, (float)(6.360 + 0.0002), 0.0f);                 /* G:74: a constant expression, folded in double then narrowed */
    if (length3(sunLum) > 0.0f) {
        if (rayIntersectSphere(viewPos, dir, groundRadiusMM) >= 0.0f) sunLum = muls3(sunLum, 0.0f);
        else sunLum = mul3(sunLum, getValFromTLUT(c, viewPos, c->sun));
    }
    return add3(col, sunLum);
}
/* G:105-116 sky() for one EYEDIR */
static v3 sky_composite(const comp_ctx *c, v3 EYEDIR) {
    v3 norm = EYEDIR;
    norm.y = fmaxf(0.0f, norm.y);
    norm = normalize3(norm);
    float ox, oy;
    vec3_to_oct(V3(norm.x, norm.z, norm.y), &ox, &oy);                           /* G:110 norm.xz = vec3_to_oct(norm.xzy) */
    v4 bf = tap16_clamp(c->cf, c->cw, c->ch, ox, oy), bt = tap16_clamp(c->ct, c->cw, c->ch, ox, oy);
    v4 clouds = V4(mixf(bf.x, bt.x, c->blend), mixf(bf.y, bt.y, c->blend), mixf(bf.z, bt.z, c->blend), mixf(bf.w, bt.w, c->blend));
    v3 background = get_atmo(c, EYEDIR);
    v3 COLOR = add3(muls3(background, 1.0f - clouds.w), V3(clouds.x, clouds.y, clouds.z));
    float k = smoothstep1(0.6f, 1.0f, 1.0f - EYEDIR.y);
    v3 a = V3(clampf(COLOR.x, 0.0f, 100.0f), clampf(COLOR.y, 0.0f, 100.0f), clampf(COLOR.z, 0.0f, 100.0f));
    v3 b = V3(clampf(background.x, 0.0f, 100.0f), clampf(background.y, 0.0f, 100.0f), clampf(background.z, 0.0f, 100.0f));
    return mix3(a, b, k);
}
/* EYEDIR of pixel (i, j) of the out_w x out_h panorama (the build-side mapping described above); exported so that oracle/glsl_exec drives the
 * reference's own sky() with bit-identical directions */
void csko_panorama_eyedir(int out_w, int out_h, int i, int j, float eye[3]) {
    float u = ((float)i + 0.5f) / (float)out_w, v = ((float)j + 0.5f) / (float)out_h;
    float az = (u * 2.0f - 1.0f) * G_PI, el = (0.5f - v) * G_PI;
    eye[0] = cosf(el) * cosf(az); eye[1] = sinf(el); eye[2] = cosf(el) * sinf(az);
}
void csko_composite(int out_w, int out_h, const uint16_t *cloud_from, const uint16_t *cloud_to, int cw, int ch, const uint16_t *sky_from,
                    const uint16_t *sky_to, int sw, int sh, const uint16_t *trans, int tw, int th, float blend_amount,
                    float sun_disk_scale, const float light_dir[3], uint16_t *out_rgba16f) {
    comp_ctx c = {cloud_from, cloud_to, cw, ch, sky_from, sky_to, sw, sh, trans, tw, th, blend_amount, sun_disk_scale,
                  V3(light_dir[0], light_dir[1], light_dir[2])};
    for (int j = 0; j < out_h; j++) for (int i = 0; i < out_w; i++) {
        float e[3];
        csko_panorama_eyedir(out_w, out_h, i, j, e);
        v3 col = sky_composite(&c, V3(e[0], e[1], e[2]));
        uint16_t *o = out_rgba16f + ((size_t)j * out_w + i) * 4;
        o[0] = csko_f2h(col.x); o[1] = csko_f2h(col.y); o[2] = csko_f2h(col.z); o[3] = csko_f2h(1.0f);
    }
}

/* The same shader per SCREEN pixel of a perspective camera: the engine supplies EYEDIR per screen pixel (clouds.gdshader:105 `void sky()` reads
 * EYEDIR); pixel (i,j) -> NDC -> view ray (x tan(fov/2) aspect, y tan(fov/2), -1) -> world through the camera basis (column-major) -> normalise. */
void csko_composite_view(int out_w, int out_h, const float basis[9], float fov_y_degrees, const uint16_t *cloud_from, const uint16_t *cloud_to, int cw, int ch,
                         const uint16_t *sky_from, const uint16_t *sky_to, int sw, int sh, const uint16_t *trans, int tw, int th, float blend_amount,
                         float sun_disk_scale, const float light_dir[3], uint16_t *out_rgba16f) {
    comp_ctx c = {cloud_from, cloud_to, cw, ch, sky_from, sky_to, sw, sh, trans, tw, th, blend_amount, sun_disk_scale,
                  V3(light_dir[0], light_dir[1], light_dir[2])};
    const float t = tanf(fov_y_degrees * 0.5f * G_PI / 180.0f), aspect = (float)out_w / (float)out_h;
    for (int j = 0; j < out_h; j++) for (int i = 0; i < out_w; i++) {
        float u = ((float)i + 0.5f) / (float)out_w, v = ((float)j + 0.5f) / (float)out_h;
        float vx = (u * 2.0f - 1.0f) * t * aspect, vy = (1.0f - v * 2.0f) * t, vz = -1.0f;
        float wx = basis[0] * vx + basis[3] * vy + basis[6] * vz, wy = basis[1] * vx + basis[4] * vy + basis[7] * vz, wz = basis[2] * vx + basis[5] * vy + basis[8] * vz;
        float l = sqrtf(wx * wx + wy * wy + wz * wz);
        v3 col = sky_composite(&c, V3(wx / l, wy / l, wz / l));
        uint16_t *o = out_rgba16f + ((size_t)j * out_w + i) * 4;
        o[0] = csko_f2h(col.x); o[1] = csko_f2h(col.y); o[2] = csko_f2h(col.z); o[3] = csko_f2h(1.0f);
    }
}

/* ------------------------------------------------------------------ probes for structural tests */
/* Sampler / store probes: the texel-fetch semantics of this file exported one tap at a time, so that oracle/glsl_exec (the reference's
 * shader text executed under a C++ GLSL-subset shim, build container only) binds texture()/textureLod() to exactly these functions. */
void csko_tap3d_repeat(const uint8_t *chain, int n0, int levels, int ch, float lod, const float s[3], float out[4]) {
    out[0] = out[1] = out[2] = 0.0f; out[3] = 1.0f;
    sample3d_repeat(chain, n0, levels, ch, lod, V3(s[0], s[1], s[2]), out);
}
void csko_tap_weather(const uint8_t *weather_rgb8, float sx, float sy, float out[3]) {
    v3 r = sample_weather(weather_rgb8, sx, sy); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void csko_tap_rgba16f_clamp(const uint16_t *t, int w, int h, float sx, float sy, float out[4]) {
    v4 r = sample_rgba16f_clamp(t, w, h, sx, sy); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
float csko_hash_probe(float px, float py, float pz) { return hash3(muls3(V3(px, py, pz), 10.0f)); }
void csko_pixel_dir(const float params[28], int px, int py, float dir[3]) {
    cloud_params P; memcpy(&P, params, sizeof(P));
    v3 d = pixel_dir(&P, px, py); dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}
void csko_sky_lut_lookup(const uint16_t *sky_lut, int sw, int sh, const float dir[3], float rgb[3]) {
    cloud_ctx c = {0, 0, sky_lut, sw, sh, 6};
    v3 r = getValFromSkyLUT(&c, V3(dir[0], dir[1], dir[2])); rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}
float csko_density_probe(const csko_textures *tex, const float params[28], const float p[3], const float weather[3], float mip) {
    cloud_params P; memcpy(&P, params, sizeof(P));
    cloud_ctx c = {tex, &P, 0, 0, 0, 6};
    return density(&c, V3(p[0], p[1], p[2]), V3(weather[0], weather[1], weather[2]), mip);
}
int csko_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
