// lut_core.h -- per-texel bodies of the two atmosphere LUT kernels (transmittance-lut.glsl, sky-lut.glsl),
// one texel per lane.  Host+device so tests/hostsim can run them on a CPU; the product only instantiates
// them in lut_kernels.hip.  These kernels are tiny (16 384 and 20 000 lanes), so everything is compiled with
// FP contraction off and correctly rounded exp/log/pow/sin/cos (below): parity (<= 1 fp16 ulp) matters, speed does not.
// Citations: T: = cloud_sky/transmittance-lut.glsl, S: = cloud_sky/sky-lut.glsl.  Units: km.
#pragma once
#include "csky_common.h"

namespace csky {
#pragma clang fp contract(off)

constexpr float EARTH_RADIUS = 6371.0f;          // T:50  S:58
constexpr float ATMOSPHERE_THICKNESS = 100.0f;   // T:51  S:59
constexpr float ATMOSPHERE_RADIUS = 6471.0f;     // T:52  S:60
constexpr int TRANSMITTANCE_STEPS = 40;          // T:45
constexpr int IN_SCATTERING_STEPS = 30;          // S:53
constexpr double LUT_PI = 3.14159265358979323846;  // S:44 (glslang folds constants in double, then narrows)

struct F4 { float x, y, z, w; };
CSKY_HD F4 f4(float x, float y, float z, float w) { F4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
CSKY_HD F4 operator+(F4 a, F4 b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
CSKY_HD F4 operator-(F4 a, F4 b) { return f4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
CSKY_HD F4 operator*(F4 a, F4 b) { return f4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
CSKY_HD F4 operator/(F4 a, F4 b) { return f4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
CSKY_HD F4 operator*(F4 a, float s) { return f4(a.x * s, a.y * s, a.z * s, a.w * s); }
// Transcendentals of the LUT kernels.  The oracle's glibc expf/logf/powf/sinf/cosf are correctly rounded in all but ~1e-5 of the
// cases; OCML's float versions are 1-ulp functions, and the Hillaire integration S - S*exp(-dt*ext) (S:270) cancels and amplifies
// that ulp to 3-4 fp16 ulp of the sky LUT.  On the device they are therefore evaluated in double and rounded once (correctly
// rounded except ~1e-8 of the cases): these kernels are 16 384 + 20 000 x 30 lanes, the fp64 rate is irrelevant (25 us).
// Measured on MI355X over 75 suns (round 3's LUT timing script, retired; the parity side is tests/test_gpu_round2.py's sun sweep): all through double (3, the default) transmittance LUT bit-identical, sky LUT worst 1 ulp,
// 0.09 % of texels off, 33.0 us per back-to-back launch; every exp only (2) or the step exps only (1) worst 3 ulp, 0.8 % off, 27 us; none (0)
// worst 3 ulp, 1.5 % off, 25.6 us: the log / pow / sin / cos matter as much as the exps, and exactness costs 7 us of a kernel that runs on the
// prologue stream beside the march.
#ifndef CSKY_LUT_CR
#define CSKY_LUT_CR 3
#endif
#if defined(__HIP_DEVICE_COMPILE__)
CSKY_HD float exp_step(float x) { return CSKY_LUT_CR >= 1 ? (float)exp((double)x) : expf(x); }
CSKY_HD float exp_cr(float x) { return CSKY_LUT_CR >= 2 ? (float)exp((double)x) : expf(x); }
CSKY_HD float log_cr(float x) { return CSKY_LUT_CR >= 3 ? (float)log((double)x) : logf(x); }
CSKY_HD float pow_cr(float x, float y) { return CSKY_LUT_CR >= 3 ? (float)pow((double)x, (double)y) : powf(x, y); }
CSKY_HD float sin_cr(float x) { return CSKY_LUT_CR >= 3 ? (float)sin((double)x) : sinf(x); }
CSKY_HD float cos_cr(float x) { return CSKY_LUT_CR >= 3 ? (float)cos((double)x) : cosf(x); }
#else
CSKY_HD float exp_step(float x) { return expf(x); }
CSKY_HD float exp_cr(float x) { return expf(x); }
CSKY_HD float log_cr(float x) { return logf(x); }
CSKY_HD float pow_cr(float x, float y) { return powf(x, y); }
CSKY_HD float sin_cr(float x) { return sinf(x); }
CSKY_HD float cos_cr(float x) { return cosf(x); }
#endif
CSKY_HD F4 exp4(F4 a) { return f4(exp_step(a.x), exp_step(a.y), exp_step(a.z), exp_step(a.w)); }

// T:89-98 / S:100-109
CSKY_HD float ray_sphere_intersection(float ox, float oy, float oz, float dx, float dy, float dz, float radius) {
    const float b = ox * dx + oy * dy + oz * dz;
    const float c = (ox * ox + oy * oy + oz * oz) - radius * radius;
    if (c > 0.0f && b > 0.0f) return -1.0f;
    const float d = b * b - c;
    if (d < 0.0f) return -1.0f;
    if (d > b * b) return (-b + sqrtf(d));
    return (-b - sqrtf(d));
}

struct Coeffs { F4 aerosol_scattering, molecular_scattering, extinction; };
// T:104-145 / S:132-135,170-202
CSKY_HD Coeffs atmosphere_collision_coefficients(float h) {
    h = fmaxf(h, 0.0f);
    const float aerosol_density = 1.3681e20f * (exp_cr(-h / 0.73f) + (float)(2e6 / 1.3681e20));
    const F4 aa = f4(2.8722e-24f, 4.6168e-24f, 7.9706e-24f, 1.3578e-23f) * aerosol_density;
    const F4 as = f4(1.5908e-22f, 1.7711e-22f, 2.0942e-22f, 2.4033e-22f) * aerosol_density;
    const float h2 = h + 1e-4f;
    const float t = log_cr(h2) - 3.22261f;
    const float ozone_density = 3.78547397e20f * (1.0f / h2) * exp_cr(-t * t * 5.55555555f);
    const F4 ma = f4((float)(3.472e-21 * 1e-4 * 350.0), (float)(3.914e-21 * 1e-4 * 350.0), (float)(1.349e-21 * 1e-4 * 350.0),
                     (float)(11.03e-23 * 1e-4 * 350.0)) * ozone_density;
    const F4 ms = f4(6.605e-3f, 1.067e-2f, 1.842e-2f, 3.156e-2f) * exp_cr(-0.07771971f * pow_cr(h, 1.16364243f));
    Coeffs c; c.aerosol_scattering = as; c.molecular_scattering = ms; c.extinction = aa + as + ma + ms;
    return c;
}

// T:157-196 main() for texel (px,py) of a w x h LUT, split like the sky LUT below so that the 40 optical-depth steps of one texel can be
// evaluated by 40 lanes in parallel: every step's extinction * dt is independent of the others, only the running sum (T:191) is
// sequential, and it is replayed in the reference's order (bit-identical to the one-lane form).
struct TransRay { float sdx, sdz, d, dt; };
CSKY_HD TransRay transmittance_ray(int px, int py, float w, float h) {
    TransRay r;
    const float uvx = (float)px / w, uvy = (float)py / h;
    const float c = uvx * 2.0f - 1.0f;
    r.sdx = -sqrtf(1.0f - c * c); r.sdz = c;
    r.d = EARTH_RADIUS * (1.0f - uvy) + ATMOSPHERE_RADIUS * uvy;  // mix()
    const float t_d = ray_sphere_intersection(0.0f, 0.0f, r.d, r.sdx, 0.0f, r.sdz, ATMOSPHERE_RADIUS);
    r.dt = t_d / (float)TRANSMITTANCE_STEPS;
    return r;
}
CSKY_HD F4 transmittance_step(const TransRay& r, int i) {            // one term of the sum at T:186-192
    const float t = ((float)i + 0.5f) * r.dt;
    const float x = 0.0f + r.sdx * t, y = 0.0f + 0.0f * t, z = r.d + r.sdz * t;
    const float altitude = sqrtf(x * x + y * y + z * z) - EARTH_RADIUS;
    return atmosphere_collision_coefficients(altitude).extinction * r.dt;
}
CSKY_HD F4 transmittance_finish(const F4& result) { return exp4(f4(-result.x, -result.y, -result.z, -result.w)); }   // T:194
// the whole texel on one lane (host-compiled unit test; the kernel spreads the steps over lanes)
CSKY_HD F4 transmittance_texel(int px, int py, float w, float h) {
    const TransRay r = transmittance_ray(px, py, w, h);
    F4 result = f4(0, 0, 0, 0);
    for (int i = 0; i < TRANSMITTANCE_STEPS; ++i) result = result + transmittance_step(r, i);
    return transmittance_finish(result);
}

// CLAMP + LINEAR tap of the transmittance LUT (fp16-rounded values widened to float), sky_lut.gd:62-68
CSKY_HD F4 lut_tap_clamp(const float4* t, int w, int h, float sx, float sy) {
    const float ux = sx * (float)w - 0.5f, uy = sy * (float)h - 0.5f;
    const float fx0 = floorf(ux), fy0 = floorf(uy), ax = ux - fx0, ay = uy - fy0;
    int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > w - 1 ? w - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
    const float4 a = t[y0 * w + x0], b = t[y0 * w + x1], c = t[y1 * w + x0], d = t[y1 * w + x1];
    return f4(lerpf(lerpf(a.x, b.x, ax), lerpf(c.x, d.x, ax), ay), lerpf(lerpf(a.y, b.y, ax), lerpf(c.y, d.y, ax), ay),
              lerpf(lerpf(a.z, b.z, ax), lerpf(c.z, d.z, ax), ay), lerpf(lerpf(a.w, b.w, ax), lerpf(c.w, d.w, ax), ay));
}
// S:137-142
CSKY_HD F4 transmittance_from_lut(const float4* t, int tw, int th, float cos_theta, float normalized_altitude) {
    return lut_tap_clamp(t, tw, th, sat(cos_theta * 0.5f + 0.5f), sat(normalized_altitude));
}

// sky-lut.glsl main() (S:278-315) + compute_inscattering (S:219-276), split so that the 30 in-scattering steps of one
// texel can be evaluated by 30 lanes in parallel: every step's source term and transmittance are independent of the
// others; only the front-to-back accumulation (S:271-272) is sequential, and it is replayed in the reference's order.
struct SkyRay { float rdx, rdy, rdz, oz, dt, sdx, sdy, sdz, molecular_phase, aerosol_phase; };
struct SkyStep { F4 S_int, step_tr; };

CSKY_HD SkyRay sky_ray(int px, int py, float w, float h, const float sun[3]) {
    SkyRay r;
    const float uvx = (float)px / w, uvy = (float)py / h;                             // S:284
    const float azimuth = (float)(2.0 * LUT_PI) * uvx;                                // S:286
    const float l = uvy * 2.0f - 1.0f;                                                // S:290
    const float elev = l * l * signf(l) * (float)LUT_PI * 0.5f;                       // S:291
    r.rdx = cos_cr(elev) * cos_cr(azimuth); r.rdy = cos_cr(elev) * sin_cr(azimuth); r.rdz = sin_cr(elev);   // S:293-295
    r.oz = 6371.5f;                                                                   // S:61-62
    const float atmos_dist = ray_sphere_intersection(0, 0, r.oz, r.rdx, r.rdy, r.rdz, ATMOSPHERE_RADIUS);
    const float ground_dist = ray_sphere_intersection(0, 0, r.oz, r.rdx, r.rdy, r.rdz, EARTH_RADIUS);
    const float t_d = (ground_dist < 0.0f) ? atmos_dist : ground_dist;                // S:303-309
    r.sdx = -sun[0]; r.sdy = -sun[2]; r.sdz = sun[1];                                 // S:221-223
    const float cos_theta = (-r.rdx) * r.sdx + (-r.rdy) * r.sdy + (-r.rdz) * r.sdz;   // S:224
    r.molecular_phase = (float)((3.0 / 16.0) * (1.0 / LUT_PI)) * (1.0f + cos_theta * cos_theta);  // S:114-117
    const float den = (float)(1.0 + 0.8 * 0.8) + (float)(2.0 * 0.8) * cos_theta;      // S:124
    r.aerosol_phase = (float)(0.25 * (1.0 / LUT_PI)) * (1.0f - (float)(0.8 * 0.8)) / (den * sqrtf(den));  // S:125
    r.dt = t_d / (float)IN_SCATTERING_STEPS;                                          // S:229
    return r;
}

// one iteration of the loop at S:234-273, up to (not including) the accumulation
CSKY_HD SkyStep sky_step(const SkyRay& r, int i, const float4* trans, int tw, int th) {
    const float dt = r.dt;
    const float t = ((float)i + 0.5f) * dt;
    const float x = 0.0f + r.rdx * t, y = 0.0f + r.rdy * t, z = r.oz + r.rdz * t;
    const float dist = sqrtf(x * x + y * y + z * z);
    const float zx = x / dist, zy = y / dist, zz = z / dist;
    const float altitude = dist - EARTH_RADIUS;
    const float nalt = altitude / ATMOSPHERE_THICKNESS;
    const float sct = zx * r.sdx + zy * r.sdy + zz * r.sdz;                          // S:243
    const Coeffs cf = atmosphere_collision_coefficients(altitude);
    const F4 t_sun = transmittance_from_lut(trans, tw, th, sct, nalt);               // S:254
    // get_multiple_scattering, S:144-164
    const float omega = (float)(2.0 * LUT_PI) * (1.0f - sqrtf(dist * dist - EARTH_RADIUS * EARTH_RADIUS) / dist);
    const F4 T_to_ground = transmittance_from_lut(trans, tw, th, sct, 0.0f);
    const F4 T_g2s = transmittance_from_lut(trans, tw, th, 1.0f, 0.0f) / transmittance_from_lut(trans, tw, th, 1.0f, nalt);
    const float ks = (float)(0.25 * (1.0 / LUT_PI)) * omega * (float)(0.3 / LUT_PI);
    const F4 L_ground = f4(ks, ks, ks, ks) * T_to_ground * T_g2s * sct;
    const float fm = 1.0f / (1.0f + 5.0f * exp_cr(-17.92f * sct));
    const F4 L_ms = f4((float)(0.02 * 0.217), (float)(0.02 * 0.347), (float)(0.02 * 0.594), (float)(0.02 * 1.0)) * fm;
    const F4 ms = L_ms + L_ground;
    const F4 irr = f4(1.679f, 1.828f, 1.986f, 1.307f);                               // S:67
    const F4 S = irr * (cf.molecular_scattering * (t_sun * r.molecular_phase + ms) + cf.aerosol_scattering * (t_sun * r.aerosol_phase + ms));  // S:261-263
    SkyStep o;
    o.step_tr = exp4(cf.extinction * (-dt));                                         // S:265
    const F4 ext_c = f4(fmaxf(cf.extinction.x, 1e-7f), fmaxf(cf.extinction.y, 1e-7f), fmaxf(cf.extinction.z, 1e-7f), fmaxf(cf.extinction.w, 1e-7f));
    o.S_int = (S - S * o.step_tr) / ext_c;                                           // S:270
    return o;
}
CSKY_HD void sky_accumulate(F4& L, F4& Tr, const SkyStep& s) { L = L + Tr * s.S_int; Tr = Tr * s.step_tr; }   // S:271-272
// S:207-217: mat4x3 M (column-major, 4 columns of 3), S:313
CSKY_HD F4 sky_output(const F4& L) {
    const float r = 137.672389239975f * L.x + 32.549094028629234f * L.y + -38.91428392614275f * L.z + 8.572844237945445f * L.w;
    const float g = -8.632904716299537f * L.x + 91.29801417199785f * L.y + 34.31665471469816f * L.z + -11.103384660054624f * L.w;
    const float b = -1.7181567391931372f * L.x + -12.005406444382531f * L.y + 29.89044807197628f * L.z + 117.47585277566478f * L.w;
    return f4(r, g, b, 1.0f);
}
// the whole texel on one lane (host-compiled unit test; the kernel spreads the steps over lanes)
CSKY_HD F4 sky_texel(int px, int py, float w, float h, const float sun[3], const float4* trans, int tw, int th) {
    const SkyRay r = sky_ray(px, py, w, h, sun);
    F4 L = f4(0, 0, 0, 0), Tr = f4(1, 1, 1, 1);
    for (int i = 0; i < IN_SCATTERING_STEPS; ++i) sky_accumulate(L, Tr, sky_step(r, i, trans, tw, th));
    return sky_output(L);
}

}  // namespace csky
