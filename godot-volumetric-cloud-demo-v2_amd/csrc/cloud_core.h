// cloud_core.h -- the per-ray cloud march of clouds.glsl, written for one ray per lane.
//
// Host+device (CSKY_HD) so that tests/hostsim can single-step the exact kernel maths on a CPU against the
// oracle before it ever runs on a GPU.  The PRODUCT only ever instantiates it inside the HIP kernels of
// cloud_kernels.hip; there is no CPU render path in libcloudsky.
//
// Section A is compiled with FP contraction OFF and IEEE-exact sqrt/div: it holds every expression whose
// fp32 rounding decides WHERE a sample lands (ray set-up, |p|, texture coordinates).  Those reproduce the
// reference's evaluation order bit for bit (positions are ~6e6 m, ulp 0.5 m; intersectSphere
// cancels catastrophically).  Section B (filtering, remaps, shading) may contract and uses the hardware
// exp2/log2/rcp approximations; its error is continuous and ~1e-6 relative (tests state the tolerance).
#pragma once
#include "csky_common.h"

namespace csky {

#if defined(__HIP_DEVICE_COMPILE__)
CSKY_HD float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
CSKY_HD float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
CSKY_HD float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
#define CSKY_WAVE_ALL(x) (__all(x) != 0)
#else
CSKY_HD float fast_rcp(float x) { return 1.0f / x; }
CSKY_HD float fast_exp2(float x) { return exp2f(x); }
CSKY_HD float fast_log2(float x) { return log2f(x); }
#define CSKY_WAVE_ALL(x) (x)
#endif
CSKY_HD float fast_exp(float x) { return fast_exp2(x * 1.44269504088896f); }
CSKY_HD float fast_pow(float x, float y) { return fast_exp2(y * fast_log2(x)); }  // x >= 0; pow(0,y>0) = 0

// =================================================================================================
// Section A: exact fp32 (no contraction).
// =================================================================================================
// Wave priority inside a light sample (round 4): 0 while the sample computes its addresses and issues its four gathers, CSKY_PRIO_MATH while it does
// the arithmetic on the fetched cells.  A wavefront whose data has arrived then wins the VALU over its neighbours, finishes the sample and issues
// the next one's gathers sooner: two frames in flight 1.626 ms per frame against 1.635 (four A/B pairs, profiles/r04/setprio_ab.txt); the reverse
// (priority to the fetch half) costs 0.8 %, priority for whole flushes 0.5 %, for the replay loop or the primary samples nothing.  -DCSKY_PRIO_MATH=0: off.
#ifndef CSKY_PRIO_MATH
#define CSKY_PRIO_MATH 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && CSKY_PRIO_MATH > 0
#define CSKY_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define CSKY_PRIO(n) ((void)0)
#endif

#pragma clang fp contract(off)

struct Ray {
    float px, py, pz;     // current sample position (starts at the shell entry point)
    float sx, sy, sz;     // dir*ss, the per-step increment (clouds.glsl:173)
    float dx, dy, dz;     // normalised direction
    float ss;             // step length (clouds.glsl:143)
    bool above;           // dir.y > 0 (clouds.glsl:221)
};

// Correctly rounded sqrt for NORMAL positive inputs (|p|^2 ~ 3.6e13, ray set-up values): v_sqrt_f32 (1 ulp) plus the
// residual test of its two neighbours -- the same fix-up LLVM emits for an IEEE sqrtf, minus the denormal pre-scaling
// that can never trigger here (17 -> 9 VALU).  Bit-identical to the oracle's sqrtf.
CSKY_HD float sqrt_exact(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float ed = __builtin_fmaf(-sd, s, x), eu = __builtin_fmaf(-su, s, x);
    float r = (ed <= 0.0f) ? sd : s;
    r = (eu > 0.0f) ? su : r;
    return r;
#else
    return sqrtf(x);
#endif
}
CSKY_HD float length3_exact(float x, float y, float z) { return sqrt_exact(x * x + y * y + z * z); }

// Correctly rounded sqrt for |p|^2 of SAMPLE positions, which only ever lie between the two cloud shells plus the light march's reach:
// |p| in [5 997 500, 6 008 000] m, i.e. x in [3.597e13, 3.6097e13] -- 30 067 consecutive floats.  On that range the two-FMA Newton step
//     y = rsq(x); g = x*y; h = y/2; d = fma(-g, g, x); r = fma(d, h, g)
// returns exactly the IEEE sqrt: checked EXHAUSTIVELY on the MI355X against the host's sqrtf by tests/test_gpu_round2.py
// (csky_test_sqrt_shell; v_rsq_f32 is a fixed hardware function).  In general this short form is only "almost always" correctly rounded
// (Markstein's proof needs one more refinement), hence the restriction to the verified range; sqrt_exact above stays for everything else.
// 8.1 + 2 x 2.3 + 2 x 2.3 = 17 issue cycles instead of 34 (v_sqrt + two neighbour residuals + two compare/select pairs, half-rate kinds):
// the march evaluates it once per primary AND per light sample, 8 % of the kernel's VALU time before.
constexpr float SHELL_SQRT_LO = 3.597e13f, SHELL_SQRT_HI = 3.6097e13f;
CSKY_HD float sqrt_shell(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
#else
    return sqrtf(x);
#endif
}
CSKY_HD float length3_shell(float x, float y, float z) { return sqrt_shell(x * x + y * y + z * z); }

// clouds.glsl:97-105 with pos = camPos = (0, g_radius, 0)
CSKY_HD float intersect_sphere_cam(float dx, float dy, float dz, float r) {
    const float a = dx * dx + dy * dy + dz * dz;
    const float b = 2.0f * (dx * 0.0f + dy * G_RADIUS + dz * 0.0f);
    const float c = (0.0f * 0.0f + G_RADIUS * G_RADIUS + 0.0f * 0.0f) - (r * r);
    const float d = sqrtf((b * b) - 4.0f * a * c);
    const float p = -b - d, p2 = -b + d;
    return fmaxf(p, p2) / (2.0f * a);
}

// clouds.glsl:258-262 (main), :248-256 (oct_to_vec3), :218-231 (sky) and :143-145 (march prologue).
// gx, gy = gl_GlobalInvocationID.xy.
CSKY_HD Ray ray_setup(const FrameConsts& fc, int gx, int gy) {
    Ray r;
    const int px = gx + fc.upd_x, py = gy + fc.upd_y;
    const float ex = (float)px / fc.tex_w, ey = (float)py / fc.tex_h;
    float nx = ex - ey;
    float ny = (ex + ey) - 1.0f;
    const float nz = 1.0f - fabsf(nx) - fabsf(ny);
    if (!(nz >= 0.0f)) {  // oct_wrap, clouds.glsl:239-244 (dead for uv in [0,1)^2, kept for fidelity)
        const float sx = nx >= 0.0f ? 1.0f : -1.0f, sy = ny >= 0.0f ? 1.0f : -1.0f;
        const float wx = (1.0f - fabsf(ny)) * sx, wy = (1.0f - fabsf(nx)) * sy;
        nx = wx; ny = wy;
    }
    const float nl = length3_exact(nx, ny, nz);
    const float dx = nx / nl, dy = nz / nl, dz = ny / nl;  // .xzy swizzle, clouds.glsl:262
    r.above = dy > 0.0f;
    if (!r.above) { r.px = r.py = r.pz = r.sx = r.sy = r.sz = r.dx = r.dy = r.dz = r.ss = 0.0f; return r; }
    const float t0 = intersect_sphere_cam(dx, dy, dz, SKY_B_RADIUS);
    const float t1 = intersect_sphere_cam(dx, dy, dz, SKY_T_RADIUS);
    const float s0x = 0.0f + dx * t0, s0y = G_RADIUS + dy * t0, s0z = 0.0f + dz * t0;  // start, clouds.glsl:224
    const float e0x = 0.0f + dx * t1, e0y = G_RADIUS + dy * t1, e0z = 0.0f + dz * t1;  // end,   clouds.glsl:225
    const float shelldist = length3_exact(e0x - s0x, e0y - s0y, e0z - s0z);
    const float rx = dx * shelldist / fc.steps_f, ry = dy * shelldist / fc.steps_f, rz = dz * shelldist / fc.steps_f;  // :230
    r.ss = length3_exact(rx, ry, rz);                                                  // :143
    r.dx = rx / r.ss; r.dy = ry / r.ss; r.dz = rz / r.ss;                              // :144
    r.sx = r.dx * r.ss; r.sy = r.dy * r.ss; r.sz = r.dz * r.ss;
    // clouds.glsl:145: p = pos + dir*hash(pos*10)*ss.  hash() == 0 for every ray in fp32: pos.y*10*0.3183099
    // >= 1.9e7 > 2^24, so fract(...) of the y component is 0 and the product vanishes (SURVEY A.6.1;
    // tests/test_oracle_structure.py checks the oracle's literal evaluation).  Hence p = pos exactly.
    r.px = s0x; r.py = s0y; r.pz = s0z;
    return r;
}

// weather texture coordinate, clouds.glsl:174 / :189 / :197 : p.xz * 0.00006 + 0.5 + weather_pos
CSKY_HD void weather_coord(float px, float pz, float wx, float wy, float& sx, float& sy) {
    sx = px * 0.00006f + 0.5f + wx;
    sy = pz * 0.00006f + 0.5f + wy;
}
// shape-noise coordinate, clouds.glsl:114,117: (p + wind) * 0.00008 ; detail, :128-132: (p + wind - detail) * 0.001
CSKY_HD void shape_coord(const FrameConsts& fc, float px, float py, float pz, float& qx, float& qy, float& qz, float& sx, float& sy, float& sz) {
    qx = px + fc.cloud_off_x; qy = py; qz = pz + fc.cloud_off_z;
    sx = qx * 0.00008f; sy = qy * 0.00008f; sz = qz * 0.00008f;
}
CSKY_HD void detail_coord(const FrameConsts& fc, float qx, float qy, float qz, float& sx, float& sy, float& sz) {
    sx = (qx - fc.det_off_x) * 0.001f; sy = (qy - fc.det_off_y) * 0.001f; sz = (qz - fc.det_off_z) * 0.001f;
}
CSKY_HD void advance(float& x, float& y, float& z, float ix, float iy, float iz) { x = x + ix; y = y + iy; z = z + iz; }

// Per-frame constants (clouds.glsl:114,128-129,148-150,160-170,187,195).  Runs once per frame (one lane).
CSKY_HD float sky_lut_uv_x(float dz, float dx) { return atan2f(dz, dx) / CLOUD_PI * 0.5f + 0.5f; }
CSKY_HD float sky_lut_uv_y(float dy) { const float th = asinf(dy); return sqrtf(fabsf(th) / (CLOUD_PI * 0.5f)) * signf(th) * 0.5f + 0.5f; }

// cell of one CLAMP + LINEAR tap (cloud_sky.gd:381-390): the four texel coordinates and the two weights
CSKY_HD void sky_lut_cell(int w, int h, float sx, float sy, int& x0, int& x1, int& y0, int& y1, float& ax, float& ay) {
    const float ux = sx * (float)w - 0.5f, uy = sy * (float)h - 0.5f;
    const float fx0 = floorf(ux), fy0 = floorf(uy);
    ax = ux - fx0; ay = uy - fy0;
    x0 = (int)fx0; y0 = (int)fy0; x1 = x0 + 1; y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > w - 1 ? w - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
}
// fetch(corner, x, y): the LUT texel (x, y), corner = 0..3 of the cell (x0 y0, x1 y0, x0 y1, x1 y1)
template <class Fetch> CSKY_HD void sky_lut_tap_f(Fetch fetch, int w, int h, float sx, float sy, float out[3]) {
    int x0, x1, y0, y1; float ax, ay;
    sky_lut_cell(w, h, sx, sy, x0, x1, y0, y1, ax, ay);
    const float4 a = fetch(0, x0, y0), b = fetch(1, x1, y0), c = fetch(2, x0, y1), d = fetch(3, x1, y1);
    out[0] = lerpf(lerpf(a.x, b.x, ax), lerpf(c.x, d.x, ax), ay);
    out[1] = lerpf(lerpf(a.y, b.y, ax), lerpf(c.y, d.y, ax), ay);
    out[2] = lerpf(lerpf(a.z, b.z, ax), lerpf(c.z, d.z, ax), ay);
}
CSKY_HD void sky_lut_tap(const float4* sky, int w, int h, float sx, float sy, float out[3]) {
    sky_lut_tap_f([sky, w](int, int x, int y) { return sky[y * w + x]; }, w, h, sx, sy, out);
}
// where the frame set-up samples the sky LUT (clouds.glsl:163,164,166): tap 0 = towards the light (unnormalised direction), 1 / 2 = 45 degrees
// above / below the horizon.  One place for frame_setup and for the kernel that renders just those texels (kernels.hip frame_setup_taps_kernel).
CSKY_HD void frame_setup_tap_uv(const float light[3], int tap, float& sx, float& sy) {
    const float inv = 1.0f / sqrtf(1.0f * 1.0f + 1.0f * 1.0f + 0.0f * 0.0f);               // normalize(vec3(1,+-1,0))
    if (tap == 0) { sx = sky_lut_uv_x(light[2], light[0]); sy = sky_lut_uv_y(light[1]); }
    else { sx = sky_lut_uv_x(0.0f, inv); sy = sky_lut_uv_y(tap == 1 ? inv : -inv); }
}

// fetch(tap, corner, x, y): texel (x, y) of the sky LUT, asked for by tap 0..2 (frame_setup_tap_uv) as corner 0..3 of its cell
template <class Fetch> CSKY_HD void frame_setup_f(const CloudParams& P, Fetch fetch, int sky_w, int sky_h, int primary_steps, int light_steps,
                                                  float early_eps, float hf_lo, float hf_hi, FrameConsts& fc) {
    const float RV[6][3] = {{0.38051305f, 0.92453449f, -0.02111345f}, {-0.50625799f, -0.03590792f, -0.86163418f},
                            {-0.32509218f, -0.94557439f, 0.01428793f}, {0.09026238f, -0.27376545f, 0.95755165f},
                            {0.28128598f, 0.42443639f, -0.86065785f}, {-0.16852403f, 0.14748697f, 0.97460106f}};  // clouds.glsl:140
    fc.tex_w = P.texture_size[0]; fc.tex_h = P.texture_size[1];
    fc.upd_x = (int)P.update_position[0]; fc.upd_y = (int)P.update_position[1];
    fc.cloud_off_x = 20.0f * P.cloud_pos[0] * 0.6f; fc.cloud_off_z = 20.0f * P.cloud_pos[1] * 0.6f;
    fc.det_off_x = P.detailed_pos[0] * 40.0f; fc.det_off_z = P.detailed_pos[1] * 40.0f; fc.det_off_y = P.time * 40.0f;
    fc.wpos_x = P.weather_pos[0]; fc.wpos_y = P.weather_pos[1];
    const float lx = P.LIGHT_DIRECTION[0], ly = P.LIGHT_DIRECTION[1], lz = P.LIGHT_DIRECTION[2];
    const float ll = length3_exact(lx, ly, lz);
    fc.ldir[0] = lx / ll; fc.ldir[1] = ly / ll; fc.ldir[2] = lz / ll;
    const float lss = (SKY_T_RADIUS - SKY_B_RADIUS) / 64.0f;                              // clouds.glsl:148-149
    for (int j = 0; j < 6; j++) for (int k = 0; k < 3; k++) fc.linc[j][k] = (fc.ldir[k] + RV[j][k] * (float)j) * lss;
    for (int k = 0; k < 3; k++) fc.ldist[k] = fc.ldir[k] * 18.0f * lss;
    fc.hg_g2 = 0.4f - 1.4f * fc.ldir[1];
    float s[3], sx, sy;
    frame_setup_tap_uv(P.LIGHT_DIRECTION, 0, sx, sy);
    sky_lut_tap_f([&fetch](int c, int x, int y) { return fetch(0, c, x, y); }, sky_w, sky_h, sx, sy, s);   // clouds.glsl:163 (unnormalised dir)
    for (int k = 0; k < 3; k++) fc.sun_c[k] = s[k] * 0.1f * P.LIGHT_ENERGY * P.LIGHT_COLOR[k];
    frame_setup_tap_uv(P.LIGHT_DIRECTION, 1, sx, sy);
    sky_lut_tap_f([&fetch](int c, int x, int y) { return fetch(1, c, x, y); }, sky_w, sky_h, sx, sy, s);   // clouds.glsl:164
    for (int k = 0; k < 3; k++) s[k] = s[k] * 0.05f;
    float len = length3_exact(s[0], s[1], s[2]);
    for (int k = 0; k < 3; k++) fc.amb_c[k] = s[k] * (1.0f - 0.5f) + len * 0.5f;           // clouds.glsl:165
    frame_setup_tap_uv(P.LIGHT_DIRECTION, 2, sx, sy);
    sky_lut_tap_f([&fetch](int c, int x, int y) { return fetch(2, c, x, y); }, sky_w, sky_h, sx, sy, s);   // clouds.glsl:166
    for (int k = 0; k < 3; k++) s[k] = s[k] * 5.0f * 0.05f;
    len = length3_exact(s[0], s[1], s[2]);
    for (int k = 0; k < 3; k++) fc.gnd_c[k] = s[k] * (1.0f - 0.5f) + (P.ground_color[k] * len) * 0.5f;  // clouds.glsl:167
    fc.density = P.density; fc.coverage = P.cloud_coverage; fc.cov255 = P.cloud_coverage * (1.0f / 255.0f);
    fc.primary_steps = primary_steps; fc.light_steps = light_steps; fc.steps_f = (float)primary_steps;
    fc.early_eps = early_eps;
    fc.hf_lo = hf_lo; fc.hf_hi = hf_hi;
    fc.ct_mode = 0;                   // set by the caller that knows the weather map's range (api.cpp; kernels.hip frame_setup_kernel)
}
CSKY_HD void frame_setup(const CloudParams& P, const float4* sky, int sky_w, int sky_h, int primary_steps, int light_steps,
                         float early_eps, float hf_lo, float hf_hi, FrameConsts& fc) {
    frame_setup_f(P, [sky, sky_w](int, int, int x, int y) { return sky[y * sky_w + x]; }, sky_w, sky_h, primary_steps, light_steps, early_eps, hf_lo, hf_hi, fc);
}

// =================================================================================================
// Section B: filtering + shading (contraction allowed).
// =================================================================================================
#pragma clang fp contract(fast)

// host-only analysis hook (tools/stage_trace): how far a sample_density() call got before an exact reject stopped it
#if defined(CSKY_TRACE_STAGES) && !defined(__HIP_DEVICE_COMPILE__)
extern thread_local int csky_stage;
#define CSKY_STAGE(s) (csky_stage = (s))
#else
#define CSKY_STAGE(s) ((void)0)
#endif

// a + f*d for an fp16 pair packed in one dword: lo = texel(x) = a, hi = texel(x+1) - texel(x) = d (the x-neighbour DIFFERENCE is
// stored, not the neighbour: an integer in [-2040, 2040], exact in fp16).  On gfx950 this is ONE v_fma_mix_f32: the f16 -> f32
// widening is free inside the FMA (4.4 cycles; byte texels needed cvt + cvt + sub + fma = 17, profiles/r01/valu_issue_rates_gfx950.txt; today's measurement: tools/ubench/valu_rates2.hip), and
// it is literally the reference sampler's a + (b - a)*f with an exact (b - a).
CSKY_HD float lerp_h(uint32_t p, float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(p), "v"(f));
    return r;
#else
    const float a = h2f((uint16_t)(p & 0xffffu)), d = h2f((uint16_t)(p >> 16));
    return fmaf(d, f, a);
#endif
}
// the same for two separately stored fp16 texels a, b (LDS variant: the neighbours are not pre-differenced)
CSKY_HD float lerp_h2(uint32_t a_lo_b_hi, float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    float t, r;
    asm volatile("v_fma_mix_f32 %0, %1, -%2, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a_lo_b_hi), "v"(f));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a_lo_b_hi), "v"(f), "v"(t));
    return r;
#else
    const float lo = h2f((uint16_t)(a_lo_b_hi & 0xffffu)), hi = h2f((uint16_t)(a_lo_b_hi >> 16));
    return fmaf(hi, f, fmaf(lo, -f, lo));
#endif
}

// Texel coordinate -> (floor as int, fraction).  gfx950 has v_cvt_flr_i32_f32 (float -> int with floor rounding) and
// v_fract_f32, so the pair costs 2 instructions instead of floor + cvt + sub; u - floor(u) is exact in fp32 and v_fract returns
// the same value (it only clamps the one-in-2^25 case u = -tiny to 1 - 2^-24 instead of 1.0).
// (Round 2 tried the full-rate alternative t = fma(s, n, 1.5*2^23 - 1), i = bits(t), f = u - (t - 1.5*2^23): four full-rate instead of one
// full + two half-rate instructions per axis.  It is NOT bit-identical -- when u is an exact integer, round-to-nearest-even lands on the cell
// below with f = 1 for half of them, which is the same sample point but three roundings instead of one in the y / z lerps -- and it measured
// SLOWER, 2.20 vs 2.04 ms: u and t are both live per axis and the kernel sits at its 72-VGPR budget, 5 -> 11 spilled registers.  Dropped.)
CSKY_HD void split_coord(float u, int& i, float& f) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(u));
    f = __builtin_amdgcn_fractf(u);
#else
    const float fl = floorf(u);
    i = (int)fl; f = u - fl;
#endif
}

// REPEAT + LINEAR bilinear tap of the quad-packed weather map (clouds.glsl:174).  Returns r (cloud type), b (coverage) ON THE TEXEL SCALE
// 0..255: the UNORM 1/255 is folded into the two consumers (coverage / 255 in FrameConsts, the gradient's slopes), two multiplies less per sample.
CSKY_HD void weather_fetch(const uint4* __restrict__ w, float sx, float sy, uint4& q, float& ax, float& ay) {
    int ix, iy;
    split_coord(sx * 512.0f - 0.5f, ix, ax); split_coord(sy * 512.0f - 0.5f, iy, ay);
    const int x0 = ix & 511, y0 = iy & 511;
    q = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(w) + ((((uint32_t)y0 << 9) | (uint32_t)x0) << 4));
}
CSKY_HD void weather_filter(const uint4& q, float ax, float ay, float& wr, float& wb) {
    wr = fmaf(ay, lerp_h(q.y, ax), lerp_h(q.x, ax));                        // polynomial cell: (c0 + c1 fx) + fy (c2 + c3 fx)
    wb = fmaf(ay, lerp_h(q.w, ax), lerp_h(q.z, ax));
}
CSKY_HD void weather_tap(const uint4* __restrict__ w, float sx, float sy, float& wr, float& wb) {
    uint4 q; float ax, ay;
    weather_fetch(w, sx, sy, q, ax, ay);
    weather_filter(q, ax, ay, wr, wb);
}
// the tap of whichever cell form the bound texture set has (TexSet: fp16 pairs; TexSet32: exact fp32 coefficients, bake_core.h): the same
// polynomial (c0 + c1 fx) + fy (c2 + c3 fx), the x stage a plain FMA instead of v_fma_mix_f32
template <class TS>
CSKY_HD void weather_tap_ts(const TS& T, float sx, float sy, float& wr, float& wb) {
    if constexpr (TS::cell32) {
        int ix, iy; float ax, ay;
        split_coord(sx * 512.0f - 0.5f, ix, ax); split_coord(sy * 512.0f - 0.5f, iy, ay);
        const float4* __restrict__ q = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T.weather32) + (((((uint32_t)(iy & 511)) << 9) | (uint32_t)(ix & 511)) << 5));
        const float4 r = q[0], b = q[1];
        wr = fmaf(ay, fmaf(r.w, ax, r.z), fmaf(r.y, ax, r.x));
        wb = fmaf(ay, fmaf(b.w, ax, b.z), fmaf(b.y, ax, b.x));
    } else {
        weather_tap(T.weather, sx, sy, wr, wb);
    }
}

// (float)(1 << e) built from the exponent field: integer SALU work when e is wave-uniform (the light march's LOD), where a cast costs a
// half-rate v_cvt per tap
CSKY_HD float pow2f(int e) { const uint32_t b = (uint32_t)(127 + e) << 23; float f; memcpy(&f, &b, 4); return f; }
// index of cell (x0,y0,z0) inside a shape level with n = 1 << sh cells per side (== bake_core.h::shape_cell_index)
CSKY_HD uint32_t shape_cell_offset(uint32_t x0, uint32_t y0, uint32_t z0, uint32_t sh) { return (((z0 << sh) | y0) << sh) | x0; }
// texel offset of mip level l inside the packed chains: sum_{i<l} (N>>i)^3 = (N^3*8 - (N>>l)^3*8) / 7, computed
// arithmetically so a per-lane level needs no table (api.cpp checks it against the baked offsets)
CSKY_HD uint32_t shape_level_offset(int l) { return ((1u << 24) - (1u << (24 - 3 * l))) / 7u; }
CSKY_HD uint32_t detail_level_offset(int l) { return ((1u << 18) - (1u << (18 - 3 * l))) / 7u; }
// REPEAT + LINEAR trilinear tap of the shape volume at integer level `lvl` (clouds.glsl:117).
// Returns r = n.r and fbm = n.g*0.625 + n.b*0.25 + n.a*0.125 (clouds.glsl:118; exact integer numerators, filtered linearly).
template <class TS>
CSKY_HD void shape_tap(const TS& T, int lvl, float sx, float sy, float sz, float& r, float& fbm) {
    const int n = SHAPE_N >> lvl, m = n - 1;
    const float fn = pow2f(7 - lvl);
    int ix, iy, iz; float ax, ay, az;
    split_coord(sx * fn - 0.5f, ix, ax); split_coord(sy * fn - 0.5f, iy, ay); split_coord(sz * fn - 0.5f, iz, az);
    const int x0 = ix & m, y0 = iy & m, z0 = iz & m;
    const uint32_t sh = (uint32_t)(7 - lvl), base = shape_level_offset(lvl) + (uint32_t)x0;   // n = 1 << sh: shifts, not v_mul_lo_u32 (quarter rate)
    if constexpr (TS::cell32) {                               // exact cells: 4 x float4 per texel, the xyz polynomial on fp32 coefficients
        (void)base;
        const float4* __restrict__ t = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T.shape32) +
                                                                       ((size_t)(shape_level_offset(lvl) + shape_cell_offset((uint32_t)x0, (uint32_t)y0, (uint32_t)z0, sh)) << 6));
        const float4 r0 = t[0], r1 = t[1], f0 = t[2], f1 = t[3];
        r = fmaf(az, fmaf(ay, fmaf(r1.w, ax, r1.z), fmaf(r1.y, ax, r1.x)), fmaf(ay, fmaf(r0.w, ax, r0.z), fmaf(r0.y, ax, r0.x))) * (1.0f / 255.0f);
        fbm = fmaf(az, fmaf(ay, fmaf(f1.w, ax, f1.z), fmaf(f1.y, ax, f1.x)), fmaf(ay, fmaf(f0.w, ax, f0.z), fmaf(f0.y, ax, f0.x))) * (1.0f / (8.0f * 255.0f));
        return;
    }
    const char* __restrict__ sb = reinterpret_cast<const char*>(T.shape);
#if CSKY_SHAPE_POLY == 1
    const int y1 = (y0 + 1) & m, z1 = (z0 + 1) & m;
    const uint32_t r00 = ((((uint32_t)z0 << sh) | (uint32_t)y0) << sh), r10 = ((((uint32_t)z0 << sh) | (uint32_t)y1) << sh);
    const uint32_t r01 = ((((uint32_t)z1 << sh) | (uint32_t)y0) << sh), r11 = ((((uint32_t)z1 << sh) | (uint32_t)y1) << sh);
    const uint2 t00 = *reinterpret_cast<const uint2*>(sb + ((base + r00) << 3)), t10 = *reinterpret_cast<const uint2*>(sb + ((base + r10) << 3));
    const uint2 t01 = *reinterpret_cast<const uint2*>(sb + ((base + r01) << 3)), t11 = *reinterpret_cast<const uint2*>(sb + ((base + r11) << 3));
    r = lerpf(lerpf(lerp_h(t00.x, ax), lerp_h(t10.x, ax), ay), lerpf(lerp_h(t01.x, ax), lerp_h(t11.x, ax), ay), az) * (1.0f / 255.0f);
    fbm = lerpf(lerpf(lerp_h(t00.y, ax), lerp_h(t10.y, ax), ay), lerpf(lerp_h(t01.y, ax), lerp_h(t11.y, ax), ay), az) * (1.0f / (8.0f * 255.0f));
#elif CSKY_SHAPE_POLY == 2
    const int z1 = (z0 + 1) & m;
    const uint32_t r0 = ((((uint32_t)z0 << sh) | (uint32_t)y0) << sh), r1 = ((((uint32_t)z1 << sh) | (uint32_t)y0) << sh);
    const uint4 t0 = *reinterpret_cast<const uint4*>(sb + ((base + r0) << 4)), t1 = *reinterpret_cast<const uint4*>(sb + ((base + r1) << 4));
    r = lerpf(fmaf(ay, lerp_h(t0.y, ax), lerp_h(t0.x, ax)), fmaf(ay, lerp_h(t1.y, ax), lerp_h(t1.x, ax)), az) * (1.0f / 255.0f);
    fbm = lerpf(fmaf(ay, lerp_h(t0.w, ax), lerp_h(t0.z, ax)), fmaf(ay, lerp_h(t1.w, ax), lerp_h(t1.z, ax)), az) * (1.0f / (8.0f * 255.0f));
#else
    (void)base;
    const uint4* __restrict__ t = reinterpret_cast<const uint4*>(sb + ((shape_level_offset(lvl) + shape_cell_offset((uint32_t)x0, (uint32_t)y0, (uint32_t)z0, sh)) << 5));
    const uint4 tr = t[0], tf = t[1];             // (marking these loads non-temporal to spare the L1 for the other textures: 1.70 -> 2.48 ms; the cells ARE re-used)
    r = fmaf(az, fmaf(ay, lerp_h(tr.w, ax), lerp_h(tr.z, ax)), fmaf(ay, lerp_h(tr.y, ax), lerp_h(tr.x, ax))) * (1.0f / 255.0f);
    fbm = fmaf(az, fmaf(ay, lerp_h(tf.w, ax), lerp_h(tf.z, ax)), fmaf(ay, lerp_h(tf.y, ax), lerp_h(tf.x, ax))) * (1.0f / (8.0f * 255.0f));
#endif
}

// REPEAT + LINEAR trilinear tap of the oct-packed detail volume (clouds.glsl:132-133): returns hfbm.
template <class TS>
CSKY_HD float detail_tap(const TS& T, int lvl, float sx, float sy, float sz) {
    // LOD 5 is 1x1x1: with REPEAT all eight corners are that one texel and a + (a - a)*f = a exactly (what the reference's
    // sampler returns), so light samples j = 5 and the distant sample (clouds.glsl:190,198: textureLod(.., 5.0)) need no tap
    if (lvl == 5) return T.detail_lod5;
    const int n = DETAIL_N >> lvl, m = n - 1;
    const float fn = pow2f(5 - lvl);
    int ix, iy, iz; float ax, ay, az;
    split_coord(sx * fn - 0.5f, ix, ax); split_coord(sy * fn - 0.5f, iy, ay); split_coord(sz * fn - 0.5f, iz, az);
    const int x0 = ix & m, y0 = iy & m, z0 = iz & m;
    if constexpr (TS::cell32) {                               // exact cells: 2 x float4 per texel
        const uint32_t sh3 = (uint32_t)(5 - lvl), idx3 = detail_level_offset(lvl) + ((((((uint32_t)z0 << sh3) | (uint32_t)y0) << sh3)) | (uint32_t)x0);
        const float4* __restrict__ t = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T.detail32) + ((size_t)idx3 << 5));
        const float4 q0 = t[0], q1 = t[1];
        return fmaf(az, fmaf(ay, fmaf(q1.w, ax, q1.z), fmaf(q1.y, ax, q1.x)), fmaf(ay, fmaf(q0.w, ax, q0.z), fmaf(q0.y, ax, q0.x))) * (1.0f / (8.0f * 255.0f));
    }
    if (T.detail_lds) {
        // "lds" variant (north star: noise bricks staged in LDS): the whole detail chain sits in LDS as unpacked fp16 texels, so
        // a tap is eight 2-byte LDS reads assembled into the same x-neighbour pairs the global layout stores pre-packed
        const int x1 = (x0 + 1) & m, y1 = (y0 + 1) & m, z1 = (z0 + 1) & m;
        const uint32_t sh2 = (uint32_t)(5 - lvl);
        const uint16_t* __restrict__ d = T.detail_lds + detail_level_offset(lvl);
        const uint32_t r00 = (((uint32_t)z0 << sh2) | (uint32_t)y0) << sh2, r10 = (((uint32_t)z0 << sh2) | (uint32_t)y1) << sh2;
        const uint32_t r01 = (((uint32_t)z1 << sh2) | (uint32_t)y0) << sh2, r11 = (((uint32_t)z1 << sh2) | (uint32_t)y1) << sh2;
        const uint32_t p00 = (uint32_t)d[r00 + x0] | ((uint32_t)d[r00 + x1] << 16), p10 = (uint32_t)d[r10 + x0] | ((uint32_t)d[r10 + x1] << 16);
        const uint32_t p01 = (uint32_t)d[r01 + x0] | ((uint32_t)d[r01 + x1] << 16), p11 = (uint32_t)d[r11 + x0] | ((uint32_t)d[r11 + x1] << 16);
        return lerpf(lerpf(lerp_h2(p00, ax), lerp_h2(p10, ax), ay), lerpf(lerp_h2(p01, ax), lerp_h2(p11, ax), ay), az) * (1.0f / (8.0f * 255.0f));
    }
    const uint32_t sh = (uint32_t)(5 - lvl), idx = detail_level_offset(lvl) + ((((((uint32_t)z0 << sh) | (uint32_t)y0) << sh)) | (uint32_t)x0);
    const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(T.detail) + (idx << 4));
    return fmaf(az, fmaf(ay, lerp_h(q.w, ax), lerp_h(q.z, ax)), fmaf(ay, lerp_h(q.y, ax), lerp_h(q.x, ax))) * (1.0f / (8.0f * 255.0f));   // polynomial cell
}

CSKY_HD float height_fraction(float r) { return sat((r - SKY_B_RADIUS) * (1.0f / (SKY_T_RADIUS - SKY_B_RADIUS))); }  // clouds.glsl:77-80

// clouds.glsl:82-95: g = smoothstep(gx, gy, hf) - smoothstep(gz, gw, hf), the four corners mixed by cloud type.
//  * mixGradients() is piecewise linear in the cloud type ct: below 0.5 only stratus (1-2ct) and stratocumulus (2ct) are non-zero, above 0.5
//    only stratocumulus (2-2ct) and cumulus (2ct-1), so each corner is A + ct*B with (A,B) picked by the branch.
//  * Round 4: the two ramps never overlap -- gy < gz for every cloud type (0.05..0.2 < 0.09..0.48 below 0.5, 0.0625..0.2 < 0.48..0.78 above) --
//    so of t1 = sat((hf-gx)/(gy-gx)) and t2 = sat((hf-gz)/(gw-gz)) at most one lies strictly inside (0,1): t1 < 1 implies t2 == 0, t2 > 0
//    implies t1 == 1.  With S(t) = t^2 (3 - 2t) and S(1 - t) = 1 - S(t):   S(t1) - S(t2) = S(t1 - t2),   ONE polynomial instead of two; and
//    both reciprocals come from ONE v_rcp_f32: r = 1 / (d1 d2), 1/d1 = r d2, 1/d2 = r d1, with the widths d1 = gy-gx, d2 = gw-gz linear in ct
//    like the corners.  55 -> 42 issue cycles per evaluation (7.5 M wave-level evaluations per C3 frame, profiles/r04/line_profile_*.txt);
//    the value moves by ~1e-7 (section B tolerance; the height window keeps 0.03 of slack in g, bake.h).
CSKY_HD float density_height_gradient(const FrameConsts& fc, float hf, float ct) {
    const float c = ct;
    constexpr float K = 1.0f / 255.0f;
    float gx, d1, gz, d2;
    if (fc.ct_mode == 1) {
        gx = 0.03f + c * (-0.02f * K); d1 = (0.3375f - 0.03f) + c * ((-0.275f + 0.02f) * K); gz = 0.18f + c * (0.6f * K); d2 = (0.25f - 0.18f) + c * ((0.75f - 0.6f) * K);
    } else if (fc.ct_mode == 2) {
        gx = 0.02f + c * 0.0f; d1 = (0.05f - 0.02f) + c * (0.3f * K); gz = 0.09f + c * (0.78f * K); d2 = (0.11f - 0.09f) + c * ((1.03f - 0.78f) * K);
    } else {
        const bool hi = ct >= 127.5f;
        gx = (hi ? 0.03f : 0.02f) + c * (hi ? -0.02f * K : 0.0f);
        d1 = (hi ? 0.3375f - 0.03f : 0.05f - 0.02f) + c * (hi ? (-0.275f + 0.02f) * K : 0.3f * K);
        gz = (hi ? 0.18f : 0.09f) + c * (hi ? 0.6f * K : 0.78f * K);
        d2 = (hi ? 0.25f - 0.18f : 0.11f - 0.09f) + c * (hi ? (0.75f - 0.6f) * K : (1.03f - 0.78f) * K);
    }
    const float r = fast_rcp(d1 * d2);
    const float t = sat((hf - gx) * (r * d2)) - sat((hf - gz) * (r * d1));
    return t * t * (3.0f - 2.0f * t);
}

// clouds.glsl:109-137 density().  (px,py,pz) sample point, hf its height fraction, (wr,wb) the weather
// tap, lod_shape = clamp(mip-2), lod_detail = mip.  Two EXACT early rejects:
//  (1) base_cloud in [0,1] so base_cloud*g <= g; if g <= 1-wc the remap at :124 is <= 0, the product with
//      wc (>= 0) is <= 0, the remap at :135 (denominator 1 - hfbm*0.4*hf >= 0.6 > 0) stays <= 0, the clamp
//      gives 0 and pow(0, e >= 0.5) = 0.  The shape tap is never needed.  wc == 0 (divide by zero -> NaN ->
//      clamp -> 0 in the oracle) also lands here.
//  (2) after :125, base_cloud <= 0 makes :135-136 return 0 for the same reason: the detail tap is dead.
template <class TS>
CSKY_HD float density(const TS& T, const FrameConsts& fc, float px, float py, float pz, float hf, float wr, float wb,
                      int lod_shape, int lod_detail) {
    const float wc = fc.cov255 * wb;                                         // :123 (wb on the texel scale)
    const float g = density_height_gradient(fc, hf, wr);                    // :121
    const float omw = 1.0f - wc;
    if (!(g > omw)) return 0.0f;                                             // exact reject (1)
    float qx, qy, qz, sx, sy, sz;
    shape_coord(fc, px, py, pz, qx, qy, qz, sx, sy, sz);
#ifdef CSKY_BRICK_BOUND
    // EXPERIMENT BUILD ONLY (make brick; profiles/r03/brick_bound_reject_analysis.txt): base_cloud = (r + 1 - fbm) / (2 - fbm) is increasing in r and
    // decreasing in fbm and a trilinear tap lies between its corners, so bmax = that expression at (max r, min fbm) over the brick the cell's base
    // index lies in (+1 texel apron) bounds it; bmax * g <= 1 - wc proves the remap at :124 <= 0, i.e. density() == 0, without the 32-byte gather.
    if (lod_shape == 0) {
        int bx, by, bz; float fx_, fy_, fz_;
        split_coord(sx * 128.0f - 0.5f, bx, fx_); split_coord(sy * 128.0f - 0.5f, by, fy_); split_coord(sz * 128.0f - 0.5f, bz, fz_);
        const uint32_t bi = ((((uint32_t)bz & 127u) >> 3) << 8) | ((((uint32_t)by & 127u) >> 3) << 4) | (((uint32_t)bx & 127u) >> 3);
        const float bm = T.brick[bi];
        if (!(bm * g > omw)) return 0.0f;
    }
#endif
    float nr, fbm;
    shape_tap(T, lod_shape, sx, sy, sz, nr, fbm);                           // :117-118
    CSKY_STAGE(2);
    const float omf = 1.0f - fbm, den1 = 1.0f + omf;                        // den1 in [1, 2]
    // :122 base = remap(n.r, -(1-fbm), 1, 0, 1) = (nr + omf) / den1;  :124-125 remap(base*g, 1-wc, 1, 0, 1) * wc = (base*g - omw) * [wc / (1 - omw)]:
    // the bracket is 1 + O(6e-8 / wc) in fp32 (1 - (1 - wc) equals wc up to one rounding of 1 - wc) and is left out (section B; the
    // coverage-0.05 sweep frame of tools/parity_sweep.py has the largest 1/wc).  Round 4: the quotient is kept as numerator / den1 through
    // reject (2), whose sign test needs no division, and divided ONCE together with :135's denominator: one v_rcp_f32 less per sample.
    float num = (nr + omf) * g - omw * den1;                                // = (base*g - omw) * den1
    if (!(num > 0.0f)) return 0.0f;                                         // reject (2): base*g - omw <= 0
#ifdef CSKY_RESTORE_WC_FACTOR
    num = num * (wc * fast_rcp(1.0f - omw));                                // experiment build (profiles/r04/wc_factor_experiment.txt): the bracket restored
#endif
    detail_coord(fc, qx, qy, qz, sx, sy, sz);                               // :128-129
    float hfbm = detail_tap(T, lod_detail, sx, sy, sz);                     // :132-133
    CSKY_STAGE(3);
    const float k = sat(hf * 4.0f);
    hfbm = hfbm + k * (1.0f - 2.0f * hfbm);                                 // :134 mix(hfbm, 1-hfbm, k)
    const float hm = hfbm * 0.4f * hf, den2 = 1.0f - hm;                    // den2 >= 0.6
    const float base = (num - hm * den1) * fast_rcp(den1 * den2);           // :135 (num/den1 - hm) / den2
    return fast_pow(sat(base), (1.0f - hf) * 0.8f + 0.5f);                  // :136
}

// One density sample of the march: weather tap (clouds.glsl:174/:189/:197) + density() (:109-137), behind a third
// EXACT reject: outside the height window (fc.hf_lo, fc.hf_hi) the height gradient cannot exceed 1 - coverage*weather.b
// for ANY texel of the bound weather map (bake.h height_window: g <= smoothstep(gx,gy,hf) below the cloud body and
// g <= 1 - smoothstep(gz,gw,hf) above it, maximised over the map's cloud-type range), so reject (1) would fire anyway:
// the weather tap and the gradient are skipped.  Samples above/below the cloud body cost ~20 VALU instead of ~100.
template <class TS>
CSKY_HD float sample_density(const TS& T, const FrameConsts& fc, float px, float py, float pz, float hf, float wx, float wy,
                             int lod_shape, int lod_detail) {
    CSKY_STAGE(0);
    if (!(hf > fc.hf_lo && hf < fc.hf_hi)) return 0.0f;
    float wsx, wsy, wr, wb;
    weather_coord(px, pz, wx, wy, wsx, wsy);
    weather_tap_ts(T, wsx, wsy, wr, wb);
    CSKY_STAGE(1);
    return density(T, fc, px, py, pz, hf, wr, wb, lod_shape, lod_detail);
}

// (Round-2 experiment, measured and removed: the cells of detail LODs 2..4 / 3..4 staged in LDS per workgroup and served to the light march
// with one ds_read_b128 per tap: frames bit-identical, C3 2.16 / 2.08 ms against 2.05 ms (LODs 2..4 need 9.3 KB and cost a wavefront per SIMD;
// LODs 3..4 remove 2 of 28 gathers per in-cloud sample and add a workgroup barrier + staging): profiles/r02/layout_lds_ab.txt.)
// (Round 6, measured and removed, profiles/r06/fp16_filter_ab.txt -- the one experiment on PRECISION, VERDICT r5 item 4: the y / z stages of every
// polynomial cell in packed fp16 -- Q = P0 + fy P1 on both halves of a cell's (c_even, c_odd) fp16 pairs with v_pk_fma_f16, weights from one
// v_cvt_pk_f16_f32, only the last stage lo + fx hi widened by v_fma_mix_f32: 69.5 M v_fma_mix + 69.5 M v_fmac per C3 frame became 69.5 M v_pk_fma_f16 +
// 30.2 M v_cvt_pk, VALU instructions -3.6 %, and the kernel got 0.8 % SLOWER alone (1.957 vs 1.940 ms) and 0.5 % slower two in flight: v_pk_fma_f16 and
// v_cvt_pk_f16_f32 issue at the half rate v_fma_mix_f32 does.  The frame left the 2-ulp gate (75 % of the values bit-identical, max |d| 1.8e-2, PSNR 74-81 dB)
// and stayed inside SURVEY 8(c)'s stated tolerance (99.96-99.99 % within 2e-3 + 1e-2 |ref|).  Exactness costs nothing here: the kernel chapter is closed.)
// (Round 5, measured and removed, profiles/r05/kernel_experiments.txt: v_pk_fma_f32 for the y / z stages of the cell pairs: no gain; packed (x, z)
// coordinate chains, v_pk_mul / v_pk_add: frame identical, +4.6 % time.)
// sample_density() with all of a sample's texture fetches issued up front ("eager"): the addresses of the weather, shape and detail
// cells depend only on the sample position, not on each other's results, so the three gathers can be in flight together instead of
// one after the other (one memory latency per sample instead of three; the kernel is as sensitive to latency as to VALU issue:
// 8 -> 5 waves/SIMD costs 25 %).  Same arithmetic, same exact rejects (a rejected sample discards what it fetched).  Used for
// the light march, where 94 % of the samples need all three taps anyway (tools/stage_trace); the primary march keeps the lazy form.
#if CSKY_SHAPE_POLY == 3
// EAGER_DETAIL = false fetches the weather and shape cells together and the detail cell only after reject (2): the form for the
// primary march, where 77 % of the samples inside the height window reach the shape tap but only 31 % the detail tap.
template <bool EAGER_DETAIL = true, class TS>
CSKY_HD float sample_density_eager(const TS& T, const FrameConsts& fc, float px, float py, float pz, float hf, float wx, float wy,
                                   int lod_shape, int lod_detail) {
    if constexpr (TS::cell32) return sample_density(T, fc, px, py, pz, hf, wx, wy, lod_shape, lod_detail);   // exact cells: the lazy form (not the tuned path)
    // (round 4: ONE result register written where the sample survives every reject -- three early `return 0` cost a v_mov each per sample -- and
    // the single-texel LOD 5 of the detail volume as its own wave-uniform case instead of zero-filled tap registers)
    float d = 0.0f;
    if (hf > fc.hf_lo && hf < fc.hf_hi) {
        // ---- addresses + fetches
        CSKY_PRIO(0);
        float wsx, wsy;
        weather_coord(px, pz, wx, wy, wsx, wsy);
        int wix, wiy; float wax, way;
        split_coord(wsx * 512.0f - 0.5f, wix, wax); split_coord(wsy * 512.0f - 0.5f, wiy, way);
        const uint4 wq = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(T.weather) + (((((uint32_t)(wiy & 511)) << 9) | (uint32_t)(wix & 511)) << 4));
        float qx, qy, qz, sx, sy, sz;
        shape_coord(fc, px, py, pz, qx, qy, qz, sx, sy, sz);
        const int sn = SHAPE_N >> lod_shape, sm = sn - 1;
        const float sfn = pow2f(7 - lod_shape);
        int six, siy, siz; float sax, say, saz;
        split_coord(sx * sfn - 0.5f, six, sax); split_coord(sy * sfn - 0.5f, siy, say); split_coord(sz * sfn - 0.5f, siz, saz);
        const uint32_t ssh = (uint32_t)(7 - lod_shape);
        const uint32_t sidx = shape_level_offset(lod_shape) + shape_cell_offset((uint32_t)(six & sm), (uint32_t)(siy & sm), (uint32_t)(siz & sm), ssh);
        const uint4* __restrict__ sp = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(T.shape) + (sidx << 5));
        const uint4 tr = sp[0], tf = sp[1];
        float dsx, dsy, dsz;
        detail_coord(fc, qx, qy, qz, dsx, dsy, dsz);
        const bool tap = EAGER_DETAIL && lod_detail != 5;         // wave-uniform; LOD 5 is one texel (detail_tap)
        uint4 dq;
        float dax, day, daz;
        if (tap) {
            const int dn = DETAIL_N >> lod_detail, dm = dn - 1;
            const float dfn = pow2f(5 - lod_detail);
            int dix, diy, diz;
            split_coord(dsx * dfn - 0.5f, dix, dax); split_coord(dsy * dfn - 0.5f, diy, day); split_coord(dsz * dfn - 0.5f, diz, daz);
            const uint32_t dsh = (uint32_t)(5 - lod_detail);
            const uint32_t didx = detail_level_offset(lod_detail) + ((((((uint32_t)(diz & dm)) << dsh) | (uint32_t)(diy & dm)) << dsh) | (uint32_t)(dix & dm));
            dq = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(T.detail) + (didx << 4));
        }
        // ---- the arithmetic of density() (clouds.glsl:109-137) on the fetched cells
        CSKY_PRIO(CSKY_PRIO_MATH);
        const float wr = fmaf(way, lerp_h(wq.y, wax), lerp_h(wq.x, wax));       // texel scale 0..255 (weather_filter)
        const float wb = fmaf(way, lerp_h(wq.w, wax), lerp_h(wq.z, wax));
        const float wc = fc.cov255 * wb;                                         // :123 (wb on the texel scale)
        const float g = density_height_gradient(fc, hf, wr);                    // :121
        const float omw = 1.0f - wc;
        if (g > omw) {                                                           // else: exact reject (1)
            const float nr = fmaf(saz, fmaf(say, lerp_h(tr.w, sax), lerp_h(tr.z, sax)), fmaf(say, lerp_h(tr.y, sax), lerp_h(tr.x, sax))) * (1.0f / 255.0f);
            const float fbm = fmaf(saz, fmaf(say, lerp_h(tf.w, sax), lerp_h(tf.z, sax)), fmaf(say, lerp_h(tf.y, sax), lerp_h(tf.x, sax))) * (1.0f / (8.0f * 255.0f));
            const float omf = 1.0f - fbm, den1 = 1.0f + omf;
            float num = (nr + omf) * g - omw * den1;                            // :122, :124-125 as numerator / den1 (see density())
            if (num > 0.0f) {                                                    // else: reject (2)
#ifdef CSKY_RESTORE_WC_FACTOR
                num = num * (wc * fast_rcp(1.0f - omw));
#endif
                float hfbm;
                if (tap) hfbm = fmaf(daz, fmaf(day, lerp_h(dq.w, dax), lerp_h(dq.z, dax)), fmaf(day, lerp_h(dq.y, dax), lerp_h(dq.x, dax))) * (1.0f / (8.0f * 255.0f));
                else if (EAGER_DETAIL) hfbm = T.detail_lod5;
                else hfbm = detail_tap(T, lod_detail, dsx, dsy, dsz);           // :132-133, fetched now
                const float k = sat(hf * 4.0f);
                hfbm = hfbm + k * (1.0f - 2.0f * hfbm);                         // :134
                const float hm = hfbm * 0.4f * hf, den2 = 1.0f - hm;
                const float base = (num - hm * den1) * fast_rcp(den1 * den2);   // :135
                d = fast_pow(sat(base), (1.0f - hf) * 0.8f + 0.5f);             // :136
            }
        }
    }
    return d;
}
#else
template <bool EAGER_DETAIL = true, class TS>
CSKY_HD float sample_density_eager(const TS& T, const FrameConsts& fc, float px, float py, float pz, float hf, float wx, float wy,
                                   int lod_shape, int lod_detail) {
    return sample_density(T, fc, px, py, pz, hf, wx, wy, lod_shape, lod_detail);
}
#endif

CSKY_HD float henyey_greenstein(float c, float g) {                         // clouds.glsl:72-75, once per ray
    const float x = 1.0f + g * g - 2.0f * g * c;                            // >= (1 - |g|)^2 >= 0
    return 0.0795774715459f * (1.0f - g * g) / (x * sqrtf(x));             // pow(x, 1.5) = x sqrt(x): ~15 instructions instead of powf's ~80, within 2 fp32 ulp of it
}

// clouds.glsl:202-210 in two halves.  shade_terms: everything about one in-cloud sample that does not depend on the ray's running state
// (density t, height fraction hf, step transmittance dt, summed light-march density cd, the ray's phase value): D = r - r dt per channel
// and q = 1 / max(1e-7, t).  composite_sample: the front-to-back recurrences on (L, alpha, T).  The compact march evaluates the first
// half in the light march's lane layout (one SAMPLE per lane, all 64 lanes busy) and only the second while replaying the steps in order
// (one owner RAY per lane, ~1/6 of the lanes busy per step); the expressions and their order are those of the one-piece form.
CSKY_HD void shade_terms(const FrameConsts& fc, float phase, float t, float hf, float dt, float cd, float& Dr, float& Dg, float& Db, float& q) {
    const float lss = (SKY_T_RADIUS - SKY_B_RADIUS) / 64.0f;
    const float nd = -fc.density;
    const float beers = fast_exp(nd * cd * lss * 3.0f);                                  // :202
    const float powder = 1.0f - beers * beers;                                           // :203 exp(2x) = exp(x)^2: one v_exp_f32 less (round 4)
    const float bt = 2.0f * beers * powder;                                              // :204
    const float sm = hf * hf * (3.0f - 2.0f * hf);                                       // smoothstep(0,1,hf), hf already in [0,1]
    const float ar = fc.gnd_c[0] * (1.0f - sm) + fc.amb_c[0] * sm;                       // :206
    const float ag = fc.gnd_c[1] * (1.0f - sm) + fc.amb_c[1] * sm;
    const float ab = fc.gnd_c[2] * (1.0f - sm) + fc.amb_c[2] * sm;
    const float k = bt * phase;
    const float rr = (ar + k * fc.sun_c[0]) * t, rg = (ag + k * fc.sun_c[1]) * t, rb = (ab + k * fc.sun_c[2]) * t;  // :208
    q = fast_rcp(fmaxf(0.0000001f, t));                                                  // :209
    Dr = rr - rr * dt; Dg = rg - rg * dt; Db = rb - rb * dt;
}
CSKY_HD void composite_sample(float dt, float q, float Dr, float Dg, float Db, float& Tr, float& alpha, float& Lr, float& Lg, float& Lb) {
    alpha += (1.0f - dt) * (1.0f - alpha);                                               // :207
    const float w = Tr * q;                                                              // :209
    Lr += Dr * w; Lg += Dg * w; Lb += Db * w;
    Tr *= dt;                                                                            // :210
}
CSKY_HD void shade_sample(const FrameConsts& fc, float phase, float t, float hf, float dt, float cd, float& Tr, float& alpha, float& Lr,
                          float& Lg, float& Lb) {
    float Dr, Dg, Db, q;
    shade_terms(fc, phase, t, hf, dt, cd, Dr, Dg, Db, q);
    composite_sample(dt, q, Dr, Dg, Db, Tr, alpha, Lr, Lg, Lb);
}

struct MarchOut { float r, g, b, a, t; uint32_t incloud; };   // L.rgb, alpha, transmittance T, #in-cloud samples

// clouds.glsl:139-215 march() for one ray.
template <class TS>
CSKY_HD MarchOut march(const TS& T, const FrameConsts& fc, Ray ray) {
    MarchOut o; o.r = o.g = o.b = o.a = 0.0f; o.t = 1.0f; o.incloud = 0;
    float phase = 0.0f;
    if (ray.above) {
        const float ct = fc.ldir[0] * ray.dx + fc.ldir[1] * ray.dy + fc.ldir[2] * ray.dz;   // :158
        phase = fmaxf(fmaxf(henyey_greenstein(ct, 0.6f), henyey_greenstein(ct, fc.hg_g2)), henyey_greenstein(ct, -0.2f));  // :160
    }
    float Tr = 1.0f, alpha = 0.0f, Lr = 0.0f, Lg = 0.0f, Lb = 0.0f;
    float px = ray.px, py = ray.py, pz = ray.pz;
    const float nd = -fc.density;
    const int steps = fc.primary_steps, ls = fc.light_steps;
    for (int i = 0; i < steps; i++) {                                                        // :172
        if (fc.early_eps > 0.0f && CSKY_WAVE_ALL(!ray.above || Tr < fc.early_eps)) break;    // build-side early-out
        if (!ray.above) continue;
        advance(px, py, pz, ray.sx, ray.sy, ray.sz);                                         // :173
        const float hf = height_fraction(length3_shell(px, py, pz));                         // :175
        const float t = sample_density(T, fc, px, py, pz, hf, fc.wpos_x, fc.wpos_y, 0, 0);   // :174,:177
        if (t > 0.0f) {                                                                      // :184
            o.incloud++;
            const float dt = fast_exp(nd * t * ray.ss);                                      // :178
            float lx = px, ly = py, lz = pz, cd = 0.0f;
            for (int j = 0; j < ls; j++) {                                                   // :186
                advance(lx, ly, lz, fc.linc[j][0], fc.linc[j][1], fc.linc[j][2]);            // :187
                const float lhf = height_fraction(length3_shell(lx, ly, lz));                // :188
                cd += sample_density(T, fc, lx, ly, lz, lhf, fc.wpos_x, fc.wpos_y, j > 2 ? j - 2 : 0, j);   // :189-191 (LOD mip-2 clamps at 0)
            }
            {   // distant sample, :195-199
                lx = px; ly = py; lz = pz;
                advance(lx, ly, lz, fc.ldist[0], fc.ldist[1], fc.ldist[2]);
                const float lhf = height_fraction(length3_shell(lx, ly, lz));
                const float ld = sample_density(T, fc, lx, ly, lz, lhf, 0.0f, 0.0f, 3, 5);   // :197 has no weather_pos
                cd += fast_pow(ld, (1.0f - lhf) * 0.8f + 0.5f);                              // :198 (second pow)
            }
            shade_sample(fc, phase, t, hf, dt, cd, Tr, alpha, Lr, Lg, Lb);                       // :202-210
        }
    }
    o.r = Lr; o.g = Lg; o.b = Lb; o.a = sat(alpha); o.t = Tr;                                // :213-214
    return o;
}

}  // namespace csky
