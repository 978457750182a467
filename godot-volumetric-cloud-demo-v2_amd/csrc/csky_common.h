// csky_common.h -- types and helpers shared by the HIP kernels, the host API and the host-compiled
// kernel-core unit test (tests/hostsim).  Plain C++17; compiles under hipcc (host+device) and g++.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CSKY_HD __host__ __device__ __forceinline__
#define CSKY_D __device__ __forceinline__
#else
#define CSKY_HD inline
#define CSKY_D inline
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct float4 { float x, y, z, w; };
#endif

namespace csky {

// ---- geometry literals, clouds.glsl:43-47 -------------------------------------------------------
constexpr float G_RADIUS = 6000000.0f;
constexpr float SKY_B_RADIUS = 6001500.0f;
constexpr float SKY_T_RADIUS = 6004000.0f;
constexpr float CLOUD_PI = 3.141592f;  // truncated literal of clouds.glsl:47, kept on purpose

constexpr int SHAPE_N = 128, SHAPE_LEVELS = 8;   // perlworlnoise.tga.import:24-27 (128 slices, mips on)
constexpr int DETAIL_N = 32, DETAIL_LEVELS = 6;  // worlnoise.bmp.import:24-27 (32 slices, mips on)
constexpr int WEATHER_N = 512;                   // weather.bmp.import:25 (no mips)
constexpr int DETAIL_CHAIN_TEXELS = 32768 + 4096 + 512 + 64 + 8 + 1;   // all six levels of the 32^3 detail volume

// ---- device texture layouts (baked by bake.h) ----------------------------------------
// Texel values are stored as fp16 (small integers, exact) so that v_fma_mix_f32 widens them for free inside the
// filtering FMAs.  Each texel stores the coefficients of the bi-/trilinear interpolant of the cell that STARTS at it
// ("polynomial cell"): with corner values v000..v111 (index = x,y,z bit) and fractions (fx,fy,fz)
//      v(fx,fy,fz) = (c0 + c1 fx) + fy (c2 + c3 fx) + fz [ (c4 + c5 fx) + fy (c6 + c7 fx) ]
//      c0 = v000, c1 = d_x, c2 = d_y, c3 = d_xy, c4 = d_z, c5 = d_xz, c6 = d_yz, c7 = d_xyz   (finite differences)
// which is the reference sampler's nested a + (b - a) f expanded once on the host.  Every dword is an fp16 pair
// {c_even, c_odd}: one v_fma_mix_f32 gives (c_even + c_odd fx), so a trilinear tap is 4 + 2 + 1 = 7 FMAs (nested lerps
// on x-pairs: 10) and a bilinear one 2 + 1 = 3 (4).  Differences are integers; fp16 holds integers up to 2048 exactly
// and the bake COUNTS coefficients that do not fit (csky_noise_inexact_coeffs(); 0 for every shipped texture, a
// synthetic checkerboard can exceed it and then carries a relative 2^-11 error on that coefficient).
// weather: per texel uint4 {r: c0c1, c2c3, b: c0c1, c2c3} (G is never read: clouds.glsl:121,123) -> ONE 16-byte load / tap
// detail : per texel uint4 {c0c1, c2c3, c4c5, c6c7} of the hfbm numerator 5r+2g+b (clouds.glsl:133)    -> ONE 16-byte load / tap
// shape  : CSKY_SHAPE_POLY selects the cell rank for {r, fbm numerator 5g+2b+a (clouds.glsl:118)}:
//          1: uint2 {r c0c1, f c0c1}: 8 B/texel, 4 x 8-byte loads / tap (x cells; y and z lerped in the kernel)
//          2: uint4 {r c0c1, r c2c3, f c0c1, f c2c3}: 16 B/texel, 2 x 16-byte loads / tap (xy cells; z lerped in the kernel)
//          3: 2 x uint4 {r c0..c7}{f c0..c7}: 32 B/texel, 2 x 16-byte loads from ONE address / tap (xyz cells)
#ifndef CSKY_SHAPE_POLY
#define CSKY_SHAPE_POLY 3   // measured on MI355X, C3 frame: rank 1 2.89 ms, rank 2 2.56 ms, rank 3 2.52 ms (profiles/r01/texture_cell_rank_ab.txt)
#endif
// Cells of a level are stored x fastest, then y, then z (the texture's own slice order).  Round 2 measured two alternatives on MI355X with
// TCP / TCC counters (profiles/r02/shape_layout_ab.txt), frames bit-identical in all three:
//   x, then z, then y (a wavefront's rays sit at one altitude = texture v, and spread over u / w)   2.031 / 1.792 ms vs 2.040 / 1.795: no difference,
//                                                                                                 L1 hit 72.6 % and L2 hit 78.0 % unchanged
//   4x4x4-cell bricks (2 KB, what texture hardware does)                                          2.081 / 1.819 ms: L1 hit unchanged (72.6 %), L2 hit
//                                                                                                 77.9 -> 78.2 %, fabric bytes -1.5 %, VALU +5.2 %
//   the two channels' cells in two planes of dense 16-byte cells instead of one 32-byte texel                2.22 / 1.97 ms vs 2.03 / 1.71: a tap then
//                                                                                                 touches two cache lines instead of one
//   (and marking the shape loads non-temporal to spare the L1 for the other textures: 2.48 ms per frame: the cells ARE re-used)
// The L1 hit rate is set by the 4 cells a 128-byte line holds and by how far apart a quad's rays land, not by the slice strides; the kernel
// is VALU-issue bound (bench.py roofline), so the bricks' six extra half-rate integer instructions per tap cost more than they save.
#if CSKY_SHAPE_POLY == 1
typedef uint2 ShapeTexel;
#elif CSKY_SHAPE_POLY == 2
typedef uint4 ShapeTexel;
#else
struct ShapeTexel { uint4 r, f; };
#endif
struct TexSet {
    static constexpr bool cell32 = false;   // fp16-pair cells (the layouts above)
    const ShapeTexel* shape;   // all mip levels back to back; level l starts at shape_level_offset(l) texels (cloud_core.h)
    const uint4* detail;    // all mip levels back to back; level l starts at detail_level_offset(l)
    const uint4* weather;   // 512*512
    const float4* sky;      // sky LUT, fp16-rounded values widened to float, sky_w x sky_h
    int sky_w, sky_h;
    const uint16_t* detail_h;     // global: unpacked fp16 numerators of the detail chain (source of the LDS copy), 37 449 texels
    const uint16_t* detail_lds;   // LDS copy of detail_h inside the "lds" kernel variant, else nullptr
    float detail_lod5;      // the single texel of detail LOD 5 (1x1x1) as hfbm = (5r+2g+b)/(8*255): the filtered value of EVERY tap at that level
#ifdef CSKY_BRICK_BOUND
    const float* brick;     // experiment build only (round 3): per 8^3-texel brick of shape level 0 (+1 apron) an upper bound of base_cloud
#endif
};

// The same set with EXACT cells (fp32 coefficients, bake_core.h): bound instead of TexSet when some coefficient of the textures does not fit fp16
struct TexSet32 : TexSet {
    static constexpr bool cell32 = true;
    const float4* shape32;     // 4 x float4 per texel, levels packed like `shape`
    const float4* detail32;    // 2 x float4 per texel
    const float4* weather32;   // 2 x float4 per texel
};

// Ray-invariant per-frame constants, computed once per frame by frame_setup() (clouds.glsl:143-170).
struct FrameConsts {
    float tex_w, tex_h;
    int upd_x, upd_y;                 // ivec2(update_position), clouds.glsl:260
    float cloud_off_x, cloud_off_z;   // 20*cloud_pos*0.6, clouds.glsl:114
    float det_off_x, det_off_z, det_off_y;  // detailed_pos*40, time*40, clouds.glsl:128-129
    float wpos_x, wpos_y;             // weather_pos, clouds.glsl:170
    float ldir[3];                    // normalize(LIGHT_DIRECTION), clouds.glsl:150
    float linc[6][3];                 // (ldir + RANDOM_VECTORS[j]*j)*lss, clouds.glsl:187
    float ldist[3];                   // ldir*18*lss, clouds.glsl:195
    float hg_g2;                      // 0.4 - 1.4*ldir.y, clouds.glsl:160
    float sun_c[3], amb_c[3], gnd_c[3];  // clouds.glsl:163-167
    float density, coverage;          // clouds.glsl:37-38
    float cov255;                     // coverage / 255: weather taps stay on the texel scale 0..255, the 1/255 is folded into their consumers
    int primary_steps, light_steps;   // clouds.glsl:228 (128), :186 (6)
    float steps_f;
    float early_eps;                  // wave early-out threshold on T (0 = off; not in the reference)
    float hf_lo, hf_hi;               // height-fraction window outside which density() is provably 0 (bake.h height_window)
    int ct_mode;                      // cloud-type range of the bound weather map: 1 = every texel >= 0.5, 2 = every texel < 0.5, 0 = mixed (density_height_gradient)
};

// Which rows a launch renders (cloudsky.h csky_bands) + output addressing.
struct RenderGeom {
    int tile_w;        // pixels per row rendered (gl_GlobalInvocationID.x range)
    int band_rows, first_band, band_stride, n_bands;
    uint32_t pitch_px; // output row pitch in pixels (8 bytes each)
    int out_full;      // 0: compact output (row = band-local row); 1: row = the pixel row inside the tile (multi-GPU: every device stores
                       // its bands at their place in ONE frame, csky_multi_render_clouds_device)
};

struct CloudParams {  // == csky_cloud_params (clouds.glsl:18-40)
    float texture_size[2], update_position[2], cloud_pos[2], detailed_pos[2], weather_pos[2], pad1[2];
    float ground_color[4], LIGHT_DIRECTION[3], LIGHT_ENERGY, LIGHT_COLOR[3], time, pad2, density, cloud_coverage, time_offset;
};
static_assert(sizeof(CloudParams) == 112, "push-constant block must be 112 bytes (clouds.glsl:18-40)");

// ---- IEEE binary16 <-> float, round-to-nearest-even, identical on host and device ---------------
CSKY_HD uint16_t f2h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, em = x & 0x7fffffffu;
    if (em >= 0x7f800000u) return (uint16_t)(sign | (em > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (em >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (em < 0x33000001u) return (uint16_t)sign;
    const int e = (int)(em >> 23) - 127;
    const uint32_t m = (em & 0x7fffffu) | 0x800000u;
    const int shift = (e < -14) ? (13 + (-14 - e)) : 13;
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    const uint32_t h = (e < -14) ? q : (((uint32_t)(e + 15) << 10) + (q - 0x400u));
    return (uint16_t)(sign | h);
}
CSKY_HD float h2f(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int s = 0; while (!(m & 0x400u)) { m <<= 1; s++; } x = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ffu) << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}

CSKY_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
CSKY_HD float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
CSKY_HD float lerpf(float a, float b, float f) { return a + (b - a) * f; }
CSKY_HD float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

}  // namespace csky
