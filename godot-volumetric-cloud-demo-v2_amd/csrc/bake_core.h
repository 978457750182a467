// bake_core.h -- one texel of each device texture layout (csky_common.h) from the 8-bit mip chains, host+device:
// the GPU bake kernels (kernels.hip) and the host loops of bake.h (tests/hostsim) run the same code, so the two bakes are
// byte-identical by construction (and tests/test_gpu_round2.py compares them).  Also the 2x2x2 box mip (Godot's
// mipmaps/generate=true on 3-D textures, perlworlnoise.tga.import:24 / worlnoise.bmp.import:24, as (sum + 4) >> 3).
#pragma once
#include "csky_common.h"

namespace csky {

// byte offset of mip level `level` inside a chain of n^3 x ch texels (== csky_mip_offset, cloudsky.h)
CSKY_HD size_t chain_offset(int n, int level, int ch) {
    size_t off = 0;
    for (int l = 0; l < level; l++) { const size_t m = (size_t)(n >> l); off += m * m * m * (size_t)ch; }
    return off;
}
// one destination texel-channel of level l from level l-1: box filter of the 2x2x2 parents, round half up
CSKY_HD uint8_t mip_texel(const uint8_t* __restrict__ src, int ns, int ch, int x, int y, int z, int c) {
    unsigned s = 0;
    for (int k = 0; k < 8; k++) s += src[((((size_t)(2 * z + (k >> 2))) * ns + (2 * y + ((k >> 1) & 1))) * ns + (2 * x + (k & 1))) * ch + c];
    return (uint8_t)((s + 4U) >> 3);
}

// fp16 pair {lo, hi} of two integer coefficients; `inexact` counts values fp16 cannot hold exactly (|v| > 2048 and odd, ...)
CSKY_HD uint32_t hpair_i(int lo, int hi, unsigned& inexact) {
    const uint16_t a = f2h((float)lo), b = f2h((float)hi);
    if (h2f(a) != (float)lo) ++inexact;
    if (h2f(b) != (float)hi) ++inexact;
    return (uint32_t)a | ((uint32_t)b << 16);
}
// polynomial-cell coefficients (csky_common.h) of the cell whose corners are v[x | y<<1 | z<<2]
CSKY_HD void cell_coeffs(const int v[8], int c[8]) {
    c[0] = v[0]; c[1] = v[1] - v[0]; c[2] = v[2] - v[0]; c[3] = v[3] - v[2] - v[1] + v[0];
    c[4] = v[4] - v[0]; c[5] = v[5] - v[4] - v[1] + v[0]; c[6] = v[6] - v[4] - v[2] + v[0];
    c[7] = v[7] - v[6] - v[5] + v[4] - v[3] + v[2] + v[1] - v[0];
}

// where the cell (x,y,z) of an n^3 level is stored inside the level (csky_common.h: x fastest, then y, then z)
CSKY_HD size_t shape_cell_index(int n, int x, int y, int z) { return ((size_t)z * n + y) * n + x; }
// shape texel (x,y,z) of a level with n^3 RGBA8 texels at `src` (REPEAT addressing)
CSKY_HD ShapeTexel bake_shape_texel(const uint8_t* __restrict__ src, int n, int x, int y, int z, unsigned& inexact) {
    int vr[8], vf[8], cr[8], cf[8];
    for (int k = 0; k < 8; k++) {
        const int xx = (x + (k & 1)) % n, yy = (y + ((k >> 1) & 1)) % n, zz = (z + (k >> 2)) % n;
        const uint8_t* t = src + (((size_t)zz * n + yy) * n + xx) * 4;
        vr[k] = t[0]; vf[k] = 5 * t[1] + 2 * t[2] + t[3];                                  // fbm numerator (clouds.glsl:118 x 8)
    }
    cell_coeffs(vr, cr); cell_coeffs(vf, cf);
    ShapeTexel o;
#if CSKY_SHAPE_POLY == 1
    o = uint2{hpair_i(cr[0], cr[1], inexact), hpair_i(cf[0], cf[1], inexact)};
#elif CSKY_SHAPE_POLY == 2
    o = uint4{hpair_i(cr[0], cr[1], inexact), hpair_i(cr[2], cr[3], inexact), hpair_i(cf[0], cf[1], inexact), hpair_i(cf[2], cf[3], inexact)};
#else
    o.r = uint4{hpair_i(cr[0], cr[1], inexact), hpair_i(cr[2], cr[3], inexact), hpair_i(cr[4], cr[5], inexact), hpair_i(cr[6], cr[7], inexact)};
    o.f = uint4{hpair_i(cf[0], cf[1], inexact), hpair_i(cf[2], cf[3], inexact), hpair_i(cf[4], cf[5], inexact), hpair_i(cf[6], cf[7], inexact)};
#endif
    return o;
}
// hfbm numerator of one RGB8 detail texel (clouds.glsl:133 x 8)
CSKY_HD int detail_numerator(const uint8_t* __restrict__ src, int n, int x, int y, int z) {
    const uint8_t* t = src + (((size_t)(z % n) * n + (y % n)) * n + (x % n)) * 3;
    return 5 * t[0] + 2 * t[1] + t[2];
}
CSKY_HD uint4 bake_detail_texel(const uint8_t* __restrict__ src, int n, int x, int y, int z, unsigned& inexact) {
    int v[8], c[8];
    for (int k = 0; k < 8; k++) v[k] = detail_numerator(src, n, x + (k & 1), y + ((k >> 1) & 1), z + (k >> 2));
    cell_coeffs(v, c);
    return uint4{hpair_i(c[0], c[1], inexact), hpair_i(c[2], c[3], inexact), hpair_i(c[4], c[5], inexact), hpair_i(c[6], c[7], inexact)};
}
// weather texel (x,y): xy cells of R (cloud type) and B (coverage); G is never read (clouds.glsl:121,123)
CSKY_HD uint4 bake_weather_texel(const uint8_t* __restrict__ rgb, int x, int y, unsigned& inexact) {
    const int n = WEATHER_N;
    uint4 q = uint4{0u, 0u, 0u, 0u};
    for (int c = 0; c < 2; c++) {
        const int k = 2 * c;
        const int v00 = rgb[(((size_t)(y % n)) * n + (x % n)) * 3 + k], v10 = rgb[(((size_t)(y % n)) * n + ((x + 1) % n)) * 3 + k];
        const int v01 = rgb[(((size_t)((y + 1) % n)) * n + (x % n)) * 3 + k], v11 = rgb[(((size_t)((y + 1) % n)) * n + ((x + 1) % n)) * 3 + k];
        const uint32_t p0 = hpair_i(v00, v10 - v00, inexact), p1 = hpair_i(v01 - v00, v11 - v01 - v10 + v00, inexact);
        if (c == 0) { q.x = p0; q.y = p1; } else { q.z = p0; q.w = p1; }
    }
    return q;
}


// ---- exact cells (round 4): the same polynomial cells with fp32 coefficients, for textures some of whose finite differences do not fit fp16
// (|c| > 2048 and not a multiple of the fp16 spacing there: white noise, checkerboards).  Integers up to 8 x 2040 are exact in fp32, so these
// cells are exact for ANY 8-bit input; for inputs that also bake exactly in fp16 the two forms filter bit-identically (v_fma_mix_f32 widens
// exactly what is stored here).  Twice the bytes per tap, so only used when csky_noise_inexact_coeffs() > 0 (or on request, csky_set_exact_cells).
//   weather: 2 x float4 per texel  {r c0..c3}{b c0..c3};  detail: 2 x float4  {c0..c3}{c4..c7};  shape: 4 x float4  {r c0..c3}{r c4..c7}{f c0..c3}{f c4..c7}
CSKY_HD void bake_shape_texel32(const uint8_t* __restrict__ src, int n, int x, int y, int z, float4 out[4]) {
    int vr[8], vf[8], cr[8], cf[8];
    for (int k = 0; k < 8; k++) {
        const int xx = (x + (k & 1)) % n, yy = (y + ((k >> 1) & 1)) % n, zz = (z + (k >> 2)) % n;
        const uint8_t* t = src + (((size_t)zz * n + yy) * n + xx) * 4;
        vr[k] = t[0]; vf[k] = 5 * t[1] + 2 * t[2] + t[3];
    }
    cell_coeffs(vr, cr); cell_coeffs(vf, cf);
    out[0] = float4{(float)cr[0], (float)cr[1], (float)cr[2], (float)cr[3]}; out[1] = float4{(float)cr[4], (float)cr[5], (float)cr[6], (float)cr[7]};
    out[2] = float4{(float)cf[0], (float)cf[1], (float)cf[2], (float)cf[3]}; out[3] = float4{(float)cf[4], (float)cf[5], (float)cf[6], (float)cf[7]};
}
CSKY_HD void bake_detail_texel32(const uint8_t* __restrict__ src, int n, int x, int y, int z, float4 out[2]) {
    int v[8], c[8];
    for (int k = 0; k < 8; k++) v[k] = detail_numerator(src, n, x + (k & 1), y + ((k >> 1) & 1), z + (k >> 2));
    cell_coeffs(v, c);
    out[0] = float4{(float)c[0], (float)c[1], (float)c[2], (float)c[3]}; out[1] = float4{(float)c[4], (float)c[5], (float)c[6], (float)c[7]};
}
CSKY_HD void bake_weather_texel32(const uint8_t* __restrict__ rgb, int x, int y, float4 out[2]) {
    const int n = WEATHER_N;
    for (int c = 0; c < 2; c++) {
        const int k = 2 * c;
        const int v00 = rgb[(((size_t)(y % n)) * n + (x % n)) * 3 + k], v10 = rgb[(((size_t)(y % n)) * n + ((x + 1) % n)) * 3 + k];
        const int v01 = rgb[(((size_t)((y + 1) % n)) * n + (x % n)) * 3 + k], v11 = rgb[(((size_t)((y + 1) % n)) * n + ((x + 1) % n)) * 3 + k];
        out[c] = float4{(float)v00, (float)(v10 - v00), (float)(v01 - v00), (float)(v11 - v01 - v10 + v00)};
    }
}

}  // namespace csky
