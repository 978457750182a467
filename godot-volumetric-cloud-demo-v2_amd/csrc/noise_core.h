// noise_core.h -- the deterministic stand-in generator for the missing cloud_sky/perlworlnoise.tga (128^3 RGBA shape
// noise, perlworlnoise.tga.import:24-27), host+device.  Used by assets.cpp (csky_generate_shape_noise, threads) and by
// kernels.hip (shape_noise_kernel; SURVEY §8f row 2 and README.md:30 TODO 3 "generate the noise on the GPU").
#pragma once
#include "csky_common.h"

namespace csky {
#pragma clang fp contract(off)

CSKY_HD uint32_t hash_u32(uint32_t x) {  // lowbias32 finaliser
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
CSKY_HD uint32_t hash_cell(int x, int y, int z, uint32_t salt) {
    return hash_u32((uint32_t)x * 0x8da6b343U ^ hash_u32((uint32_t)y * 0xd8163841U ^ hash_u32((uint32_t)z * 0xcb1ab31fU ^ salt)));
}
CSKY_HD float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }  // exact: 24-bit / 2^24
CSKY_HD int nwrap(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

// Inverted tileable Worley (cellular) noise: 1 - distance to the nearest feature point, period `freq` cells.
CSKY_HD float worley(float x, float y, float z, int freq, uint32_t salt) {
    float px = x * (float)freq, py = y * (float)freq, pz = z * (float)freq;
    int cx = (int)floorf(px), cy = (int)floorf(py), cz = (int)floorf(pz);
    float best = 1e9f;
    for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
        int gx = cx + dx, gy = cy + dy, gz = cz + dz;
        uint32_t h = hash_cell(nwrap(gx, freq), nwrap(gy, freq), nwrap(gz, freq), salt);
        float fx = (float)gx + u01(h), fy = (float)gy + u01(hash_u32(h + 0x9e3779b9U)), fz = (float)gz + u01(hash_u32(h + 0x3c6ef372U));
        float ex = fx - px, ey = fy - py, ez = fz - pz;
        float d2 = ex * ex + ey * ey + ez * ez;
        if (d2 < best) best = d2;
    }
    float d = sqrtf(best);
    float v = 1.0f - d;
    return v < 0.0f ? 0.0f : v;
}
CSKY_HD float worley_fbm(float x, float y, float z, int freq, uint32_t salt) {
    return worley(x, y, z, freq, salt) * 0.625f + worley(x, y, z, freq * 2, salt + 1) * 0.25f + worley(x, y, z, freq * 4, salt + 2) * 0.125f;
}

// Tileable gradient (Perlin) noise, 12 edge gradients, quintic fade; result roughly in [-1,1].
CSKY_HD float grad(uint32_t h, float x, float y, float z) {
    switch (h % 12U) {
        case 0: return x + y;  case 1: return -x + y; case 2: return x - y;  case 3: return -x - y;
        case 4: return x + z;  case 5: return -x + z; case 6: return x - z;  case 7: return -x - z;
        case 8: return y + z;  case 9: return -y + z; case 10: return y - z; default: return -y - z;
    }
}
CSKY_HD float fade(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }
CSKY_HD float nlerp(float a, float b, float t) { return a + (b - a) * t; }
CSKY_HD float perlin(float x, float y, float z, int freq, uint32_t salt) {
    float px = x * (float)freq, py = y * (float)freq, pz = z * (float)freq;
    int ix = (int)floorf(px), iy = (int)floorf(py), iz = (int)floorf(pz);
    float fx = px - (float)ix, fy = py - (float)iy, fz = pz - (float)iz;
    float u = fade(fx), v = fade(fy), w = fade(fz);
    float c[2][2][2];
    for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++)
        c[dz][dy][dx] = grad(hash_cell(nwrap(ix + dx, freq), nwrap(iy + dy, freq), nwrap(iz + dz, freq), salt), fx - (float)dx, fy - (float)dy, fz - (float)dz);
    return nlerp(nlerp(nlerp(c[0][0][0], c[0][0][1], u), nlerp(c[0][1][0], c[0][1][1], u), v),
                 nlerp(nlerp(c[1][0][0], c[1][0][1], u), nlerp(c[1][1][0], c[1][1][1], u), v), w);
}
CSKY_HD float perlin_fbm(float x, float y, float z, int freq, int octaves, uint32_t salt) {
    float amp = 1.0f, sum = 0.0f, norm = 0.0f;
    for (int o = 0; o < octaves; o++) {
        sum += amp * perlin(x, y, z, freq << o, salt + 17U * (uint32_t)o);
        norm += amp; amp *= 0.5f;
    }
    return sum / norm;
}
CSKY_HD float remapf(float v, float omin, float omax, float nmin, float nmax) { return nmin + ((v - omin) / (omax - omin)) * (nmax - nmin); }
CSKY_HD float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
CSKY_HD uint8_t unorm8(float v) { return (uint8_t)(int)(clamp01(v) * 255.0f + 0.5f); }

// The knobs of the stand-in shape generator (README.md:30 TODO 3: "a noise generator so custom noise can be created and tweaked").  The defaults
// are the calibration every benchmark and parity input uses (seed 1: tests/golden/INPUTS.txt pins its SHA-256); tools/demo_scene.py sweeps them to
// say which one moves which statistic of the rendered sky towards the reference's screenshots (profiles/r05/demo_scene_fit.txt).
// == csky_shape_noise_params of include/cloudsky.h, field for field.
struct ShapeNoiseParams {
    int perlin_freq;        // base period of the R channel's Perlin fBm in cells per volume edge (4: ~3 km features at 97.66 m per texel)
    int perlin_octaves;     // octaves of that fBm (5), amplitude halving per octave
    int worley_freq;        // base frequency of the G channel's Worley fBm (4); B and A use 2x and 4x of it
    float perlin_gain;      // p01 = clamp(fBm * gain + 0.5)                                   (0.9)
    float dilate;           // R = remap(p01, 0, 1, G * dilate, 1): how far the Worley cells carve the Perlin field  (0.55)
    float centre, contrast, offset;   // R = clamp((R - centre) * contrast + offset): the contrast curve        (0.38, 1.75, 0.32)
};
CSKY_HD ShapeNoiseParams shape_noise_defaults() { ShapeNoiseParams p; p.perlin_freq = 4; p.perlin_octaves = 5; p.worley_freq = 4; p.perlin_gain = 0.9f; p.dilate = 0.55f; p.centre = 0.38f; p.contrast = 1.75f; p.offset = 0.32f; return p; }

// One voxel of the stand-in shape volume (R = Perlin-Worley, G/B/A = Worley fBm octaves).  Integer hashing + IEEE
// +,-,*,/,sqrt only, FP contraction off: bit-identical on the host and on the GPU.
CSKY_HD void shape_voxel(uint32_t seed, int n, int x, int y, int z, const ShapeNoiseParams& P, uint8_t o[4]) {
    const float inv = 1.0f / (float)n;
    const float u = ((float)x + 0.5f) * inv, v = ((float)y + 0.5f) * inv, w = ((float)z + 0.5f) * inv;
    // G/B/A: inverted Worley fBm at rising base frequency (Schneider / "Nubis" layout)
    const float g = worley_fbm(u, v, w, P.worley_freq, seed * 101U + 11U);
    const float b = worley_fbm(u, v, w, P.worley_freq * 2, seed * 101U + 23U);
    const float a = worley_fbm(u, v, w, P.worley_freq * 4, seed * 101U + 37U);
    // R: low-frequency Perlin fBm dilated by the first Worley fBm ("Perlin-Worley"), then a contrast curve whose defaults were
    // calibrated so that the default coverage (0.2) gives mean alpha in 0.3-0.6 (the original asset is
    // missing so this is a calibration, not a reconstruction).
    const float pf = perlin_fbm(u, v, w, P.perlin_freq, P.perlin_octaves, seed * 101U + 53U);   // ~[-0.6, 0.6]
    const float p01 = clamp01(pf * P.perlin_gain + 0.5f);
    const float pw = remapf(p01, 0.0f, 1.0f, g * P.dilate, 1.0f);         // dilate towards the worley cells
    const float r = clamp01((pw - P.centre) * P.contrast + P.offset);
    o[0] = unorm8(r); o[1] = unorm8(g); o[2] = unorm8(b); o[3] = unorm8(a);
}
CSKY_HD void shape_voxel(uint32_t seed, int n, int x, int y, int z, uint8_t o[4]) { shape_voxel(seed, n, x, y, z, shape_noise_defaults(), o); }

// One voxel of a generated 32^3 RGB detail volume in the role of cloud_sky/worlnoise.bmp (README.md:30 TODO 3: "generate the noise on
// the GPU"): three tileable inverted-Worley fBm channels of rising frequency.  Calibrated against the shipped worlnoise.bmp (SURVEY A.8;
// tests/test_assets.py compares against the asset itself): channel means 0.71, std 0.11 / 0.11 / 0.14, maxima at 1.0, dominant
// |k| = 2.4 / 4.6 / 7.0 cycles per 32 texels (generated: 2.1 / 4.8 / 6.6), tileable in x, y, z.  Same arithmetic rules as shape_voxel:
// bit-identical on the host and on the GPU.
CSKY_HD void detail_voxel(uint32_t seed, int n, int x, int y, int z, uint8_t o[3]) {
    const float inv = 1.0f / (float)n;
    const float u = ((float)x + 0.5f) * inv, v = ((float)y + 0.5f) * inv, w = ((float)z + 0.5f) * inv;
    const int freq[3] = {2, 5, 7};
    const float centre[3] = {0.4928f, 0.4795f, 0.4801f}, gain[3] = {1.0f, 0.95f, 1.2f};
    for (int c = 0; c < 3; c++) {
        const uint32_t salt = seed * 211U + 7U + (uint32_t)c;
        const float d = (1.0f - worley(u, v, w, freq[c], salt)) * 0.625f + (1.0f - worley(u, v, w, freq[c] * 2, salt + 10U)) * 0.25f +
                        (1.0f - worley(u, v, w, freq[c] * 4, salt + 20U)) * 0.125f;       // fBm of the nearest-feature distance (worley() is 1 - distance, clamped)
        o[c] = unorm8(((1.0f - d) - centre[c]) * gain[c] + 0.712f);
    }
}

}  // namespace csky
