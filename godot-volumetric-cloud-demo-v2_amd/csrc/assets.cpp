// assets.cpp -- host-side asset layer of libcloudsky (no GPU code, no torch).
//
// Replaces what the reference gets from the Godot importers (un-vendored, Godot >= 4.2):
//   * weather.bmp.import:25           2-D texture, no mips
//   * worlnoise.bmp.import:24-27      3-D texture, slices/horizontal=32, vertical=1, mipmaps on
//   * perlworlnoise.tga.import:24-27  3-D texture, slices/horizontal=128, vertical=1, mipmaps on
// cloud_sky/perlworlnoise.tga itself is missing from the reference checkout (.MISSING_LARGE_BLOBS), so
// csky_generate_shape_noise() synthesises a deterministic stand-in with the channel roles the shader
// fixes (clouds.glsl:117-122: R = Perlin-Worley base, G/B/A = Worley fBm octaves).
//
// The generator uses only integer hashing and IEEE +,-,*,/,sqrt, so the bytes are identical on every
// machine/compiler (pinned by SHA-256 in tests/test_assets.py).
#pragma clang fp contract(off)
#include "../../include/cloudsky.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint32_t hash_u32(uint32_t x) {  // lowbias32 finaliser
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
inline uint32_t hash_cell(int x, int y, int z, uint32_t salt) {
    return hash_u32((uint32_t)x * 0x8da6b343U ^ hash_u32((uint32_t)y * 0xd8163841U ^ hash_u32((uint32_t)z * 0xcb1ab31fU ^ salt)));
}
inline float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }  // exact: 24-bit / 2^24
inline int wrap(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

// Inverted tileable Worley (cellular) noise: 1 - distance to the nearest feature point, period `freq` cells.
float worley(float x, float y, float z, int freq, uint32_t salt) {
    float px = x * (float)freq, py = y * (float)freq, pz = z * (float)freq;
    int cx = (int)std::floor(px), cy = (int)std::floor(py), cz = (int)std::floor(pz);
    float best = 1e9f;
    for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
        int gx = cx + dx, gy = cy + dy, gz = cz + dz;
        uint32_t h = hash_cell(wrap(gx, freq), wrap(gy, freq), wrap(gz, freq), salt);
        float fx = (float)gx + u01(h), fy = (float)gy + u01(hash_u32(h + 0x9e3779b9U)), fz = (float)gz + u01(hash_u32(h + 0x3c6ef372U));
        float ex = fx - px, ey = fy - py, ez = fz - pz;
        float d2 = ex * ex + ey * ey + ez * ez;
        if (d2 < best) best = d2;
    }
    float d = std::sqrt(best);
    float v = 1.0f - d;
    return v < 0.0f ? 0.0f : v;
}
float worley_fbm(float x, float y, float z, int freq, uint32_t salt) {
    return worley(x, y, z, freq, salt) * 0.625f + worley(x, y, z, freq * 2, salt + 1) * 0.25f + worley(x, y, z, freq * 4, salt + 2) * 0.125f;
}

// Tileable gradient (Perlin) noise, 12 edge gradients, quintic fade; result roughly in [-1,1].
inline float grad(uint32_t h, float x, float y, float z) {
    switch (h % 12U) {
        case 0: return x + y;  case 1: return -x + y; case 2: return x - y;  case 3: return -x - y;
        case 4: return x + z;  case 5: return -x + z; case 6: return x - z;  case 7: return -x - z;
        case 8: return y + z;  case 9: return -y + z; case 10: return y - z; default: return -y - z;
    }
}
inline float fade(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }
inline float lerp(float a, float b, float t) { return a + (b - a) * t; }
float perlin(float x, float y, float z, int freq, uint32_t salt) {
    float px = x * (float)freq, py = y * (float)freq, pz = z * (float)freq;
    int ix = (int)std::floor(px), iy = (int)std::floor(py), iz = (int)std::floor(pz);
    float fx = px - (float)ix, fy = py - (float)iy, fz = pz - (float)iz;
    float u = fade(fx), v = fade(fy), w = fade(fz);
    float c[2][2][2];
    for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++)
        c[dz][dy][dx] = grad(hash_cell(wrap(ix + dx, freq), wrap(iy + dy, freq), wrap(iz + dz, freq), salt), fx - (float)dx, fy - (float)dy, fz - (float)dz);
    return lerp(lerp(lerp(c[0][0][0], c[0][0][1], u), lerp(c[0][1][0], c[0][1][1], u), v),
                lerp(lerp(c[1][0][0], c[1][0][1], u), lerp(c[1][1][0], c[1][1][1], u), v), w);
}
float perlin_fbm(float x, float y, float z, int freq, int octaves, uint32_t salt) {
    float amp = 1.0f, sum = 0.0f, norm = 0.0f;
    for (int o = 0; o < octaves; o++) {
        sum += amp * perlin(x, y, z, freq << o, salt + 17U * (uint32_t)o);
        norm += amp; amp *= 0.5f;
    }
    return sum / norm;
}
inline float remapf(float v, float omin, float omax, float nmin, float nmax) { return nmin + ((v - omin) / (omax - omin)) * (nmax - nmin); }
inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
inline uint8_t unorm8(float v) { return (uint8_t)(int)(clamp01(v) * 255.0f + 0.5f); }

void shape_rows(uint32_t seed, int n, int z0, int z1, uint8_t* out) {
    const float inv = 1.0f / (float)n;
    for (int z = z0; z < z1; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
        float u = ((float)x + 0.5f) * inv, v = ((float)y + 0.5f) * inv, w = ((float)z + 0.5f) * inv;
        // G/B/A: inverted Worley fBm at rising base frequency (Schneider / "Nubis" layout)
        float g = worley_fbm(u, v, w, 4, seed * 101U + 11U);
        float b = worley_fbm(u, v, w, 8, seed * 101U + 23U);
        float a = worley_fbm(u, v, w, 16, seed * 101U + 37U);
        // R: low-frequency Perlin fBm dilated by the first Worley fBm ("Perlin-Worley"), then a fixed
        // contrast curve calibrated so that the default coverage (0.2) gives mean alpha in 0.3-0.6
        // (the original asset is missing so this is a calibration, not a reconstruction).
        float pf = perlin_fbm(u, v, w, 4, 5, seed * 101U + 53U);        // ~[-0.6, 0.6]
        float p01 = clamp01(pf * 0.9f + 0.5f);
        float pw = remapf(p01, 0.0f, 1.0f, g * 0.55f, 1.0f);            // dilate towards the worley cells
        float r = clamp01((pw - 0.38f) * 1.75f + 0.32f);
        uint8_t* o = out + ((((size_t)z * n + y) * n + x) * 4);
        o[0] = unorm8(r); o[1] = unorm8(g); o[2] = unorm8(b); o[3] = unorm8(a);
    }
}

thread_local char g_asset_err[256];
}  // namespace

extern "C" {

const char* csky_assets_last_error(void) { return g_asset_err; }

int csky_generate_shape_noise(uint32_t seed, int n, uint8_t* out_rgba8) {
    if (!out_rgba8 || n < 8 || (n & (n - 1))) { snprintf(g_asset_err, sizeof g_asset_err, "generate_shape_noise: n must be a power of two >= 8"); return CSKY_ERR_INVALID; }
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw == 0 ? 1 : (hw > 32 ? 32 : hw));
    if (nt > n) nt = n;
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) {
        int z0 = (int)((long)n * t / nt), z1 = (int)((long)n * (t + 1) / nt);
        th.emplace_back(shape_rows, seed, n, z0, z1, out_rgba8);
    }
    for (auto& t : th) t.join();
    return CSKY_OK;
}

size_t csky_mip_offset(int n, int level, int ch) {
    size_t off = 0;
    for (int l = 0; l < level; l++) { size_t m = (size_t)(n >> l); off += m * m * m * (size_t)ch; }
    return off;
}

// 3-D mip chain, 2x2x2 box, integer round-half-up.  `vol` holds level 0 on entry and must have room for
// csky_mip_offset(n, levels, ch) bytes.
int csky_build_mips(uint8_t* vol, int n, int ch, int levels) {
    if (!vol || n < 1 || ch < 1 || levels < 1 || (n >> (levels - 1)) < 1) { snprintf(g_asset_err, sizeof g_asset_err, "build_mips: bad arguments"); return CSKY_ERR_INVALID; }
    for (int l = 1; l < levels; l++) {
        const uint8_t* src = vol + csky_mip_offset(n, l - 1, ch);
        uint8_t* dst = vol + csky_mip_offset(n, l, ch);
        const int ns = n >> (l - 1), nd = n >> l;
        for (int z = 0; z < nd; z++) for (int y = 0; y < nd; y++) for (int x = 0; x < nd; x++) for (int c = 0; c < ch; c++) {
            unsigned s = 0;
            for (int k = 0; k < 8; k++) s += src[((((size_t)(2 * z + (k >> 2))) * ns + (2 * y + ((k >> 1) & 1))) * ns + (2 * x + (k & 1))) * ch + c];
            dst[(((size_t)z * nd + y) * nd + x) * ch + c] = (uint8_t)((s + 4U) >> 3);
        }
    }
    return CSKY_OK;
}

// Uncompressed 24-bpp BMP -> tightly packed RGB8, top row first (what Godot's / PIL's loaders present).
int csky_load_bmp_rgb8(const char* path, int* w, int* h, uint8_t* out, size_t out_capacity) {
    FILE* f = fopen(path, "rb");
    if (!f) { snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: cannot open %s", path); return CSKY_ERR_IO; }
    uint8_t hd[54];
    if (fread(hd, 1, 54, f) != 54 || hd[0] != 'B' || hd[1] != 'M') { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: %s is not a BMP", path); return CSKY_ERR_IO; }
    uint32_t off; int32_t bw, bh; uint16_t bpp; uint32_t comp;
    memcpy(&off, hd + 10, 4); memcpy(&bw, hd + 18, 4); memcpy(&bh, hd + 22, 4); memcpy(&bpp, hd + 28, 2); memcpy(&comp, hd + 30, 4);
    if (bpp != 24 || comp != 0 || bw <= 0 || bh == 0) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: only uncompressed 24-bpp supported (%s)", path); return CSKY_ERR_IO; }
    const bool bottom_up = bh > 0; const int H = bh > 0 ? bh : -bh, W = bw;
    if (w) *w = W; if (h) *h = H;
    if (!out) { fclose(f); return CSKY_OK; }                       // size query
    if (out_capacity < (size_t)W * H * 3) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: output buffer too small"); return CSKY_ERR_INVALID; }
    const size_t stride = ((size_t)W * 3 + 3) & ~(size_t)3;
    std::vector<uint8_t> row(stride);
    fseek(f, (long)off, SEEK_SET);
    for (int r = 0; r < H; r++) {
        if (fread(row.data(), 1, stride, f) != stride) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: truncated %s", path); return CSKY_ERR_IO; }
        uint8_t* o = out + (size_t)(bottom_up ? H - 1 - r : r) * W * 3;
        for (int x = 0; x < W; x++) { o[3 * x + 0] = row[3 * x + 2]; o[3 * x + 1] = row[3 * x + 1]; o[3 * x + 2] = row[3 * x + 0]; }  // BGR -> RGB
    }
    fclose(f);
    return CSKY_OK;
}

// Godot 3-D texture import with slices/horizontal = n, slices/vertical = 1 (worlnoise.bmp.import:26-27,
// perlworlnoise.tga.import:26-27): the strip image is (n*n) x n; voxel (x,y,z) = strip[row y][col n*z + x].
int csky_strip_to_volume(const uint8_t* strip, int n, int ch, uint8_t* vol) {
    if (!strip || !vol || n < 1 || ch < 1) { snprintf(g_asset_err, sizeof g_asset_err, "strip_to_volume: bad arguments"); return CSKY_ERR_INVALID; }
    for (int z = 0; z < n; z++) for (int y = 0; y < n; y++)
        memcpy(vol + (((size_t)z * n + y) * n) * ch, strip + ((size_t)y * n * n + (size_t)n * z) * ch, (size_t)n * ch);
    return CSKY_OK;
}

}  // extern "C"
