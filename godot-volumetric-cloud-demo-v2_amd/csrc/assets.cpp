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
#include "noise_core.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {

void shape_rows(uint32_t seed, int n, csky::ShapeNoiseParams P, int z0, int z1, uint8_t* out) {
    for (int z = z0; z < z1; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++)
        csky::shape_voxel(seed, n, x, y, z, P, out + ((((size_t)z * n + y) * n + x) * 4));
}

}  // namespace
namespace csky { thread_local char g_asset_err[256]; }   // shared with godot_import.cpp
using csky::g_asset_err;

extern "C" {

const char* csky_assets_last_error(void) { return g_asset_err; }

void csky_shape_noise_default_params(csky_shape_noise_params* p) {
    if (!p) return;
    const csky::ShapeNoiseParams d = csky::shape_noise_defaults();
    static_assert(sizeof(csky_shape_noise_params) == sizeof(csky::ShapeNoiseParams), "csky_shape_noise_params mirrors noise_core.h::ShapeNoiseParams");
    memcpy(p, &d, sizeof d);
}
int csky_check_shape_noise_params(const csky_shape_noise_params* p, int n) {
    // (1) Bounds that hold for every n: frequencies >= 1 -- the lattices wrap with `i % freq` -- and small enough that `freq << octave` and
    //     `worley_freq * 16` cannot leave an int; 1..8 octaves (the fBm loop runs that many times per voxel); every float finite (+inf passes `> 0`, and a
    //     NaN or an infinity reaching the UNORM8 conversion is undefined behaviour).  Nothing throws or aborts across the ABI, whatever the caller passes.
    const int max_freq = 4096;
    if (!p || p->perlin_freq < 1 || p->perlin_freq > max_freq || p->perlin_octaves < 1 || p->perlin_octaves > 8 || p->worley_freq < 1 || p->worley_freq > max_freq ||
        !std::isfinite(p->perlin_gain) || !std::isfinite(p->dilate) || !std::isfinite(p->centre) || !std::isfinite(p->contrast) || !std::isfinite(p->offset) ||
        !(p->perlin_gain > 0.0f) || !(p->dilate >= 0.0f && p->dilate <= 1.0f) || !(p->contrast > 0.0f)) {
        snprintf(g_asset_err, sizeof g_asset_err, "shape noise parameters out of range (1 <= perlin_freq, worley_freq <= %d, 1 <= perlin_octaves <= 8, finite floats, gain > 0, 0 <= dilate <= 1, contrast > 0)", max_freq);
        return CSKY_ERR_INVALID;
    }
    // (2) At least one texel per lattice cell in every octave, or the lattice aliases (64-bit: the bounds above already keep it far from overflow).  Volumes under
    //     64^3 are previews whose default knobs break this rule by design; they keep (1) only.
    if (n >= 64 && (((int64_t)p->perlin_freq << (p->perlin_octaves - 1)) > (int64_t)n || (int64_t)p->worley_freq * 16 > (int64_t)n)) {
        snprintf(g_asset_err, sizeof g_asset_err, "shape noise parameters alias at n = %d (perlin_freq << (octaves - 1) <= n, worley_freq * 16 <= n)", n);
        return CSKY_ERR_INVALID;
    }
    return CSKY_OK;
}
int csky_generate_shape_noise(uint32_t seed, int n, uint8_t* out_rgba8) {
    csky_shape_noise_params p; csky_shape_noise_default_params(&p);
    return csky_generate_shape_noise_tuned(seed, n, &p, out_rgba8);
}
int csky_generate_shape_noise_tuned(uint32_t seed, int n, const csky_shape_noise_params* params, uint8_t* out_rgba8) {
    if (!out_rgba8 || n < 8 || n > 1024 || (n & (n - 1))) { snprintf(g_asset_err, sizeof g_asset_err, "generate_shape_noise: n must be a power of two in [8, 1024]"); return CSKY_ERR_INVALID; }
    csky::ShapeNoiseParams P = csky::shape_noise_defaults();
    if (params) { int rc = csky_check_shape_noise_params(params, n); if (rc) return rc; memcpy(&P, params, sizeof P); }
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw == 0 ? 1 : (hw > 32 ? 32 : hw));
    if (nt > n) nt = n;
    std::vector<std::thread> th;
    int done_to = 0;                                             // slices [0, done_to) are owned by a started thread
    try {                                                        // (thread creation can throw: nothing crosses the ABI; the rest is rendered here)
        for (int t = 0; t < nt; t++) {
            int z0 = (int)((long)n * t / nt), z1 = (int)((long)n * (t + 1) / nt);
            th.emplace_back(shape_rows, seed, n, P, z0, z1, out_rgba8);
            done_to = z1;
        }
    } catch (...) {}
    if (done_to < n) shape_rows(seed, n, P, done_to, n, out_rgba8);
    for (auto& t : th) t.join();
    return CSKY_OK;
}

int csky_generate_detail_noise(uint32_t seed, int n, uint8_t* out_rgb8) {
    if (!out_rgb8 || n < 8 || n > 1024 || (n & (n - 1))) { snprintf(g_asset_err, sizeof g_asset_err, "generate_detail_noise: n must be a power of two in [8, 1024]"); return CSKY_ERR_INVALID; }
    for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) csky::detail_voxel(seed, n, x, y, z, out_rgb8 + ((((size_t)z * n + y) * n + x) * 3));
    return CSKY_OK;
}

size_t csky_mip_offset(int n, int level, int ch) {
    if (n < 1 || ch < 1) return 0;
    if (level > 31) level = 31;                                  // n >> l is 0 from l = 31 at the latest (a shift by >= 32 is undefined)
    size_t off = 0;
    for (int l = 0; l < level; l++) { size_t m = (size_t)(n >> l); off += m * m * m * (size_t)ch; }
    return off;
}

// 3-D mip chain, 2x2x2 box, integer round-half-up.  `vol` holds level 0 on entry and must have room for
// csky_mip_offset(n, levels, ch) bytes.
int csky_build_mips(uint8_t* vol, int n, int ch, int levels) {
    if (!vol || n < 1 || ch < 1 || levels < 1 || levels > 31 || (n >> (levels - 1)) < 1) { snprintf(g_asset_err, sizeof g_asset_err, "build_mips: bad arguments"); return CSKY_ERR_INVALID; }
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
static int load_bmp_impl(const char* path, int* w, int* h, uint8_t* out, size_t out_capacity) {
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) { snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: cannot open %s", path ? path : "(null)"); return CSKY_ERR_IO; }
    uint8_t hd[54];
    if (fread(hd, 1, 54, f) != 54 || hd[0] != 'B' || hd[1] != 'M') { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: %s is not a BMP", path); return CSKY_ERR_IO; }
    uint32_t off; int32_t bw, bh; uint16_t bpp; uint32_t comp;
    memcpy(&off, hd + 10, 4); memcpy(&bw, hd + 18, 4); memcpy(&bh, hd + 22, 4); memcpy(&bpp, hd + 28, 2); memcpy(&comp, hd + 30, 4);
    // (sizes beyond 65536 are no texture of this path and would make `-bh`, the row buffer and the index arithmetic below a liability)
    if (bpp != 24 || comp != 0 || bw <= 0 || bw > 65536 || bh == 0 || bh > 65536 || bh < -65536) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: only uncompressed 24-bpp supported (%s)", path); return CSKY_ERR_IO; }
    const bool bottom_up = bh > 0; const int H = bh > 0 ? bh : -bh, W = bw;
    if (w) *w = W; if (h) *h = H;
    if (!out) { fclose(f); return CSKY_OK; }                       // size query
    if (out_capacity < (size_t)W * H * 3) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: output buffer too small"); return CSKY_ERR_INVALID; }
    const size_t stride = ((size_t)W * 3 + 3) & ~(size_t)3;
    std::vector<uint8_t> row;
    try { row.resize(stride); } catch (...) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: out of memory"); return CSKY_ERR_IO; }
    if (fseek(f, (long)off, SEEK_SET) != 0) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: bad pixel-data offset in %s", path); return CSKY_ERR_IO; }
    for (int r = 0; r < H; r++) {
        if (fread(row.data(), 1, stride, f) != stride) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_bmp: truncated %s", path); return CSKY_ERR_IO; }
        uint8_t* o = out + (size_t)(bottom_up ? H - 1 - r : r) * W * 3;
        for (int x = 0; x < W; x++) { o[3 * x + 0] = row[3 * x + 2]; o[3 * x + 1] = row[3 * x + 1]; o[3 * x + 2] = row[3 * x + 0]; }  // BGR -> RGB
    }
    fclose(f);
    return CSKY_OK;
}

int csky_load_bmp_rgb8(const char* path, int* w, int* h, uint8_t* out, size_t out_capacity) { return load_bmp_impl(path, w, h, out, out_capacity); }

// Truevision TGA (types 2 = uncompressed and 10 = RLE true colour, 24 or 32 bpp, either origin) -> tightly packed RGBA8,
// top row first: the container of the reference's shape noise cloud_sky/perlworlnoise.tga (16384 x 128, 128 slices;
// missing from the reference checkout but loadable here when a user has it).
int csky_load_tga_rgba8(const char* path, int* w, int* h, uint8_t* out, size_t out_capacity) {
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) { snprintf(g_asset_err, sizeof g_asset_err, "load_tga: cannot open %s", path ? path : "(null)"); return CSKY_ERR_IO; }
    uint8_t hd[18];
    if (fread(hd, 1, 18, f) != 18) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_tga: %s is truncated", path); return CSKY_ERR_IO; }
    const int id_len = hd[0], cmap_type = hd[1], type = hd[2], W = hd[12] | (hd[13] << 8), H = hd[14] | (hd[15] << 8), bpp = hd[16];
    const bool top_down = (hd[17] & 0x20) != 0, right_left = (hd[17] & 0x10) != 0;
    if (cmap_type != 0 || (type != 2 && type != 10) || (bpp != 24 && bpp != 32) || W <= 0 || H <= 0 || right_left) {
        fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_tga: only true-colour 24/32-bpp TGA (type 2/10) supported (%s)", path); return CSKY_ERR_IO;
    }
    if (w) *w = W; if (h) *h = H;
    if (!out) { fclose(f); return CSKY_OK; }                       // size query
    if (out_capacity < (size_t)W * H * 4) { fclose(f); snprintf(g_asset_err, sizeof g_asset_err, "load_tga: output buffer too small"); return CSKY_ERR_INVALID; }
    fseek(f, 18 + id_len, SEEK_SET);
    const int bytes = bpp / 8;
    const size_t npx = (size_t)W * H;
    size_t i = 0;
    uint8_t px[4] = {0, 0, 0, 255};
    auto put = [&](size_t idx) {
        const size_t row = idx / W, col = idx % W;
        uint8_t* o = out + ((top_down ? row : (size_t)H - 1 - row) * W + col) * 4;
        o[0] = px[2]; o[1] = px[1]; o[2] = px[0]; o[3] = bytes == 4 ? px[3] : 255;      // BGRA -> RGBA
    };
    bool ok = true;
    if (type == 2) {
        for (; i < npx && ok; i++) { ok = fread(px, 1, bytes, f) == (size_t)bytes; if (ok) put(i); }
    } else {
        while (i < npx && ok) {
            int c = fgetc(f);
            if (c == EOF) { ok = false; break; }
            const size_t run = (size_t)(c & 0x7f) + 1;
            if (i + run > npx) { ok = false; break; }
            if (c & 0x80) { ok = fread(px, 1, bytes, f) == (size_t)bytes; for (size_t k = 0; k < run && ok; k++) put(i + k); }
            else for (size_t k = 0; k < run && ok; k++) { ok = fread(px, 1, bytes, f) == (size_t)bytes; if (ok) put(i + k); }
            i += run;
        }
    }
    fclose(f);
    if (!ok) { snprintf(g_asset_err, sizeof g_asset_err, "load_tga: %s is truncated or corrupt", path); return CSKY_ERR_IO; }
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
