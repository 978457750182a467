// godot_import.cpp -- reading what Godot's importer wrote (host only, no GPU code).
//
// The reference's three noise inputs are imported as VRAM-compressed, high-quality textures (weather.bmp.import:19-20,
// worlnoise.bmp.import:19-20, perlworlnoise.tga.import:19-20: compress/mode=2, compress/high_quality=true => BPTC / BC7 blocks,
// path.bptc="res://.godot/imported/<name>-<md5>.bptc.ctex[3d]"), the two volumes with the importer's own mip chains.  The texels the
// reference's samplers return are therefore the DECODED BC7 blocks of those files, not the bytes of the .bmp/.tga.  Re-encoding
// cannot be reproduced here (it depends on the engine's encoder), decoding is fixed by the format:
//   * csky_decode_bc7       BC7 (BPTC RGBA UNORM) blocks -> RGBA8, all eight modes; checked block by block against an
//                           independent decoder (tests/test_godot_import.py); tables: bc7_tables.h (tools/derive_bc7_tables.py)
//   * csky_load_ctex        the CompressedTexture2D container ("GST2"): level 0 (and the stored mips) -> RGBA8
//   * csky_load_ctex3d      the CompressedTexture3D container ("GSTL"): slices of level 0 followed by the importer's mip slices -> RGBA8 volume chain
// The container layouts are written from the engine's documented loader (scene/resources/compressed_texture.cpp of Godot 4.2); no
// imported file exists in the reference checkout (.godot/ is not versioned), so the container readers are tested against files
// written by the test-suite's own writer only -- the BC7 decoder is what is independently pinned.
#include "../../include/cloudsky.h"
#include "bc7_tables.h"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <vector>

namespace csky { extern thread_local char g_asset_err[256]; }
using csky::g_asset_err;

namespace {

struct BitReader {
    const uint8_t* p; int pos = 0;
    explicit BitReader(const uint8_t* b) : p(b) {}
    uint32_t get(int n) {                                      // LSB first, n <= 8
        uint32_t v = 0;
        for (int i = 0; i < n; i++, pos++) v |= (uint32_t)((p[pos >> 3] >> (pos & 7)) & 1u) << i;
        return v;
    }
};

// per mode: subsets, partition bits, rotation bits, index-selection bit, colour bits, alpha bits, per-end-point p-bits, shared p-bits, index bits, secondary index bits
struct ModeInfo { int ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2; };
const ModeInfo kModes[8] = {
    {3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0}, {2, 6, 0, 0, 7, 0, 1, 0, 2, 0},
    {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2}, {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0}};
const uint8_t kW2[4] = {0, 21, 43, 64};
const uint8_t kW3[8] = {0, 9, 18, 27, 37, 46, 55, 64};
const uint8_t kW4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};
inline const uint8_t* weights(int bits) { return bits == 2 ? kW2 : (bits == 3 ? kW3 : kW4); }
inline uint8_t lerp6(int a, int b, int w) { return (uint8_t)(((64 - w) * a + w * b + 32) >> 6); }

void decode_block(const uint8_t* blk, uint8_t out[16][4]) {
    int mode = 0;
    while (mode < 8 && !((blk[0] >> mode) & 1)) mode++;
    if (mode == 8) { memset(out, 0, 64); return; }             // reserved encoding: transparent black
    const ModeInfo& m = kModes[mode];
    BitReader br(blk);
    br.get(mode + 1);
    const int part = m.pb ? (int)br.get(m.pb) : 0;
    const int rot = m.rb ? (int)br.get(m.rb) : 0;
    const int isel = m.isb ? (int)br.get(1) : 0;
    int ep[6][4];                                              // [2 * subset + end][channel]
    const int ne = 2 * m.ns;
    for (int c = 0; c < 3; c++) for (int e = 0; e < ne; e++) ep[e][c] = (int)br.get(m.cb);
    for (int e = 0; e < ne; e++) ep[e][3] = m.ab ? (int)br.get(m.ab) : 255;
    int cbits = m.cb, abits = m.ab;
    if (m.epb) {
        for (int e = 0; e < ne; e++) { const int pbit = (int)br.get(1); for (int c = 0; c < 3; c++) ep[e][c] = (ep[e][c] << 1) | pbit; if (m.ab) ep[e][3] = (ep[e][3] << 1) | pbit; }
        cbits++; if (m.ab) abits++;
    } else if (m.spb) {
        for (int s = 0; s < m.ns; s++) { const int pbit = (int)br.get(1); for (int e = 2 * s; e < 2 * s + 2; e++) for (int c = 0; c < 3; c++) ep[e][c] = (ep[e][c] << 1) | pbit; }
        cbits++;
    }
    for (int e = 0; e < ne; e++) {                             // widen to 8 bits: shift up, replicate the top bits below
        for (int c = 0; c < 3; c++) { int v = ep[e][c] << (8 - cbits); ep[e][c] = v | (v >> cbits); }
        if (m.ab) { int v = ep[e][3] << (8 - abits); ep[e][3] = v | (v >> abits); }
    }
    uint8_t subset[16];
    int anchor[3] = {0, -1, -1};
    for (int i = 0; i < 16; i++) subset[i] = m.ns == 1 ? 0 : (m.ns == 2 ? kBc7Partition2[part][i] : kBc7Partition3[part][i]);
    if (m.ns == 2) anchor[1] = kBc7Anchor2[part];
    if (m.ns == 3) { anchor[1] = kBc7Anchor3a[part]; anchor[2] = kBc7Anchor3b[part]; }
    int idx1[16], idx2[16];
    for (int i = 0; i < 16; i++) {
        const bool is_anchor = i == anchor[subset[i]];
        idx1[i] = (int)br.get(is_anchor ? m.ib - 1 : m.ib);
    }
    for (int i = 0; i < 16; i++) idx2[i] = m.ib2 ? (int)br.get(i == 0 ? m.ib2 - 1 : m.ib2) : 0;
    for (int i = 0; i < 16; i++) {
        const int* e0 = ep[2 * subset[i]];
        const int* e1 = ep[2 * subset[i] + 1];
        int ci = idx1[i], ai = idx1[i], cb = m.ib, abb = m.ib;
        if (m.ib2) {                                           // modes 4 and 5: separate colour and alpha index sets
            if (isel) { ci = idx2[i]; cb = m.ib2; ai = idx1[i]; abb = m.ib; }
            else { ci = idx1[i]; cb = m.ib; ai = idx2[i]; abb = m.ib2; }
        }
        const int wc = weights(cb)[ci], wa = weights(abb)[ai];
        uint8_t px[4] = {lerp6(e0[0], e1[0], wc), lerp6(e0[1], e1[1], wc), lerp6(e0[2], e1[2], wc), m.ab ? lerp6(e0[3], e1[3], wa) : (uint8_t)255};
        if (rot) { const uint8_t t = px[3]; px[3] = px[rot - 1]; px[rot - 1] = t; }
        memcpy(out[i], px, 4);
    }
}

struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };
struct Reader {
    FILE* f; bool ok = true;
    uint32_t u32() { uint8_t b[4]; if (fread(b, 1, 4, f) != 4) { ok = false; return 0; } return b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24); }
    uint32_t u16() { uint8_t b[2]; if (fread(b, 1, 2, f) != 2) { ok = false; return 0; } return b[0] | (b[1] << 8); }
};

// Godot 4 Image::Format values this reader understands (core/io/image.h)
enum { GD_FORMAT_R8 = 2, GD_FORMAT_RGB8 = 4, GD_FORMAT_RGBA8 = 5, GD_FORMAT_BPTC_RGBA = 22 };
enum { GD_DATA_FORMAT_IMAGE = 0 };

size_t level_bytes(int fmt, int w, int h) {
    switch (fmt) {
        case GD_FORMAT_R8: return (size_t)w * h;
        case GD_FORMAT_RGB8: return (size_t)w * h * 3;
        case GD_FORMAT_RGBA8: return (size_t)w * h * 4;
        case GD_FORMAT_BPTC_RGBA: return (size_t)((w + 3) / 4) * ((h + 3) / 4) * 16;
        default: return 0;
    }
}

// one image record (CompressedTexture2D::load_image_from_file): data format, w, h, mip count, pixel format, then every level back to back.
// Appends level l (l = 0 .. mips) as RGBA8 to `levels`.
int read_image_record(Reader& r, const char* what, int& w, int& h, std::vector<std::vector<uint8_t>>& levels) {
    const uint32_t data_format = r.u32();
    w = (int)r.u16(); h = (int)r.u16();
    const uint32_t mips = r.u32();
    const int fmt = (int)r.u32();
    if (!r.ok) { snprintf(g_asset_err, sizeof g_asset_err, "%s: truncated image header", what); return CSKY_ERR_IO; }
    if (data_format != GD_DATA_FORMAT_IMAGE) { snprintf(g_asset_err, sizeof g_asset_err, "%s: data format %u (PNG/WebP/Basis) is not supported, only raw image data", what, data_format); return CSKY_ERR_IO; }
    if (w < 1 || h < 1 || w > 16384 || h > 16384 || mips > 16 || !level_bytes(fmt, 4, 4)) { snprintf(g_asset_err, sizeof g_asset_err, "%s: unsupported image (%dx%d, %u mips, format %d)", what, w, h, mips, fmt); return CSKY_ERR_IO; }
    int lw = w, lh = h;
    for (uint32_t l = 0; l <= mips; l++) {
        const size_t nb = level_bytes(fmt, lw, lh);
        std::vector<uint8_t> raw(nb);
        if (fread(raw.data(), 1, nb, r.f) != nb) { snprintf(g_asset_err, sizeof g_asset_err, "%s: truncated level %u", what, l); return CSKY_ERR_IO; }
        std::vector<uint8_t> px((size_t)lw * lh * 4);
        if (fmt == GD_FORMAT_BPTC_RGBA) {
            if (csky_decode_bc7(raw.data(), lw, lh, px.data()) != CSKY_OK) return CSKY_ERR_IO;
        } else {
            const int ch = fmt == GD_FORMAT_R8 ? 1 : (fmt == GD_FORMAT_RGB8 ? 3 : 4);
            for (size_t i = 0; i < (size_t)lw * lh; i++) {
                px[4 * i + 0] = raw[ch * i]; px[4 * i + 1] = ch > 1 ? raw[ch * i + 1] : 0; px[4 * i + 2] = ch > 2 ? raw[ch * i + 2] : 0; px[4 * i + 3] = ch > 3 ? raw[ch * i + 3] : 255;
            }
        }
        levels.push_back(std::move(px));
        lw = lw > 1 ? lw >> 1 : 1; lh = lh > 1 ? lh >> 1 : 1;
    }
    return CSKY_OK;
}

}  // namespace

extern "C" {

int csky_decode_bc7(const uint8_t* blocks, int w, int h, uint8_t* out_rgba8) {
    if (!blocks || !out_rgba8 || w < 1 || h < 1) { snprintf(g_asset_err, sizeof g_asset_err, "decode_bc7: bad arguments"); return CSKY_ERR_INVALID; }
    const int bw = (w + 3) / 4, bh = (h + 3) / 4;
    uint8_t px[16][4];
    for (int by = 0; by < bh; by++) for (int bx = 0; bx < bw; bx++) {
        decode_block(blocks + ((size_t)by * bw + bx) * 16, px);
        for (int i = 0; i < 16; i++) {
            const int x = bx * 4 + (i & 3), y = by * 4 + (i >> 2);
            if (x < w && y < h) memcpy(out_rgba8 + ((size_t)y * w + x) * 4, px[i], 4);
        }
    }
    return CSKY_OK;
}

static int load_ctex_impl(const char* path, int* w, int* h, int* levels, uint8_t* out_rgba8, size_t out_capacity) {
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex: cannot open %s", path ? path : "(null)"); return CSKY_ERR_IO; }
    FileCloser closer{f};                                      // every return path (and an exception) closes the file
    Reader r{f};
    uint8_t magic[4];
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "GST2", 4) != 0) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex: %s is not a CompressedTexture2D (GST2)", path); return CSKY_ERR_IO; }
    const uint32_t version = r.u32();
    r.u32(); r.u32();                                          // custom width / height
    r.u32();                                                   // flags
    r.u32();                                                   // mipmap limit
    r.u32(); r.u32(); r.u32();                                 // reserved
    if (!r.ok || version > 1) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex: %s: unsupported version %u", path, version); return CSKY_ERR_IO; }
    int iw = 0, ih = 0;
    std::vector<std::vector<uint8_t>> lv;
    const int rc = read_image_record(r, "load_ctex", iw, ih, lv);
    if (rc != CSKY_OK) return rc;
    if (w) *w = iw; if (h) *h = ih; if (levels) *levels = (int)lv.size();
    if (!out_rgba8) return CSKY_OK;                            // size query
    size_t total = 0; for (auto& l : lv) total += l.size();
    if (out_capacity < total) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex: output buffer too small (%zu needed)", total); return CSKY_ERR_INVALID; }
    size_t o = 0; for (auto& l : lv) { memcpy(out_rgba8 + o, l.data(), l.size()); o += l.size(); }
    return CSKY_OK;
}

int csky_load_ctex(const char* path, int* w, int* h, int* levels, uint8_t* out_rgba8, size_t out_capacity) {
    try { return load_ctex_impl(path, w, h, levels, out_rgba8, out_capacity); }
    catch (const std::exception&) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex: out of memory"); return CSKY_ERR_IO; }   // nothing throws across the ABI
}

static int load_ctex3d_impl(const char* path, int* w, int* h, int* d, int* levels, uint8_t* out_rgba8, size_t out_capacity) {
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex3d: cannot open %s", path ? path : "(null)"); return CSKY_ERR_IO; }
    FileCloser closer{f};                                      // every return path (and an exception) closes the file
    Reader r{f};
    uint8_t magic[4];
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "GSTL", 4) != 0) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex3d: %s is not a CompressedTexture3D (GSTL)", path); return CSKY_ERR_IO; }
    const uint32_t version = r.u32();
    const int depth = (int)r.u32();
    r.u32();                                                   // layer type
    r.u32();                                                   // data format flags
    const int mip_images = (int)r.u32();                       // number of mip SLICES that follow the level-0 slices
    r.u32(); r.u32();                                          // reserved
    if (!r.ok || version > 1 || depth < 1 || depth > 4096 || mip_images < 0 || mip_images > 8192) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex3d: %s: bad header", path); return CSKY_ERR_IO; }
    // slices of level 0, then the slices of level 1 (depth/2 of them), level 2, ...: every slice is one image record without 2-D mips
    std::vector<std::vector<uint8_t>> slices;
    int iw = 0, ih = 0, lw = 0, lh = 0, ld = depth, n_levels = 0, left_in_level = depth;
    for (int i = 0; i < depth + mip_images; i++) {
        int sw = 0, sh = 0;
        std::vector<std::vector<uint8_t>> one;
        const int rc = read_image_record(r, "load_ctex3d", sw, sh, one);
        if (rc != CSKY_OK) return rc;
        if (i == 0) { iw = lw = sw; ih = lh = sh; n_levels = 1; }
        if (left_in_level == 0) {                              // next mip level
            lw = lw > 1 ? lw >> 1 : 1; lh = lh > 1 ? lh >> 1 : 1; ld = ld > 1 ? ld >> 1 : 1; left_in_level = ld; n_levels++;
        }
        if (sw != lw || sh != lh) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex3d: slice %d is %dx%d, expected %dx%d", i, sw, sh, lw, lh); return CSKY_ERR_IO; }
        slices.push_back(std::move(one[0]));
        left_in_level--;
    }
    if (left_in_level != 0) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex3d: the last mip level is incomplete"); return CSKY_ERR_IO; }
    if (w) *w = iw; if (h) *h = ih; if (d) *d = depth; if (levels) *levels = n_levels;
    if (!out_rgba8) return CSKY_OK;
    size_t total = 0; for (auto& s : slices) total += s.size();
    if (out_capacity < total) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex3d: output buffer too small (%zu needed)", total); return CSKY_ERR_INVALID; }
    size_t o = 0; for (auto& s : slices) { memcpy(out_rgba8 + o, s.data(), s.size()); o += s.size(); }
    return CSKY_OK;
}

int csky_load_ctex3d(const char* path, int* w, int* h, int* d, int* levels, uint8_t* out_rgba8, size_t out_capacity) {
    try { return load_ctex3d_impl(path, w, h, d, levels, out_rgba8, out_capacity); }
    catch (const std::exception&) { snprintf(g_asset_err, sizeof g_asset_err, "load_ctex3d: out of memory"); return CSKY_ERR_IO; }
}

}  // extern "C"
