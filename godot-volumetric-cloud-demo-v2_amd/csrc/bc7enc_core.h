// bc7enc_core.h -- one BC7 (BPTC RGBA UNORM) block from 4 x 4 RGBA8 texels, host+device: the GPU encoder (bc7enc.hip::bc7_encode_kernel, one block
// per lane) and the host twin of tests/hostsim run the same code and produce the same 16 bytes.
//
// Why it exists (VERDICT r3 missing 2): the reference imports its three noise inputs with compress/mode=2, high_quality=true
// (weather.bmp.import:19-20, worlnoise.bmp.import:19-20, perlworlnoise.tga.import:19-20), i.e. its samplers return DECODED BC7 blocks, not the
// bytes of the .bmp / .tga.  The engine's encoder cannot be reproduced (it is the engine's), so no frame made here can be "what Godot samples"; what
// an encoder of this class gives is the SIZE of that difference: textures passed through encode -> csky_decode_bc7 (the decoder is pinned against an
// independent one, tests/test_godot_import.py) and marched, next to the same frame from the uncompressed bytes (tools/bc7_sensitivity.py).
//
// All eight modes: 6 (one subset, RGBA 7777 + p-bit per end point, 4-bit indices); for opaque blocks 1 and 3 (two subsets out of 64 partitions:
// RGB 666 + shared p-bit with 3-bit indices, RGB 777 + p-bits with 2-bit indices) and 0 and 2 (three subsets: RGB 444 + p-bits with 3-bit indices out
// of 16 partitions, RGB 555 with 2-bit indices out of 64); for blocks whose alpha varies 5 and 4 (RGB + a separately indexed scalar, four channel
// rotations: 777 / 8 with two 2-bit index sets; 555 / 6 with a 2-bit and a 3-bit set either way round) and 7 (two subsets, RGBA 5555 + p-bits,
// 2-bit indices).  Per candidate:
// principal axis of the subset's texels (covariance, power iteration), end points at the extreme projections, quantisation to the mode's
// precision over the p-bit choices, exhaustive index search against the palette THE DECODER builds (integer, bit for bit), up to three least-squares
// refits of the end points for those indices; the candidate with the smallest summed squared error over the four channels wins.
// Floating point is used for the fits only (+ - * / in IEEE single precision, contraction off: the same on both sides); everything that decides
// the emitted bits beyond that is integer.
#pragma once
#include "csky_common.h"

#pragma clang fp contract(off)

namespace csky {

CSKY_HD int bc7_weight(int bits, int i) {
    const unsigned char w2[4] = {0, 21, 43, 64}, w3[8] = {0, 9, 18, 27, 37, 46, 55, 64}, w4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};
    return bits == 2 ? w2[i] : (bits == 3 ? w3[i] : w4[i]);
}
// bit i = subset of texel i in two-subset partition p (bc7_tables.h::kBc7Partition2, derived by tools/derive_bc7_tables.py; tests check the two agree)
CSKY_HD unsigned bc7_part2_mask(int p) {
    const unsigned short t[64] = {0xCCCC, 0x8888, 0xEEEE, 0xECC8, 0xC880, 0xFEEC, 0xFEC8, 0xEC80, 0xC800, 0xFFEC, 0xFE80, 0xE800, 0xFFE8, 0xFF00, 0xFFF0, 0xF000,
                                  0xF710, 0x008E, 0x7100, 0x08CE, 0x008C, 0x7310, 0x3100, 0x8CCE, 0x088C, 0x3110, 0x6666, 0x366C, 0x17E8, 0x0FF0, 0x718E, 0x399C,
                                  0xAAAA, 0xF0F0, 0x5A5A, 0x33CC, 0x3C3C, 0x55AA, 0x9696, 0xA55A, 0x73CE, 0x13C8, 0x324C, 0x3BDC, 0x6996, 0xC33C, 0x9966, 0x0660,
                                  0x0272, 0x04E4, 0x4E40, 0x2720, 0xC936, 0x936C, 0x39C6, 0x639C, 0x9336, 0x9CC6, 0x817E, 0xE718, 0xCCF0, 0x0FCC, 0x7744, 0xEE22};
    return t[p];
}
CSKY_HD int bc7_anchor2(int p) {
    const unsigned char t[64] = {15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 2, 8, 2, 2, 8, 8, 15, 2, 8, 2, 2, 8, 8, 2, 2,
                                 15, 15, 6, 8, 2, 8, 15, 15, 2, 8, 2, 2, 2, 15, 15, 6, 6, 2, 6, 8, 15, 15, 2, 2, 15, 15, 15, 15, 15, 2, 2, 15};
    return t[p];
}
// two bits per texel: its subset in three-subset partition p (kBc7Partition3), and the anchors of subsets 1 and 2 (kBc7Anchor3a / 3b)
CSKY_HD unsigned bc7_part3_mask(int p) {
    const unsigned t[64] = {0xAA685050u, 0x6A5A5040u, 0x5A5A4200u, 0x5450A0A8u, 0xA5A50000u, 0xA0A05050u, 0x5555A0A0u, 0x5A5A5050u, 0xAA550000u, 0xAA555500u, 0xAAAA5500u, 0x90909090u,
                            0x94949494u, 0xA4A4A4A4u, 0xA9A59450u, 0x2A0A4250u, 0xA5945040u, 0x0A425054u, 0xA5A5A500u, 0x55A0A0A0u, 0xA8A85454u, 0x6A6A4040u, 0xA4A45000u, 0x1A1A0500u,
                            0x0050A4A4u, 0xAAA59090u, 0x14696914u, 0x69691400u, 0xA08585A0u, 0xAA821414u, 0x50A4A450u, 0x6A5A0200u, 0xA9A58000u, 0x5090A0A8u, 0xA8A09050u, 0x24242424u,
                            0x00AA5500u, 0x24924924u, 0x24499224u, 0x50A50A50u, 0x500AA550u, 0xAAAA4444u, 0x66660000u, 0xA5A0A5A0u, 0x50A050A0u, 0x69286928u, 0x44AAAA44u, 0x66666600u,
                            0xAA444444u, 0x54A854A8u, 0x95809580u, 0x96969600u, 0xA85454A8u, 0x80959580u, 0xAA141414u, 0x96960000u, 0xAAAA1414u, 0xA05050A0u, 0xA0A5A5A0u, 0x96000000u,
                            0x40804080u, 0xA9A8A9A8u, 0xAAAAAA44u, 0x2A4A5254u};
    return t[p];
}
CSKY_HD int bc7_anchor3(int p, int sub) {
    const unsigned char a[64] = {3, 3, 15, 15, 8, 3, 15, 15, 8, 8, 6, 6, 6, 5, 3, 3, 3, 3, 8, 15, 3, 3, 6, 10, 5, 8, 8, 6, 8, 5, 15, 15,
                                 8, 15, 3, 5, 6, 10, 8, 15, 15, 3, 15, 5, 15, 15, 15, 15, 3, 15, 5, 5, 5, 8, 5, 10, 5, 10, 8, 13, 15, 12, 3, 3};
    const unsigned char b[64] = {15, 8, 8, 3, 15, 15, 3, 8, 15, 15, 15, 15, 15, 15, 15, 8, 15, 8, 15, 3, 15, 8, 15, 8, 3, 15, 6, 10, 15, 15, 10, 8,
                                 15, 3, 15, 10, 10, 8, 9, 10, 6, 15, 8, 15, 3, 6, 6, 8, 15, 3, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 3, 15, 15, 8};
    return sub == 0 ? 0 : (sub == 1 ? a[p] : b[p]);
}
// subset of texel i / anchor texel of subset `sub` in partition p of an ns-subset mode
CSKY_HD int bc7_subset_of(int ns, int p, int i) { return ns == 2 ? (int)((bc7_part2_mask(p) >> i) & 1u) : (int)((bc7_part3_mask(p) >> (2 * i)) & 3u); }
CSKY_HD int bc7_anchor_of(int ns, int p, int sub) { return ns == 2 ? (sub ? bc7_anchor2(p) : 0) : bc7_anchor3(p, sub); }
CSKY_HD int bc7_lerp6(int a, int b, int w) { return ((64 - w) * a + w * b + 32) >> 6; }

struct Bc7Bits {
    uint32_t w[4];
    int pos;
    CSKY_HD void put(uint32_t v, int n) {                        // LSB first
        for (int i = 0; i < n; i++, pos++) w[pos >> 5] |= ((v >> i) & 1u) << (pos & 31);
    }
};

// One subset's fit in D channels: end points (real-valued, 0..255) along the principal axis of the texels sel[0..n)
template <int D> CSKY_HD void bc7_axis_fit(const float v[16][4], const int* sel, int n, float e0[4], float e1[4]) {
    float mean[D];
    for (int c = 0; c < D; c++) { float s = 0.0f; for (int i = 0; i < n; i++) s += v[sel[i]][c]; mean[c] = s / (float)n; }
    float cov[D][D];
    for (int a = 0; a < D; a++) for (int b = a; b < D; b++) {
        float s = 0.0f;
        for (int i = 0; i < n; i++) s += (v[sel[i]][a] - mean[a]) * (v[sel[i]][b] - mean[b]);
        cov[a][b] = s; cov[b][a] = s;
    }
    int k = 0;
    for (int c = 1; c < D; c++) if (cov[c][c] > cov[k][k]) k = c;
    float ax[D];
    for (int c = 0; c < D; c++) ax[c] = cov[c][k];             // one power step from the unit vector of the widest channel
    if (!(cov[k][k] > 0.0f)) { for (int c = 0; c < D; c++) { e0[c] = mean[c]; e1[c] = mean[c]; } return; }   // all texels equal
    for (int it = 0; it < 6; it++) {
        float nx[D], big = 0.0f;
        for (int a = 0; a < D; a++) { float s = 0.0f; for (int b = 0; b < D; b++) s += cov[a][b] * ax[b]; nx[a] = s; const float m = s < 0.0f ? -s : s; if (m > big) big = m; }
        if (!(big > 0.0f)) break;
        for (int a = 0; a < D; a++) ax[a] = nx[a] / big;
    }
    float len2 = 0.0f;
    for (int c = 0; c < D; c++) len2 += ax[c] * ax[c];
    float lo = 0.0f, hi = 0.0f;
    for (int i = 0; i < n; i++) {
        float t = 0.0f;
        for (int c = 0; c < D; c++) t += (v[sel[i]][c] - mean[c]) * ax[c];
        t = t / len2;
        if (i == 0 || t < lo) lo = t;
        if (i == 0 || t > hi) hi = t;
    }
    for (int c = 0; c < D; c++) {
        e0[c] = fminf(fmaxf(mean[c] + lo * ax[c], 0.0f), 255.0f);
        e1[c] = fminf(fmaxf(mean[c] + hi * ax[c], 0.0f), 255.0f);
    }
}
// least-squares end points of one channel for fixed weights w[i] / 64 (false: the system is singular, all texels on one index)
CSKY_HD bool bc7_lsq(const float v[16][4], const int* sel, const int* wt, int n, int c, float& a, float& b) {
    float saa = 0.0f, sab = 0.0f, sbb = 0.0f, pa = 0.0f, pb = 0.0f;
    for (int i = 0; i < n; i++) {
        const float w1 = (float)wt[i] * (1.0f / 64.0f), w0 = 1.0f - w1, x = v[sel[i]][c];
        saa += w0 * w0; sab += w0 * w1; sbb += w1 * w1; pa += w0 * x; pb += w1 * x;
    }
    const float det = saa * sbb - sab * sab;
    if (!(det > 1e-4f)) return false;
    a = fminf(fmaxf((pa * sbb - pb * sab) / det, 0.0f), 255.0f);
    b = fminf(fmaxf((pb * saa - pa * sab) / det, 0.0f), 255.0f);
    return true;
}
CSKY_HD int bc7_round(float x) { return (int)(x + 0.5f); }       // x >= 0
// the stored value (`bits` wide, p-bit `p` appended when pbit) whose 8-bit expansion is nearest to x; out8 = that expansion
CSKY_HD int bc7_quant(float x, int bits, bool pbit, int p, int& out8) {
    const int total = bits + (pbit ? 1 : 0), top = (1 << bits) - 1;
    int best = 0, best_err = 1 << 30;
    const int guess = pbit ? ((bc7_round(x * (float)((1 << total) - 1) / 255.0f) - p) >> 1) : bc7_round(x * (float)top / 255.0f);
    for (int d = -1; d <= 1; d++) {
        int q = guess + d; q = q < 0 ? 0 : (q > top ? top : q);
        const int raw = pbit ? ((q << 1) | p) : q;
        const int v8 = ((raw << (8 - total)) | (raw >> (2 * total - 8))) & 255;
        const int xi = bc7_round(x), e = v8 > xi ? v8 - xi : xi - v8;
        if (e < best_err) { best_err = e; best = q; out8 = v8; }
    }
    return best;
}
// best index per texel against the decoder's palette of the 8-bit end points a8 / b8 over channels [c0, c1); returns the summed squared error
CSKY_HD unsigned bc7_indices(const unsigned char px[16][4], const int* sel, int n, const int a8[4], const int b8[4], int c0, int c1, int ibits, int* idx) {
    unsigned total = 0;
    const int levels = 1 << ibits;
    for (int i = 0; i < n; i++) {
        unsigned be = 0xffffffffu; int bi = 0;
        for (int l = 0; l < levels; l++) {
            const int w = bc7_weight(ibits, l);
            unsigned e = 0;
            for (int c = c0; c < c1; c++) { const int d = bc7_lerp6(a8[c], b8[c], w) - (int)px[sel[i]][c]; e += (unsigned)(d * d); }
            if (e < be) { be = e; bi = l; }
        }
        idx[i] = bi; total += be;
    }
    return total;
}

constexpr int BC7_PARTITIONS_TRIED = 4;                          // mode 1: partitions fitted in full, after a plain estimate of all 64
constexpr int BC7_PARTITIONS_MAX = 8;                            // ... at quality 1
constexpr int BC7_REFITS = 3;                                   // least-squares refits of the end points per candidate
// A fitted subset: stored end points q0 / q1 (+ their 8-bit expansions), p-bits, indices, error.  Channels [c0, c1) of `px`.
struct Bc7Subset { int q0[4] = {0, 0, 0, 0}, q1[4] = {0, 0, 0, 0}, a8[4] = {0, 0, 0, 0}, b8[4] = {0, 0, 0, 0}, p0 = 0, p1 = 0, idx[16] = {}; unsigned err = 0xffffffffu; };

// fit, quantise over the p-bit choices, index, refit once.  pmode: 0 none, 1 one p-bit per end point, 2 one shared by both end points
template <int D> CSKY_HD void bc7_fit_subset(const unsigned char px[16][4], const float v[16][4], const int* sel, int n, int c0, int bits, int pmode, int ibits, Bc7Subset& out, bool refine = false) {
    float e0[4] = {0, 0, 0, 0}, e1[4] = {0, 0, 0, 0};
    float vv[16][4];
    for (int i = 0; i < 16; i++) for (int c = 0; c < D; c++) vv[i][c] = v[i][c0 + c];
    bc7_axis_fit<D>(vv, sel, n, e0, e1);
    out.err = 0xffffffffu;
    for (int pass = 0; pass < BC7_REFITS + 1; pass++) {
        const int np = pmode == 0 ? 1 : (pmode == 1 ? 4 : 2);
        Bc7Subset best; best.err = 0xffffffffu;
        for (int pc = 0; pc < np; pc++) {
            Bc7Subset s;
            s.p0 = pmode == 1 ? (pc & 1) : (pmode == 2 ? pc : 0);
            s.p1 = pmode == 1 ? (pc >> 1) : (pmode == 2 ? pc : 0);
            for (int c = 0; c < 4; c++) { s.q0[c] = s.q1[c] = 0; s.a8[c] = s.b8[c] = 0; }
            for (int c = 0; c < D; c++) {
                s.q0[c0 + c] = bc7_quant(e0[c], bits, pmode != 0, s.p0, s.a8[c0 + c]);
                s.q1[c0 + c] = bc7_quant(e1[c], bits, pmode != 0, s.p1, s.b8[c0 + c]);
            }
            s.err = bc7_indices(px, sel, n, s.a8, s.b8, c0, c0 + D, ibits, s.idx);
            if (s.err < best.err) best = s;
        }
        if (best.err < out.err) out = best;
        if (pass == BC7_REFITS || out.err == 0 || best.err > out.err) break;        // (a refit that made it worse ends the search)
        int wt[16];
        for (int i = 0; i < n; i++) wt[i] = bc7_weight(ibits, out.idx[i]);
        bool ok = true;
        for (int c = 0; c < D && ok; c++) ok = bc7_lsq(vv, sel, wt, n, c, e0[c], e1[c]);
        if (!ok) break;
    }
    // quality 1 (round 5: the second row of the sensitivity study, tools/bc7_sensitivity.py): coordinate descent on the STORED end points -- every
    // channel of either end moved by -2 .. +2 steps with the p-bits kept, indices searched again against the decoder's palette, a move kept when the
    // subset's squared error drops; until a whole sweep improves nothing (at most four sweeps).  Integer throughout.
    if (refine && out.err != 0u && out.err != 0xffffffffu) {
        const int total = bits + (pmode != 0 ? 1 : 0), top = (1 << bits) - 1;
        for (int sweep = 0; sweep < 4; sweep++) {
            bool improved = false;
            for (int c = c0; c < c0 + D; c++) for (int end = 0; end < 2; end++) for (int d = -2; d <= 2; d++) {
                if (d == 0) continue;
                Bc7Subset t = out;
                const int q = (end ? t.q1[c] : t.q0[c]) + d;
                if (q < 0 || q > top) continue;
                const int raw = pmode != 0 ? ((q << 1) | (end ? t.p1 : t.p0)) : q;
                const int v8 = ((raw << (8 - total)) | (raw >> (2 * total - 8))) & 255;
                if (end) { t.q1[c] = q; t.b8[c] = v8; } else { t.q0[c] = q; t.a8[c] = v8; }
                t.err = bc7_indices(px, sel, n, t.a8, t.b8, c0, c0 + D, ibits, t.idx);
                if (t.err < out.err) { out = t; improved = true; }
            }
            if (!improved || out.err == 0u) break;
        }
    }
}
// the anchor texel's index must have a clear top bit: swap the end points and mirror the subset's indices
CSKY_HD void bc7_fix_anchor(Bc7Subset& s, int n, int anchor_pos, int ibits, int c0, int c1) {
    if (!(s.idx[anchor_pos] >> (ibits - 1))) return;
    for (int c = c0; c < c1; c++) { int t = s.q0[c]; s.q0[c] = s.q1[c]; s.q1[c] = t; t = s.a8[c]; s.a8[c] = s.b8[c]; s.b8[c] = t; }
    const int t = s.p0; s.p0 = s.p1; s.p1 = t;
    for (int i = 0; i < n; i++) s.idx[i] = ((1 << ibits) - 1) - s.idx[i];
}

CSKY_HD void bc7_encode_block(const unsigned char px[16][4], uint32_t out[4], int quality = 0) {
    const bool refine = quality >= 1;
    const int ntry = quality >= 1 ? BC7_PARTITIONS_MAX : BC7_PARTITIONS_TRIED;   // partitions fitted in full per two- / three-subset mode
    float v[16][4];
    bool opaque = true;
    for (int i = 0; i < 16; i++) { for (int c = 0; c < 4; c++) v[i][c] = (float)px[i][c]; opaque = opaque && px[i][3] == 255; }
    int all[16];
    for (int i = 0; i < 16; i++) all[i] = i;
    unsigned best_err = 0xffffffffu;
    Bc7Bits best; best.pos = 0; for (int k = 0; k < 4; k++) best.w[k] = 0;

    {   // ---- mode 6
        Bc7Subset s;
        bc7_fit_subset<4>(px, v, all, 16, 0, 7, 1, 4, s, refine);
        bc7_fix_anchor(s, 16, 0, 4, 0, 4);
        Bc7Bits b; b.pos = 0; for (int k = 0; k < 4; k++) b.w[k] = 0;
        b.put(1u << 6, 7);
        for (int c = 0; c < 4; c++) { b.put((uint32_t)s.q0[c], 7); b.put((uint32_t)s.q1[c], 7); }
        b.put((uint32_t)s.p0, 1); b.put((uint32_t)s.p1, 1);
        for (int i = 0; i < 16; i++) b.put((uint32_t)s.idx[i], i == 0 ? 3 : 4);
        best_err = s.err; best = b;
    }
    if (best_err != 0 && !opaque) {   // ---- mode 5: three channels as a vector, the fourth on its own; rotation r puts channel r - 1 into the scalar slot
        for (int rot = 0; rot < 4; rot++) {
            unsigned char rp[16][4]; float rv[16][4];
            for (int i = 0; i < 16; i++) {
                for (int c = 0; c < 4; c++) rp[i][c] = px[i][c];
                if (rot) { const unsigned char t = rp[i][3]; rp[i][3] = rp[i][rot - 1]; rp[i][rot - 1] = t; }
                for (int c = 0; c < 4; c++) rv[i][c] = (float)rp[i][c];
            }
            Bc7Subset col, sc;
            bc7_fit_subset<3>(rp, rv, all, 16, 0, 7, 0, 2, col, refine);
            bc7_fit_subset<1>(rp, rv, all, 16, 3, 8, 0, 2, sc, refine);
            const unsigned err = col.err + sc.err;
            if (err >= best_err) continue;
            bc7_fix_anchor(col, 16, 0, 2, 0, 3);
            bc7_fix_anchor(sc, 16, 0, 2, 3, 4);
            Bc7Bits b; b.pos = 0; for (int k = 0; k < 4; k++) b.w[k] = 0;
            b.put(1u << 5, 6); b.put((uint32_t)rot, 2);
            for (int c = 0; c < 3; c++) { b.put((uint32_t)col.q0[c], 7); b.put((uint32_t)col.q1[c], 7); }
            b.put((uint32_t)sc.q0[3], 8); b.put((uint32_t)sc.q1[3], 8);
            for (int i = 0; i < 16; i++) b.put((uint32_t)col.idx[i], i == 0 ? 1 : 2);
            for (int i = 0; i < 16; i++) b.put((uint32_t)sc.idx[i], i == 0 ? 1 : 2);
            best_err = err; best = b;
        }
    }
    if (best_err != 0 && !opaque) {   // ---- mode 4: like 5 with RGB 555 / A 6 and a 2-bit + a 3-bit index set, either of which may serve the colour (index selector)
        for (int rot = 0; rot < 4; rot++) {
            unsigned char rp[16][4]; float rv[16][4];
            for (int i = 0; i < 16; i++) {
                for (int c = 0; c < 4; c++) rp[i][c] = px[i][c];
                if (rot) { const unsigned char t = rp[i][3]; rp[i][3] = rp[i][rot - 1]; rp[i][rot - 1] = t; }
                for (int c = 0; c < 4; c++) rv[i][c] = (float)rp[i][c];
            }
            for (int isel = 0; isel < 2; isel++) {
                const int cb = isel ? 3 : 2, ab = isel ? 2 : 3;
                Bc7Subset col, sc;
                bc7_fit_subset<3>(rp, rv, all, 16, 0, 5, 0, cb, col, refine);
                bc7_fit_subset<1>(rp, rv, all, 16, 3, 6, 0, ab, sc, refine);
                const unsigned err = col.err + sc.err;
                if (err >= best_err) continue;
                bc7_fix_anchor(col, 16, 0, cb, 0, 3);
                bc7_fix_anchor(sc, 16, 0, ab, 3, 4);
                Bc7Bits b; b.pos = 0; for (int k = 0; k < 4; k++) b.w[k] = 0;
                b.put(1u << 4, 5); b.put((uint32_t)rot, 2); b.put((uint32_t)isel, 1);
                for (int c = 0; c < 3; c++) { b.put((uint32_t)col.q0[c], 5); b.put((uint32_t)col.q1[c], 5); }
                b.put((uint32_t)sc.q0[3], 6); b.put((uint32_t)sc.q1[3], 6);
                const Bc7Subset& two = isel ? sc : col;            // the 2-bit set is stored first, then the 3-bit one
                const Bc7Subset& three = isel ? col : sc;
                for (int i = 0; i < 16; i++) b.put((uint32_t)two.idx[i], i == 0 ? 1 : 2);
                for (int i = 0; i < 16; i++) b.put((uint32_t)three.idx[i], i == 0 ? 2 : 3);
                best_err = err; best = b;
            }
        }
    }
    // ---- two and three subsets: every partition estimated with a plain fit, the best few fitted in full in the modes the block class allows:
    //      opaque: 1 (two subsets, RGB 666 + shared p-bit, 3-bit indices), 3 (two, RGB 777 + p-bit per end point, 2-bit), 0 (three subsets out of the
    //      first 16 partitions, RGB 444 + p-bits, 3-bit), 2 (three, RGB 555, 2-bit); blocks whose alpha varies: 7 (two subsets, RGBA 5555 + p-bits, 2-bit)
    for (int ns = 2; ns <= (opaque ? 3 : 2) && best_err != 0; ns++) {
        const int D = opaque ? 3 : 4;
        int cand[BC7_PARTITIONS_MAX], cand16[BC7_PARTITIONS_MAX]; unsigned cerr[BC7_PARTITIONS_MAX], cerr16[BC7_PARTITIONS_MAX];
        for (int k = 0; k < BC7_PARTITIONS_MAX; k++) { cand[k] = cand16[k] = 0; cerr[k] = cerr16[k] = 0xffffffffu; }
        for (int p = 0; p < 64; p++) {
            unsigned est = 0;
            for (int sub = 0; sub < ns; sub++) {
                int sel[16], n = 0;
                for (int i = 0; i < 16; i++) if (bc7_subset_of(ns, p, i) == sub) sel[n++] = i;
                int lo[4] = {255, 255, 255, 255}, hi[4] = {0, 0, 0, 0};
                for (int i = 0; i < n; i++) for (int c = 0; c < D; c++) { const int x = px[sel[i]][c]; lo[c] = x < lo[c] ? x : lo[c]; hi[c] = x > hi[c] ? x : hi[c]; }
                // squared distance of every texel to the box diagonal: what a line fit cannot remove
                float d[4] = {0, 0, 0, 0}, dd = 0.0f;
                for (int c = 0; c < D; c++) { d[c] = (float)(hi[c] - lo[c]); dd += d[c] * d[c]; }
                for (int i = 0; i < n; i++) {
                    float t = 0.0f, r2 = 0.0f, r[4] = {0, 0, 0, 0};
                    for (int c = 0; c < D; c++) { r[c] = (float)(px[sel[i]][c] - lo[c]); t += r[c] * d[c]; }
                    t = dd > 0.0f ? t / dd : 0.0f;
                    for (int c = 0; c < D; c++) { const float q = r[c] - t * d[c]; r2 += q * q; }
                    est += (unsigned)(r2 + 0.5f);
                }
            }
            for (int k = 0; k < ntry; k++) if (est < cerr[k]) {         // insertion into the short list
                for (int j = ntry - 1; j > k; j--) { cerr[j] = cerr[j - 1]; cand[j] = cand[j - 1]; }
                cerr[k] = est; cand[k] = p; break;
            }
            if (p < 16) for (int k = 0; k < ntry; k++) if (est < cerr16[k]) {   // mode 0 stores four partition bits
                for (int j = ntry - 1; j > k; j--) { cerr16[j] = cerr16[j - 1]; cand16[j] = cand16[j - 1]; }
                cerr16[k] = est; cand16[k] = p; break;
            }
        }
        const int n_modes = opaque ? 2 : 1;
        for (int mi = 0; mi < n_modes; mi++) {
            const int mode = !opaque ? 7 : (ns == 2 ? (mi == 0 ? 1 : 3) : (mi == 0 ? 0 : 2));
            const int bits = mode == 1 ? 6 : (mode == 3 ? 7 : (mode == 0 ? 4 : 5)), pmode = mode == 1 ? 2 : (mode == 2 ? 0 : 1), ibits = (mode == 1 || mode == 0) ? 3 : 2;
            for (int k = 0; k < ntry; k++) {
                const int p = mode == 0 ? cand16[k] : cand[k];
                Bc7Subset s[3]; int sel[3][16], n[3] = {0, 0, 0}, apos[3] = {0, 0, 0};
                for (int i = 0; i < 16; i++) { const int sub = bc7_subset_of(ns, p, i); if (i == bc7_anchor_of(ns, p, sub)) apos[sub] = n[sub]; sel[sub][n[sub]++] = i; }
                unsigned err = 0;
                for (int sub = 0; sub < ns; sub++) {
                    if (opaque) bc7_fit_subset<3>(px, v, sel[sub], n[sub], 0, bits, pmode, ibits, s[sub], refine);
                    else bc7_fit_subset<4>(px, v, sel[sub], n[sub], 0, bits, pmode, ibits, s[sub], refine);
                    err += s[sub].err;
                }
                if (err >= best_err) continue;
                for (int sub = 0; sub < ns; sub++) bc7_fix_anchor(s[sub], n[sub], apos[sub], ibits, 0, D);
                Bc7Bits b; b.pos = 0; for (int q = 0; q < 4; q++) b.w[q] = 0;
                b.put(1u << mode, mode + 1); b.put((uint32_t)p, mode == 0 ? 4 : 6);
                for (int c = 0; c < D; c++) for (int sub = 0; sub < ns; sub++) { b.put((uint32_t)s[sub].q0[c], bits); b.put((uint32_t)s[sub].q1[c], bits); }
                if (pmode == 2) { for (int sub = 0; sub < ns; sub++) b.put((uint32_t)s[sub].p0, 1); }
                else if (pmode == 1) for (int sub = 0; sub < ns; sub++) { b.put((uint32_t)s[sub].p0, 1); b.put((uint32_t)s[sub].p1, 1); }
                int pos[3] = {0, 0, 0};
                for (int i = 0; i < 16; i++) {
                    const int sub = bc7_subset_of(ns, p, i);
                    b.put((uint32_t)s[sub].idx[pos[sub]++], i == bc7_anchor_of(ns, p, sub) ? ibits - 1 : ibits);
                }
                best_err = err; best = b;
            }
        }
    }
    for (int k = 0; k < 4; k++) out[k] = best.w[k];
}

// the 4 x 4 texels of block (bx, by) of a w x h RGBA8 image; texels beyond the edge repeat the last row / column (what an encoder pads with)
CSKY_HD void bc7_gather_block(const unsigned char* __restrict__ img, int w, int h, int bx, int by, unsigned char px[16][4]) {
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) {
        const int x = bx * 4 + i < w ? bx * 4 + i : w - 1, y = by * 4 + j < h ? by * 4 + j : h - 1;
        for (int c = 0; c < 4; c++) px[j * 4 + i][c] = img[((size_t)y * w + x) * 4 + c];
    }
}

}  // namespace csky

#pragma clang fp contract(fast)
