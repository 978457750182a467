// bake.h -- host-side conversion of the 8-bit mip chains into the device texture layouts of csky_common.h.
// Internal to libcloudsky (api.cpp); also included by the host-compiled kernel-core unit test (tests/hostsim).
#pragma once
#include <cmath>
#include <vector>
#include "csky_common.h"
#include "bake_core.h"
#include "../../include/cloudsky.h"

namespace csky {
// ---- texture baking on the HOST (layouts documented in csky_common.h; per-texel code shared with the GPU bake: bake_core.h).
// libcloudsky bakes on the device (api.cpp::csky_set_noise); these loops serve tests/hostsim and the device-vs-host byte comparison.
inline void bake_shape(const std::vector<uint8_t>& chain, std::vector<ShapeTexel>& out, uint32_t off[SHAPE_LEVELS], unsigned long long* inexact = nullptr) {
    size_t total = 0;
    for (int l = 0; l < SHAPE_LEVELS; l++) { off[l] = (uint32_t)total; size_t n = SHAPE_N >> l; total += n * n * n; }
    out.resize(total);
    unsigned bad = 0;
    for (int l = 0; l < SHAPE_LEVELS; l++) {
        const int n = SHAPE_N >> l;
        const uint8_t* src = chain.data() + chain_offset(SHAPE_N, l, 4);
        for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) out[off[l] + shape_cell_index(n, x, y, z)] = bake_shape_texel(src, n, x, y, z, bad);
    }
    if (inexact) *inexact += bad;
}
inline void bake_detail(const std::vector<uint8_t>& chain, std::vector<uint4>& out, uint32_t off[DETAIL_LEVELS], unsigned long long* inexact = nullptr) {
    size_t total = 0;
    for (int l = 0; l < DETAIL_LEVELS; l++) { off[l] = (uint32_t)total; size_t n = DETAIL_N >> l; total += n * n * n; }
    out.resize(total);
    unsigned bad = 0;
    for (int l = 0; l < DETAIL_LEVELS; l++) {
        const int n = DETAIL_N >> l;
        const uint8_t* src = chain.data() + chain_offset(DETAIL_N, l, 3);
        for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) out[off[l] + ((size_t)z * n + y) * n + x] = bake_detail_texel(src, n, x, y, z, bad);
    }
    if (inexact) *inexact += bad;
}
// unpacked fp16 numerators of the whole detail chain (the source of the LDS copy of the "lds" kernel variant)
inline void bake_detail_unpacked(const std::vector<uint8_t>& chain, std::vector<uint16_t>& out) {
    out.clear();
    for (int l = 0; l < DETAIL_LEVELS; l++) {
        const int n = DETAIL_N >> l;
        const uint8_t* src = chain.data() + chain_offset(DETAIL_N, l, 3);
        for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) out.push_back(f2h((float)detail_numerator(src, n, x, y, z)));
    }
}
inline void bake_weather(const uint8_t* rgb, std::vector<uint4>& out, unsigned long long* inexact = nullptr) {
    const int n = WEATHER_N;
    out.resize((size_t)n * n);
    unsigned bad = 0;
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) out[(size_t)y * n + x] = bake_weather_texel(rgb, x, y, bad);
    if (inexact) *inexact += bad;
}

// Height-fraction window outside which density() (clouds.glsl:109-137) is provably 0 for the WHOLE bound weather map.
// density() needs g > 1 - coverage*weather.b (clouds.glsl:121-125) with g = smoothstep(gx,gy,hf) - smoothstep(gz,gw,hf)
// (:92-95).  Both smoothsteps lie in [0,1] and are non-decreasing in hf, so for any cloud type ct
//      g <= S1(hf,ct) = smoothstep(gx,gy,hf)            and            g <= 1 - S2(hf,ct) = 1 - smoothstep(gz,gw,hf).
// With tau = 1 - coverage*max(weather.b) the smallest threshold anywhere on the map: every sample with
// max_ct S1(hf,ct) <= tau (below the cloud body) or max_ct (1 - S2(hf,ct)) <= tau (above it) fails reject (1) of density().
// The maxima run over the map's cloud-type range [rmin,rmax] (bilinear filtering stays inside the texel range) on a
// 4096-point grid; 0.03 of slack in g covers the grid spacing and fp32 rounding by two orders of magnitude, and the
// window is widened by 5e-4 in hf (the fp32 height-fraction quantum is 2e-4).  lo = -1 / hi = 2 disable the reject.
inline void height_window(double coverage, double rmin, double rmax, double bmax, float& lo, float& hi) {
    lo = -1.0f; hi = 2.0f;
    if (!(coverage <= 1.0) || !(bmax <= 1.0) || !(rmin >= 0.0) || !(rmax <= 1.0)) return;   // wc could exceed 1: keep the plain path
    const double tm = (1.0 - coverage * bmax) - 0.03;
    if (!(tm > 0.0)) return;
    auto sat = [](double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); };
    auto smooth = [&](double e0, double e1, double x) { const double t = sat((x - e0) / (e1 - e0)); return t * t * (3.0 - 2.0 * t); };
    auto bound = [&](double h, bool upper) {              // max over ct of S1 (lower side) or 1 - S2 (upper side)
        double best = 0.0;
        const int N = 4096;
        for (int i = 0; i <= N; i++) {
            const double ct = rmin + (rmax - rmin) * (double)i / N;
            const double st = 1.0 - sat(ct * 2.0), sc = 1.0 - std::fabs(ct - 0.5) * 2.0, cu = sat(ct - 0.5) * 2.0;   // clouds.glsl:86-88
            const double gx = 0.02 * st + 0.02 * sc + 0.01 * cu, gy = 0.05 * st + 0.2 * sc + 0.0625 * cu;
            const double gz = 0.09 * st + 0.48 * sc + 0.78 * cu, gw = 0.11 * st + 0.625 * sc + 1.0 * cu;
            const double v = upper ? 1.0 - smooth(gz, gw, h) : smooth(gx, gy, h);
            if (v > best) best = v;
        }
        return best;
    };
    if (bound(1.0, false) <= tm) { lo = 2.0f; hi = 2.0f; return; }        // nothing can ever pass: reject everything
    double a = 0.0, b = 1.0;                                             // largest h with S1max(h) <= tm (S1max non-decreasing)
    if (bound(0.0, false) <= tm) { for (int it = 0; it < 40; it++) { const double m = 0.5 * (a + b); if (bound(m, false) <= tm) a = m; else b = m; } lo = (float)(a - 5e-4); }
    a = 0.0; b = 1.0;                                                    // smallest h with (1 - S2)max(h) <= tm (non-increasing)
    if (bound(1.0, true) <= tm) { for (int it = 0; it < 40; it++) { const double m = 0.5 * (a + b); if (bound(m, true) <= tm) b = m; else a = m; } hi = (float)(b + 5e-4); }
}

}  // namespace csky
