// bake.h -- host-side conversion of the 8-bit mip chains into the device texture layouts of csky_common.h.
// Internal to libcloudsky (api.cpp); also included by the host-compiled kernel-core unit test (tests/hostsim).
#pragma once
#include <cmath>
#include <vector>
#include "csky_common.h"
#include "../../include/cloudsky.h"

namespace csky {
// ---- texture baking (layouts documented in csky_common.h) ----
// fp16 pair {lo, hi} of two integer coefficients; *inexact counts values fp16 cannot hold exactly (|v| > 2048 and odd, ...)
inline uint32_t hpair_i(int lo, int hi, unsigned long long* inexact) {
    const uint16_t a = f2h((float)lo), b = f2h((float)hi);
    if (inexact) { if (h2f(a) != (float)lo) ++*inexact; if (h2f(b) != (float)hi) ++*inexact; }
    return (uint32_t)a | ((uint32_t)b << 16);
}
// polynomial-cell coefficients (csky_common.h) of the cell whose corners are v[x | y<<1 | z<<2]
inline void cell_coeffs(const int v[8], int c[8]) {
    c[0] = v[0]; c[1] = v[1] - v[0]; c[2] = v[2] - v[0]; c[3] = v[3] - v[2] - v[1] + v[0];
    c[4] = v[4] - v[0]; c[5] = v[5] - v[4] - v[1] + v[0]; c[6] = v[6] - v[4] - v[2] + v[0];
    c[7] = v[7] - v[6] - v[5] + v[4] - v[3] + v[2] + v[1] - v[0];
}

inline void bake_shape(const std::vector<uint8_t>& chain, std::vector<ShapeTexel>& out, uint32_t off[SHAPE_LEVELS], unsigned long long* inexact = nullptr) {
    size_t total = 0;
    for (int l = 0; l < SHAPE_LEVELS; l++) { off[l] = (uint32_t)total; size_t n = SHAPE_N >> l; total += n * n * n; }
    out.resize(total);
    for (int l = 0; l < SHAPE_LEVELS; l++) {
        const int n = SHAPE_N >> l;
        const uint8_t* src = chain.data() + csky_mip_offset(SHAPE_N, l, 4);
        auto tx = [&](int x, int y, int z) { return src + (((size_t)(z % n) * n + (y % n)) * n + (x % n)) * 4; };
        for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
            int vr[8], vf[8], cr[8], cf[8];
            for (int k = 0; k < 8; k++) {
                const uint8_t* t = tx(x + (k & 1), y + ((k >> 1) & 1), z + (k >> 2));
                vr[k] = t[0]; vf[k] = 5 * t[1] + 2 * t[2] + t[3];                                  // fbm numerator (clouds.glsl:118 x 8)
            }
            cell_coeffs(vr, cr); cell_coeffs(vf, cf);
            ShapeTexel& o = out[off[l] + ((size_t)z * n + y) * n + x];
#if CSKY_SHAPE_POLY == 1
            o = uint2{hpair_i(cr[0], cr[1], inexact), hpair_i(cf[0], cf[1], inexact)};
#elif CSKY_SHAPE_POLY == 2
            o = uint4{hpair_i(cr[0], cr[1], inexact), hpair_i(cr[2], cr[3], inexact), hpair_i(cf[0], cf[1], inexact), hpair_i(cf[2], cf[3], inexact)};
#else
            o.r = uint4{hpair_i(cr[0], cr[1], inexact), hpair_i(cr[2], cr[3], inexact), hpair_i(cr[4], cr[5], inexact), hpair_i(cr[6], cr[7], inexact)};
            o.f = uint4{hpair_i(cf[0], cf[1], inexact), hpair_i(cf[2], cf[3], inexact), hpair_i(cf[4], cf[5], inexact), hpair_i(cf[6], cf[7], inexact)};
#endif
        }
    }
}
inline void bake_detail(const std::vector<uint8_t>& chain, std::vector<uint4>& out, uint32_t off[DETAIL_LEVELS], unsigned long long* inexact = nullptr) {
    size_t total = 0;
    for (int l = 0; l < DETAIL_LEVELS; l++) { off[l] = (uint32_t)total; size_t n = DETAIL_N >> l; total += n * n * n; }
    out.resize(total);
    for (int l = 0; l < DETAIL_LEVELS; l++) {
        const int n = DETAIL_N >> l;
        const uint8_t* src = chain.data() + csky_mip_offset(DETAIL_N, l, 3);
        auto num = [&](int x, int y, int z) -> int {
            const uint8_t* t = src + (((size_t)(z % n) * n + (y % n)) * n + (x % n)) * 3;
            return 5 * t[0] + 2 * t[1] + t[2];                                        // hfbm numerator (clouds.glsl:133 x 8)
        };
        for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
            int v[8], c[8];
            for (int k = 0; k < 8; k++) v[k] = num(x + (k & 1), y + ((k >> 1) & 1), z + (k >> 2));
            cell_coeffs(v, c);
            out[off[l] + ((size_t)z * n + y) * n + x] = uint4{hpair_i(c[0], c[1], inexact), hpair_i(c[2], c[3], inexact), hpair_i(c[4], c[5], inexact), hpair_i(c[6], c[7], inexact)};
        }
    }
}
// unpacked fp16 numerators of the whole detail chain (the source of the LDS copy of the "lds" kernel variant)
inline void bake_detail_unpacked(const std::vector<uint8_t>& chain, std::vector<uint16_t>& out) {
    out.clear();
    for (int l = 0; l < DETAIL_LEVELS; l++) {
        const size_t n = DETAIL_N >> l;
        const uint8_t* src = chain.data() + csky_mip_offset(DETAIL_N, l, 3);
        for (size_t i = 0; i < n * n * n; i++) out.push_back(f2h((float)(5 * src[3 * i] + 2 * src[3 * i + 1] + src[3 * i + 2])));
    }
}
inline void bake_weather(const uint8_t* rgb, std::vector<uint4>& out, unsigned long long* inexact = nullptr) {
    const int n = WEATHER_N;
    out.resize((size_t)n * n);
    auto ch = [&](int x, int y, int c) -> int { return rgb[(((size_t)(y % n)) * n + (x % n)) * 3 + c]; };
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
        uint4 q;
        for (int c = 0; c < 2; c++) {                                  // R: cloud type, B: coverage
            const int k = 2 * c, v00 = ch(x, y, k), v10 = ch(x + 1, y, k), v01 = ch(x, y + 1, k), v11 = ch(x + 1, y + 1, k);
            const uint32_t p0 = hpair_i(v00, v10 - v00, inexact), p1 = hpair_i(v01 - v00, v11 - v01 - v10 + v00, inexact);
            if (c == 0) { q.x = p0; q.y = p1; } else { q.z = p0; q.w = p1; }
        }
        out[(size_t)y * n + x] = q;
    }
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
