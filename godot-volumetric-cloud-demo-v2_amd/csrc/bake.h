// bake.h -- host-side conversion of the 8-bit mip chains into the device texture layouts of csky_common.h.
// Internal to libcloudsky (api.cpp); also included by the host-compiled kernel-core unit test (tests/hostsim).
#pragma once
#include <vector>
#include "csky_common.h"
#include "../../include/cloudsky.h"

namespace csky {
// ---- texture baking (layouts documented in csky_common.h) ----
inline void bake_shape(const std::vector<uint8_t>& chain, std::vector<uint2>& out, uint32_t off[SHAPE_LEVELS]) {
    size_t total = 0;
    for (int l = 0; l < SHAPE_LEVELS; l++) { off[l] = (uint32_t)total; size_t n = SHAPE_N >> l; total += n * n * n; }
    out.resize(total);
    for (int l = 0; l < SHAPE_LEVELS; l++) {
        const int n = SHAPE_N >> l;
        const uint8_t* src = chain.data() + csky_mip_offset(SHAPE_N, l, 4);
        auto tex = [&](int x, int y, int z) -> uint32_t {
            const uint8_t* t = src + (((size_t)z * n + y) * n + x) * 4;
            return (uint32_t)t[0] | ((uint32_t)(5 * t[1] + 2 * t[2] + t[3]) << 16);   // r | fbm numerator (clouds.glsl:118 x 8)
        };
        for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++)
            out[off[l] + ((size_t)z * n + y) * n + x] = uint2{tex(x, y, z), tex((x + 1) % n, y, z)};
    }
}
inline void bake_detail(const std::vector<uint8_t>& chain, std::vector<uint4>& out, uint32_t off[DETAIL_LEVELS]) {
    size_t total = 0;
    for (int l = 0; l < DETAIL_LEVELS; l++) { off[l] = (uint32_t)total; size_t n = DETAIL_N >> l; total += n * n * n; }
    out.resize(total);
    for (int l = 0; l < DETAIL_LEVELS; l++) {
        const int n = DETAIL_N >> l;
        const uint8_t* src = chain.data() + csky_mip_offset(DETAIL_N, l, 3);
        auto num = [&](int x, int y, int z) -> uint32_t {
            const uint8_t* t = src + (((size_t)(z % n) * n + (y % n)) * n + (x % n)) * 3;
            return (uint32_t)(5 * t[0] + 2 * t[1] + t[2]);                            // hfbm numerator (clouds.glsl:133 x 8)
        };
        for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
            uint4 q;
            q.x = num(x, y, z) | (num(x + 1, y, z) << 16);
            q.y = num(x, y + 1, z) | (num(x + 1, y + 1, z) << 16);
            q.z = num(x, y, z + 1) | (num(x + 1, y, z + 1) << 16);
            q.w = num(x, y + 1, z + 1) | (num(x + 1, y + 1, z + 1) << 16);
            out[off[l] + ((size_t)z * n + y) * n + x] = q;
        }
    }
}
inline void bake_weather(const uint8_t* rgb, std::vector<uint2>& out) {
    const int n = WEATHER_N;
    out.resize((size_t)n * n);
    auto ch = [&](int x, int y, int c) -> uint32_t { return rgb[(((size_t)(y % n)) * n + (x % n)) * 3 + c]; };
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
        uint2 q;
        q.x = ch(x, y, 0) | (ch(x + 1, y, 0) << 8) | (ch(x, y + 1, 0) << 16) | (ch(x + 1, y + 1, 0) << 24);   // R: cloud type
        q.y = ch(x, y, 2) | (ch(x + 1, y, 2) << 8) | (ch(x, y + 1, 2) << 16) | (ch(x + 1, y + 1, 2) << 24);   // B: coverage
        out[(size_t)y * n + x] = q;
    }
}
}  // namespace csky
