// brick_bound.cpp -- ANALYSIS TOOL (host, g++): how many of the primary samples that pay the 32-byte shape gather and are then rejected
// at exact reject (2) (cloud_core.h::density: base*g - (1 - wc) <= 0) could a CONSERVATIVE per-brick bound have rejected without the tap?
//   base = (r + 1 - fbm) / (2 - fbm) is increasing in r and decreasing in fbm, and a trilinear tap lies between its
// eight corners, so with rmax / fmin over the corners a sample's cell can touch,  base <= (rmax + 1 - fmin) / (2 - fmin)  =: bmax  and
// bmax*g <= 1 - wc proves density() == 0.  Bricks of B^3 texels with a one-texel apron on the high side (the cell of a base index inside
// the brick reads texels idx and idx+1).  Counts, per brick size: samples reaching the shape tap, rejected there, and provably rejected.
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../godot-volumetric-cloud-demo-v2_amd/csrc/cloud_core.h"
#include "../../godot-volumetric-cloud-demo-v2_amd/csrc/lut_core.h"
#include "../../godot-volumetric-cloud-demo-v2_amd/csrc/bake.h"
using namespace csky;
size_t csky_mip_offset(int n, int level, int ch) { size_t off = 0; for (int l = 0; l < level; l++) { size_t m = (size_t)(n >> l); off += m * m * m * (size_t)ch; } return off; }

extern "C" void brick_bound(const uint8_t* large_chain, const uint8_t* small_chain, const uint8_t* weather_rgb8, const float params[28], int primary_steps,
                            int w, int h, const int* brick_sizes, int n_sizes, uint64_t* out /* [n_sizes][4]: reach shape tap, rejected at (2), provably rejected, false rejects (must be 0) */) {
    std::vector<uint8_t> lc(large_chain, large_chain + csky_mip_offset(SHAPE_N, SHAPE_LEVELS, 4));
    std::vector<uint8_t> sc(small_chain, small_chain + csky_mip_offset(DETAIL_N, DETAIL_LEVELS, 3));
    std::vector<ShapeTexel> shape; std::vector<uint4> detail, weather;
    uint32_t so[SHAPE_LEVELS], dof[DETAIL_LEVELS];
    bake_shape(lc, shape, so); bake_detail(sc, detail, dof); bake_weather(weather_rgb8, weather);
    const int tw = 256, th = 64, sw = 200, sh = 100;
    std::vector<float4> tf((size_t)tw * th), sky((size_t)sw * sh);
    for (int y = 0; y < th; y++) for (int x = 0; x < tw; x++) { F4 t = transmittance_texel(x, y, (float)tw, (float)th); tf[(size_t)y * tw + x] = float4{h2f(f2h(t.x)), h2f(f2h(t.y)), h2f(f2h(t.z)), h2f(f2h(t.w))}; }
    CloudParams P; memcpy(&P, params, sizeof P);
    const float sun[3] = {P.LIGHT_DIRECTION[0], P.LIGHT_DIRECTION[1], P.LIGHT_DIRECTION[2]};
    for (int y = 0; y < sh; y++) for (int x = 0; x < sw; x++) { F4 c = sky_texel(x, y, (float)sw, (float)sh, sun, tf.data(), tw, th); sky[(size_t)y * sw + x] = float4{h2f(f2h(c.x)), h2f(f2h(c.y)), h2f(f2h(c.z)), h2f(f2h(c.w))}; }
    TexSet T; T.shape = shape.data(); T.detail = detail.data(); T.weather = weather.data(); T.sky = sky.data(); T.sky_w = sw; T.sky_h = sh; T.detail_h = nullptr; T.detail_lds = nullptr;
    { const uint8_t* t5 = sc.data() + csky_mip_offset(DETAIL_N, 5, 3); T.detail_lod5 = (float)(5 * t5[0] + 2 * t5[1] + t5[2]) * (1.0f / (8.0f * 255.0f)); }
    int rmin = 255, rmax = 0, bmax = 0;
    for (size_t i = 0; i < (size_t)WEATHER_N * WEATHER_N; i++) { const int r = weather_rgb8[3 * i], b = weather_rgb8[3 * i + 2]; rmin = std::min(rmin, r); rmax = std::max(rmax, r); bmax = std::max(bmax, b); }
    float hlo, hhi;
    height_window((double)P.cloud_coverage, rmin / 255.0, rmax / 255.0, bmax / 255.0, hlo, hhi);
    FrameConsts fc;
    frame_setup(P, sky.data(), sw, sh, primary_steps, 6, 0.0f, hlo, hhi, fc);
    fc.ct_mode = rmin >= 128 ? 1 : (rmax <= 127 ? 2 : 0);
    // per-brick (max r, min fbm numerator) tables of level 0
    std::vector<std::vector<float>> tab_r(n_sizes), tab_f(n_sizes);
    for (int s = 0; s < n_sizes; s++) {
        const int B = brick_sizes[s], nb = SHAPE_N / B;
        tab_r[s].assign((size_t)nb * nb * nb, 0.0f); tab_f[s].assign((size_t)nb * nb * nb, 0.0f);
        for (int bz = 0; bz < nb; bz++) for (int by = 0; by < nb; by++) for (int bx = 0; bx < nb; bx++) {
            int rm = 0, fm = 1 << 30;
            for (int z = bz * B; z <= bz * B + B; z++) for (int y = by * B; y <= by * B + B; y++) for (int x = bx * B; x <= bx * B + B; x++) {
                const uint8_t* t = lc.data() + ((((size_t)(z & 127) * 128 + (y & 127)) * 128 + (x & 127)) * 4);
                rm = std::max(rm, (int)t[0]); fm = std::min(fm, 5 * t[1] + 2 * t[2] + t[3]);
            }
            tab_r[s][((size_t)bz * nb + by) * nb + bx] = rm * (1.0f / 255.0f);
            tab_f[s][((size_t)bz * nb + by) * nb + bx] = fm * (1.0f / (8.0f * 255.0f));
        }
    }
    for (int k = 0; k < 4 * n_sizes; k++) out[k] = 0;
    for (int gy = 0; gy < h; gy++) for (int gx = 0; gx < w; gx++) {
        Ray ray = ray_setup(fc, gx, gy);
        if (!ray.above) continue;
        float px = ray.px, py = ray.py, pz = ray.pz;
        for (int i = 0; i < primary_steps; i++) {
            advance(px, py, pz, ray.sx, ray.sy, ray.sz);
            const float hf = height_fraction(length3_exact(px, py, pz));
            if (!(hf > fc.hf_lo && hf < fc.hf_hi)) continue;
            float wsx, wsy, wr, wb;
            weather_coord(px, pz, fc.wpos_x, fc.wpos_y, wsx, wsy);
            weather_tap(T.weather, wsx, wsy, wr, wb);
            const float wc = fc.cov255 * wb, g = density_height_gradient(fc, hf, wr), omw = 1.0f - wc;
            if (!(g > omw)) continue;                                         // exact reject (1): no shape tap
            float qx, qy, qz, sx, sy, sz, nr, fbm;
            shape_coord(fc, px, py, pz, qx, qy, qz, sx, sy, sz);
            shape_tap(T, 0, sx, sy, sz, nr, fbm);
            const float omf = 1.0f - fbm;
            const float base = (nr + omf) * fast_rcp(1.0f + omf) * g - omw;
            const bool rejected = !(base > 0.0f);
            int ix, iy, iz; float ax, ay, az;
            split_coord(sx * 128.0f - 0.5f, ix, ax); split_coord(sy * 128.0f - 0.5f, iy, ay); split_coord(sz * 128.0f - 0.5f, iz, az);
            for (int s = 0; s < n_sizes; s++) {
                const int B = brick_sizes[s], nb = SHAPE_N / B;
                const size_t bi = ((size_t)((iz & 127) / B) * nb + ((iy & 127) / B)) * nb + ((ix & 127) / B);
                const float rm = tab_r[s][bi], fm = tab_f[s][bi];
                const float bmaxv = (rm + (1.0f - fm)) / (2.0f - fm) * 1.000001f;   // one ulp-scale margin: the bound must hold in fp32 too
                const bool proven = !(bmaxv * g - omw > 0.0f);
                out[4 * s + 0]++;
                if (rejected) out[4 * s + 1]++;
                if (proven) out[4 * s + 2]++;
                if (proven && !rejected) out[4 * s + 3]++;
            }
        }
    }
}
