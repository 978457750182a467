// composite_core.h -- per-pixel body of the sky compositor: clouds.gdshader sky() (cited G:line), SURVEY §8f row 1.
// Godot supplies EYEDIR per screen pixel; here sky() is evaluated for the pixels of an equirectangular panorama
// (u -> azimuth (2u-1)*pi, v -> elevation (0.5-v)*pi; EYEDIR = (cos e cos a, sin e, cos e sin a), y up).
// Host+device like the other cores; the product instantiates it only in kernels.hip.  Small kernel: accurate OCML
// maths, FP contraction off.
#pragma once
#include "csky_common.h"

namespace csky {
#pragma clang fp contract(off)

constexpr float G_PI = 3.14159265358979323846f;     // the shading language's built-in PI (full precision, unlike clouds.glsl:47)

struct CompositeArgs {
    const uint16_t* cloud_from; const uint16_t* cloud_to; int cw, ch;   // blend_from_texture / blend_to_texture (G:4-5), RGBA16F
    const uint16_t* sky_from; const uint16_t* sky_to; int sw, sh;       // sky_blend_from_texture / sky_blend_to_texture (G:7-8)
    const float4* trans; int tw, th;                                    // source_transmittance (G:10), fp16 values widened
    float blend_amount, sun_disk_scale;                                 // G:12-13
    float sun[3];                                                       // LIGHT0_DIRECTION
    int out_w, out_h;
    // projection of output pixel -> EYEDIR: 0 = equirectangular panorama; 1 = perspective camera (what the engine feeds the sky shader: one
    // EYEDIR per SCREEN pixel, clouds.gdshader:105-116): cam = Camera3D.global_transform.basis columns (x right, y up, z back), Godot's
    // vertical field of view and the viewport's aspect ratio
    int view_mode;
    float cam[9];                                                       // column-major: cam[0..2] = basis.x, [3..5] = basis.y, [6..8] = basis.z
    float tan_half_fov_y, aspect;
};

struct C3 { float x, y, z; };
struct C4 { float x, y, z, w; };

// filter_linear + repeat_disable tap of an RGBA16F image
CSKY_HD C4 tap_half_clamp(const uint16_t* t, int w, int h, float sx, float sy) {
    const float ux = sx * (float)w - 0.5f, uy = sy * (float)h - 0.5f;
    const float fx0 = floorf(ux), fy0 = floorf(uy), ax = ux - fx0, ay = uy - fy0;
    int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > w - 1 ? w - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
    const uint16_t *a = t + ((size_t)y0 * w + x0) * 4, *b = t + ((size_t)y0 * w + x1) * 4, *c = t + ((size_t)y1 * w + x0) * 4, *d = t + ((size_t)y1 * w + x1) * 4;
    C4 r;
    r.x = lerpf(lerpf(h2f(a[0]), h2f(b[0]), ax), lerpf(h2f(c[0]), h2f(d[0]), ax), ay);
    r.y = lerpf(lerpf(h2f(a[1]), h2f(b[1]), ax), lerpf(h2f(c[1]), h2f(d[1]), ax), ay);
    r.z = lerpf(lerpf(h2f(a[2]), h2f(b[2]), ax), lerpf(h2f(c[2]), h2f(d[2]), ax), ay);
    r.w = lerpf(lerpf(h2f(a[3]), h2f(b[3]), ax), lerpf(h2f(c[3]), h2f(d[3]), ax), ay);
    return r;
}
CSKY_HD float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
CSKY_HD float smoothstepf(float e0, float e1, float x) { const float t = sat((x - e0) / (e1 - e0)); return t * t * (3.0f - 2.0f * t); }

// EYEDIR of output pixel (i, j)
CSKY_HD void composite_eyedir(const CompositeArgs& A, int i, int j, float& ex, float& ey, float& ez) {
    const float u = ((float)i + 0.5f) / (float)A.out_w, v = ((float)j + 0.5f) / (float)A.out_h;
    if (A.view_mode == 1) {
        // screen pixel -> view-space ray (x right, y up, looking down -z) -> world space through the camera basis, normalised: the EYEDIR the
        // engine hands a sky shader for this screen pixel
        const float vx = (u * 2.0f - 1.0f) * A.tan_half_fov_y * A.aspect, vy = (1.0f - v * 2.0f) * A.tan_half_fov_y, vz = -1.0f;
        const float wx = A.cam[0] * vx + A.cam[3] * vy + A.cam[6] * vz, wy = A.cam[1] * vx + A.cam[4] * vy + A.cam[7] * vz, wz = A.cam[2] * vx + A.cam[5] * vy + A.cam[8] * vz;
        const float l = sqrtf(wx * wx + wy * wy + wz * wz);
        ex = wx / l; ey = wy / l; ez = wz / l;
        return;
    }
    const float az = (u * 2.0f - 1.0f) * G_PI, el = (0.5f - v) * G_PI;
    ex = cosf(el) * cosf(az); ey = sinf(el); ez = cosf(el) * sinf(az);
}

CSKY_HD C3 composite_pixel(const CompositeArgs& A, int i, int j) {
    float ex, ey, ez;
    composite_eyedir(A, i, j, ex, ey, ez);                                               // EYEDIR
    // G:106-110: clamp below the horizon, hemi-octahedral encode of norm.xzy
    float nx = ex, ny = fmaxf(0.0f, ey), nz = ez;
    const float nl = sqrtf(nx * nx + ny * ny + nz * nz);
    nx = nx / nl; ny = ny / nl; nz = nz / nl;
    float ox = nx, oy = nz, oz = ny;                                                     // e = norm.xzy
    const float dsum = fabsf(ox) + fabsf(oy) + fabsf(oz);                                // G:23
    ox = ox / dsum; oy = oy / dsum; oz = oz / dsum;
    if (!(oz >= 0.0f)) {                                                                 // G:24 (dead: norm.y >= 0)
        const float sx = ox >= 0.0f ? 1.0f : -1.0f, sy = oy >= 0.0f ? 1.0f : -1.0f;
        const float wx = (1.0f - fabsf(oy)) * sx, wy = (1.0f - fabsf(ox)) * sy;
        ox = wx; oy = wy;
    }
    float uvy = oy * 0.5f + 0.5f;                                                        // G:27-29
    const float uvx = ox * 0.5f + uvy;
    uvy = ox * -0.5f + uvy;
    const C4 bf = tap_half_clamp(A.cloud_from, A.cw, A.ch, uvx, uvy), bt = tap_half_clamp(A.cloud_to, A.cw, A.ch, uvx, uvy);   // G:111-112
    const float cr = mixf(bf.x, bt.x, A.blend_amount), cg = mixf(bf.y, bt.y, A.blend_amount), cb = mixf(bf.z, bt.z, A.blend_amount),
                ca = mixf(bf.w, bt.w, A.blend_amount);                                   // G:113
    // get_atmo(EYEDIR), G:87-103
    const float phi = atan2f(ez, ex), theta = asinf(ey);                                 // G:34-45
    const float sux = (phi / G_PI * 0.5f + 0.5f);
    const float suy = sqrtf(fabsf(theta) / (G_PI * 0.5f)) * signf(theta) * 0.5f + 0.5f;
    const C4 sf = tap_half_clamp(A.sky_from, A.sw, A.sh, sux, suy), st = tap_half_clamp(A.sky_to, A.sw, A.sh, sux, suy);
    float br = mixf(sf.x, st.x, A.blend_amount) / 50.0f, bg = mixf(sf.y, st.y, A.blend_amount) / 50.0f, bb = mixf(sf.z, st.z, A.blend_amount) / 50.0f;
    const float sunSolidAngle = A.sun_disk_scale * 0.53f * G_PI / 180.0f;                // G:49
    const float minSunCosTheta = cosf(sunSolidAngle);
    const float cosTheta = ex * A.sun[0] + ey * A.sun[1] + ez * A.sun[2];
    float sl;
    if (cosTheta >= minSunCosTheta) sl = 1.0f;
    else {
        const float offset = minSunCosTheta - cosTheta;
        const float gaussianBloom = expf(-offset * 50000.0f) * 0.5f;
        const float invBloom = 1.0f / (0.02f + offset * 300.0f) * 0.01f;
        sl = gaussianBloom + invBloom;
    }
    sl = smoothstepf(0.002f, 1.0f, sl);                                                  // G:91
    if (sqrtf(sl * sl + sl * sl + sl * sl) > 0.0f) {                                     // G:92
        const float vpy = (float)(6.360 + 0.0002);                                       // viewPos, G:74: a constant expression, glslang folds it in double and narrows once
        // rayIntersectSphere(viewPos, dir, groundRadiusMM) >= 0, G:61-70,93
        const float b = 0.0f * ex + vpy * ey + 0.0f * ez, c = (0.0f * 0.0f + vpy * vpy + 0.0f * 0.0f) - 6.360f * 6.360f;
        float hit;
        if (c > 0.0f && b > 0.0f) hit = -1.0f;
        else { const float discr = b * b - c; hit = discr < 0.0f ? -1.0f : (discr > b * b ? (-b + sqrtf(discr)) : (-b - sqrtf(discr))); }
        if (hit >= 0.0f) { br += sl * 0.0f; bg += sl * 0.0f; bb += sl * 0.0f; }
        else {
            // getValFromTLUT(source_transmittance, tLUTRes, viewPos, LIGHT0_DIRECTION), G:77-85
            const float height = sqrtf(0.0f * 0.0f + vpy * vpy + 0.0f * 0.0f);
            const float sunCos = (0.0f / height) * A.sun[0] + (vpy / height) * A.sun[1] + (0.0f / height) * A.sun[2];
            float tux = 256.0f * clampf(0.5f + 0.5f * sunCos, 0.0f, 1.0f);
            float tuy = 64.0f * fmaxf(0.0f, fminf(1.0f, (height - 6.360f) / (float)(6.460 - 6.360)));   // (the constant difference folds to 0.1, not to 6.46f - 6.36f)
            tux /= 256.0f; tuy /= 64.0f;
            const float ux = tux * (float)A.tw - 0.5f, uy = tuy * (float)A.th - 0.5f;
            const float fx0 = floorf(ux), fy0 = floorf(uy), ax = ux - fx0, ay = uy - fy0;
            int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
            x0 = x0 < 0 ? 0 : (x0 > A.tw - 1 ? A.tw - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > A.tw - 1 ? A.tw - 1 : x1);
            y0 = y0 < 0 ? 0 : (y0 > A.th - 1 ? A.th - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > A.th - 1 ? A.th - 1 : y1);
            const float4 p = A.trans[y0 * A.tw + x0], q = A.trans[y0 * A.tw + x1], r = A.trans[y1 * A.tw + x0], s = A.trans[y1 * A.tw + x1];
            br += sl * lerpf(lerpf(p.x, q.x, ax), lerpf(r.x, s.x, ax), ay);             // .rgb of the 4-wavelength LUT (reference quirk)
            bg += sl * lerpf(lerpf(p.y, q.y, ax), lerpf(r.y, s.y, ax), ay);
            bb += sl * lerpf(lerpf(p.z, q.z, ax), lerpf(r.z, s.z, ax), ay);
        }
    }
    // G:114-115
    const float k = smoothstepf(0.6f, 1.0f, 1.0f - ey);
    C3 o;
    o.x = mixf(clampf(br * (1.0f - ca) + cr, 0.0f, 100.0f), clampf(br, 0.0f, 100.0f), k);
    o.y = mixf(clampf(bg * (1.0f - ca) + cg, 0.0f, 100.0f), clampf(bg, 0.0f, 100.0f), k);
    o.z = mixf(clampf(bb * (1.0f - ca) + cb, 0.0f, 100.0f), clampf(bb, 0.0f, 100.0f), k);
    return o;
}

}  // namespace csky
