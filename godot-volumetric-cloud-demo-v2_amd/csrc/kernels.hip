// kernels.hip -- gfx950 (CDNA4, wave64) kernels of libcloudsky.
//
//   transmittance_kernel : transmittance-lut.glsl, one texel per lane                 (16 384 lanes, once)
//   sky_lut_kernel       : sky-lut.glsl, one texel per lane                           (20 000 lanes, per sun change)
//   frame_setup_kernel   : the ray-invariant prologue of clouds.glsl march()          (1 lane, per frame)
//   clouds_kernel        : clouds.glsl main(): one ray per lane, one 8x8-pixel tile per wavefront
//
// No MFMA anywhere: the path is fetch/latency-bound gather + fp32 VALU, not a contraction.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "cloud_core.h"
#include "lut_core.h"

namespace csky {

// ------------------------------------------------------------------------------------------------ LUTs
__global__ __launch_bounds__(64) void transmittance_kernel(int w, int h, uint16_t* __restrict__ out_h, float4* __restrict__ out_f) {
    const int px = blockIdx.x * 8 + (threadIdx.x & 7), py = blockIdx.y * 8 + (threadIdx.x >> 3);  // 8x8 groups, T:5
    if (px >= w || py >= h) return;  // (T:159 tests `>`; the extra row/column would be an out-of-image store)
    const F4 t = transmittance_texel(px, py, (float)w, (float)h);
    const uint16_t hx = f2h(t.x), hy = f2h(t.y), hz = f2h(t.z), hw = f2h(t.w);
    const size_t i = (size_t)py * w + px;
    reinterpret_cast<uint2*>(out_h)[i] = make_uint2((uint32_t)hx | ((uint32_t)hy << 16), (uint32_t)hz | ((uint32_t)hw << 16));
    out_f[i] = make_float4(h2f(hx), h2f(hy), h2f(hz), h2f(hw));
}

struct Sun3 { float v[3]; };
__global__ __launch_bounds__(64) void sky_lut_kernel(int w, int h, Sun3 sun, const float4* __restrict__ trans, int tw, int th,
                                                     uint16_t* __restrict__ out_h, float4* __restrict__ out_f) {
    const int px = blockIdx.x * 8 + (threadIdx.x & 7), py = blockIdx.y * 8 + (threadIdx.x >> 3);  // dispatch 25x13, sky_lut.gd:140
    if (px >= w || py >= h) return;  // rows 100..103 of the reference dispatch are discarded image stores (S:281)
    const F4 c = sky_texel(px, py, (float)w, (float)h, sun.v, trans, tw, th);
    const uint16_t hx = f2h(c.x), hy = f2h(c.y), hz = f2h(c.z), hw = f2h(c.w);
    const size_t i = (size_t)py * w + px;
    reinterpret_cast<uint2*>(out_h)[i] = make_uint2((uint32_t)hx | ((uint32_t)hy << 16), (uint32_t)hz | ((uint32_t)hw << 16));
    out_f[i] = make_float4(h2f(hx), h2f(hy), h2f(hz), h2f(hw));
}

hipError_t launch_transmittance(int w, int h, uint16_t* d_half, float4* d_float, hipStream_t s) {
    transmittance_kernel<<<dim3((w + 7) / 8, (h + 7) / 8), 64, 0, s>>>(w, h, d_half, d_float);
    return hipGetLastError();
}
hipError_t launch_sky_lut(int w, int h, const float sun[3], const float4* d_trans, int tw, int th, uint16_t* d_half, float4* d_float,
                          hipStream_t s) {
    Sun3 sv; sv.v[0] = sun[0]; sv.v[1] = sun[1]; sv.v[2] = sun[2];
    sky_lut_kernel<<<dim3((w + 7) / 8, (h + 7) / 8), 64, 0, s>>>(w, h, sv, d_trans, tw, th, d_half, d_float);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ clouds
__global__ __launch_bounds__(64) void frame_setup_kernel(CloudParams p, const float4* __restrict__ sky, int sw, int sh, int primary_steps,
                                                         int light_steps, float early_eps, FrameConsts* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        FrameConsts fc;
        frame_setup(p, sky, sw, sh, primary_steps, light_steps, early_eps, fc);
        *out = fc;
    }
}
hipError_t launch_frame_setup(const CloudParams& p, const float4* d_sky, int sw, int sh, int primary_steps, int light_steps, float early_eps,
                              FrameConsts* d_fc, hipStream_t s) {
    frame_setup_kernel<<<1, 64, 0, s>>>(p, d_sky, sw, sh, primary_steps, light_steps, early_eps, d_fc);
    return hipGetLastError();
}

// Pixel <-> lane mapping: a 256-thread workgroup = 4 wavefronts = a 32 x 8 pixel slab; each wavefront owns
// one 8x8 tile (lane = ly*8 + lx) so its 64 rays are angularly adjacent: their texture footprints overlap
// (L1/TA coalescing) and they enter/leave cloud together (less divergence).  The reference uses the same
// 8x8 footprint per workgroup (clouds.glsl:5).
// XCD-aware order: workgroup b runs on XCD b % 8 (observed, speed only); the remap gives every XCD one
// contiguous eighth of the frame, so the slice of the noise volumes its rays touch stays in ITS 4 MiB L2.
template <int VARIANT>
__global__ __launch_bounds__(256) void clouds_kernel(TexSet T, const FrameConsts* __restrict__ fcp, RenderGeom G, uint2* __restrict__ out,
                                                     unsigned long long* __restrict__ stats) {
    const int tiles_x = (G.tile_w + 31) >> 5;
    const int local_rows = G.n_bands * G.band_rows;
    const int slabs = (local_rows + 7) >> 3;
    const int nblocks = tiles_x * slabs;
    const int per_xcd = (nblocks + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= nblocks) return;
    const int slab = logical / tiles_x, bx = logical - slab * tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gx = bx * 32 + wave * 8 + (lane & 7);
    const int lr = slab * 8 + (lane >> 3);
    const bool valid = gx < G.tile_w && lr < local_rows;
    const int band = lr / G.band_rows, rib = lr - band * G.band_rows;
    const int gy = (G.first_band + band * G.band_stride) * G.band_rows + rib;

    const FrameConsts& fc = *fcp;
    Ray ray = ray_setup(fc, valid ? gx : 0, valid ? gy : 0);
    if (!valid) ray.above = false;
    const MarchOut o = march(T, fc, ray);
    if (valid) {
        const uint32_t lo = (uint32_t)f2h(o.r) | ((uint32_t)f2h(o.g) << 16), hi = (uint32_t)f2h(o.b) | ((uint32_t)f2h(o.a) << 16);
        out[(size_t)lr * G.pitch_px + gx] = make_uint2(lo, hi);  // imageStore, clouds.glsl:264
    }
    if (stats) {
        unsigned ic = o.incloud, ab = ray.above ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) { ic += __shfl_down(ic, off); ab += __shfl_down(ab, off); }
        if (lane == 0) { atomicAdd(&stats[0], (unsigned long long)ic); atomicAdd(&stats[1], (unsigned long long)ab); }
    }
}

static const char* const kVariantNames[] = {"lockstep"};
int cloud_variant_count() { return (int)(sizeof(kVariantNames) / sizeof(kVariantNames[0])); }
const char* cloud_variant_name(int v) { return (v >= 0 && v < cloud_variant_count()) ? kVariantNames[v] : nullptr; }

hipError_t launch_clouds(int variant, const TexSet& t, const FrameConsts* d_fc, const RenderGeom& g, uint2* d_out, unsigned long long* d_stats,
                         hipStream_t s) {
    const int tiles_x = (g.tile_w + 31) >> 5, slabs = (g.n_bands * g.band_rows + 7) >> 3;
    const int nblocks = tiles_x * slabs;
    if (nblocks <= 0) return hipSuccess;
    const int grid = ((nblocks + 7) >> 3) << 3;
    switch (variant) {
        case 0: clouds_kernel<0><<<grid, 256, 0, s>>>(t, d_fc, g, d_out, d_stats); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace csky
