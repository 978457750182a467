// kernels.hip -- gfx950 (CDNA4, wave64) kernels of libcloudsky.
//
//   transmittance_kernel : transmittance-lut.glsl, one texel per wavefront, 40 steps on 40 lanes   (16 384 wavefronts, once)
//   sky_lut_kernel       : sky-lut.glsl, one texel per half wavefront, 30 steps on 30 lanes       (20 000 texels, per sun change)
//   frame_setup_kernel   : the ray-invariant prologue of clouds.glsl march()          (1 lane, per frame)
//   clouds_kernel        : clouds.glsl main(): one ray per lane, one 8x8-pixel tile per wavefront
//
// No MFMA anywhere: the path is fetch/latency-bound gather + fp32 VALU, not a contraction.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "cloud_core.h"
#include "lut_core.h"
#include "composite_core.h"
#include "noise_core.h"
#include "bake_core.h"

namespace csky {

// ------------------------------------------------------------------------------------------------ LUTs
// transmittance-lut.glsl: one texel per wavefront: lanes 0..39 evaluate the 40 optical-depth steps in parallel (each ~150 VALU with five
// correctly rounded transcendentals, independent of the others), park extinction * dt in LDS, then lane 0 replays the sum in the reference's
// order (T:186-192; bit-identical to the one-lane-per-texel form, 40x shorter critical path).  Round 1 had the GLSL's own dispatch shape here
// (8x8 groups, one texel per lane, a 40-step serial loop: 2 048 one-wave groups on 6 % of the chip).
__global__ __launch_bounds__(256) void transmittance_kernel(int w, int h, uint16_t* __restrict__ out_h, float4* __restrict__ out_f) {
    __shared__ float terms[4][TRANSMITTANCE_STEPS][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int texel = blockIdx.x * 4 + wave;
    const bool live = texel < w * h;                              // (T:159 tests `>`; the extra row/column would be an out-of-image store)
    const int px = live ? texel % w : 0, py = live ? texel / w : 0;
    const TransRay r = transmittance_ray(px, py, (float)w, (float)h);
    if (lane < TRANSMITTANCE_STEPS) {
        const F4 e = transmittance_step(r, lane);
        float* d = terms[wave][lane];
        d[0] = e.x; d[1] = e.y; d[2] = e.z; d[3] = e.w;
    }
    __syncthreads();
    if (lane == 0 && live) {
        F4 result = f4(0, 0, 0, 0);
        for (int i = 0; i < TRANSMITTANCE_STEPS; ++i) { const float* d = terms[wave][i]; result = result + f4(d[0], d[1], d[2], d[3]); }
        const F4 t = transmittance_finish(result);
        const uint16_t hx = f2h(t.x), hy = f2h(t.y), hz = f2h(t.z), hw = f2h(t.w);
        reinterpret_cast<uint2*>(out_h)[texel] = make_uint2((uint32_t)hx | ((uint32_t)hy << 16), (uint32_t)hz | ((uint32_t)hw << 16));
        out_f[texel] = make_float4(h2f(hx), h2f(hy), h2f(hz), h2f(hw));
    }
}

struct Sun3 { float v[3]; };
// sky-lut.glsl: one texel per HALF wavefront: lanes 0..29 evaluate the 30 in-scattering steps in parallel (each step is
// ~600 VALU with 20 LUT loads and 12 transcendentals and independent of the others), park source term + transmittance
// in LDS, then lane 0 of the half replays the front-to-back accumulation in the reference's order (bit-identical to the
// one-lane-per-texel form, 4x shorter critical path: this kernel sits on the critical path of every frame).
// One texel; store(px, py, hx, hy, hz, hw) takes the four fp16 values (lane 0 of the half wavefront, live texels only).
template <class Store> __device__ __forceinline__ void sky_texel(float (*steps)[8], int sub, bool live, int px, int py, int w, int h, const Sun3& sun,
                                                                 const float4* __restrict__ trans, int tw, int th, Store store) {
    const SkyRay r = sky_ray(px, py, (float)w, (float)h, sun.v);
    if (sub < IN_SCATTERING_STEPS) {
        const SkyStep s = sky_step(r, sub, trans, tw, th);
        float* d = steps[sub];
        d[0] = s.S_int.x; d[1] = s.S_int.y; d[2] = s.S_int.z; d[3] = s.S_int.w;
        d[4] = s.step_tr.x; d[5] = s.step_tr.y; d[6] = s.step_tr.z; d[7] = s.step_tr.w;
    }
    __syncthreads();
    if (sub == 0 && live) {
        F4 L = f4(0, 0, 0, 0), Tr = f4(1, 1, 1, 1);
        for (int i = 0; i < IN_SCATTERING_STEPS; ++i) {
            const float* d = steps[i];
            SkyStep s; s.S_int = f4(d[0], d[1], d[2], d[3]); s.step_tr = f4(d[4], d[5], d[6], d[7]);
            sky_accumulate(L, Tr, s);
        }
        const F4 c = sky_output(L);
        store(px, py, f2h(c.x), f2h(c.y), f2h(c.z), f2h(c.w));
    }
}
__device__ __forceinline__ uint2 pack_half4(uint16_t hx, uint16_t hy, uint16_t hz, uint16_t hw) {
    return make_uint2((uint32_t)hx | ((uint32_t)hy << 16), (uint32_t)hz | ((uint32_t)hw << 16));
}
__global__ __launch_bounds__(256) void sky_lut_kernel(int w, int h, Sun3 sun, const float4* __restrict__ trans, int tw, int th,
                                                     uint16_t* __restrict__ out_h, float4* __restrict__ out_f) {
    __shared__ float steps[8][IN_SCATTERING_STEPS][8];
    const int half = threadIdx.x >> 5, sub = threadIdx.x & 31;
    const int texel = blockIdx.x * 8 + half;                      // rows 100..103 of the reference dispatch are discarded stores (S:281)
    const bool live = texel < w * h;
    const int px = live ? texel % w : 0, py = live ? texel / w : 0;
    sky_texel(steps[half], sub, live, px, py, w, h, sun, trans, tw, th, [=](int x, int y, uint16_t hx, uint16_t hy, uint16_t hz, uint16_t hw) {
        reinterpret_cast<uint2*>(out_h)[y * w + x] = pack_half4(hx, hy, hz, hw);
        out_f[y * w + x] = make_float4(h2f(hx), h2f(hy), h2f(hz), h2f(hw));
    });
}
// One rank's rows of the LUT when N ranks / devices split a frame: rows row0, row0 + row_stride, ... (n_rows of them).  out_f == nullptr
// (csky_render_sky_lut_rows_device): stored COMPACT and as RGBA16F only, straight into the buffer that travels to the gathering rank with the
// rank's bands.  out_f != nullptr (csky_multi_render_sky_lut): stored at the texel's own place in the whole LUT (half + float copies) of the
// handle's first device, over xGMI peer access, like the frame's bands.
__global__ __launch_bounds__(256) void sky_lut_rows_kernel(int w, int h, int row0, int row_stride, int n_rows, Sun3 sun, const float4* __restrict__ trans,
                                                          int tw, int th, uint2* __restrict__ out_h, float4* __restrict__ out_f) {
    __shared__ float steps[8][IN_SCATTERING_STEPS][8];
    const int half = threadIdx.x >> 5, sub = threadIdx.x & 31;
    const int t = blockIdx.x * 8 + half;
    const bool live = t < w * n_rows;
    const int px = live ? t % w : 0, py = live ? row0 + (t / w) * row_stride : 0;
    sky_texel(steps[half], sub, live, px, py, w, h, sun, trans, tw, th, [=](int x, int y, uint16_t hx, uint16_t hy, uint16_t hz, uint16_t hw) {
        if (out_f) { out_h[y * w + x] = pack_half4(hx, hy, hz, hw); out_f[y * w + x] = make_float4(h2f(hx), h2f(hy), h2f(hz), h2f(hw)); }
        else out_h[t] = pack_half4(hx, hy, hz, hw);
    });
}

hipError_t launch_transmittance(int w, int h, uint16_t* d_half, float4* d_float, hipStream_t s) {
    transmittance_kernel<<<(w * h + 3) / 4, 256, 0, s>>>(w, h, d_half, d_float);
    return hipGetLastError();
}
hipError_t launch_sky_lut(int w, int h, const float sun[3], const float4* d_trans, int tw, int th, uint16_t* d_half, float4* d_float,
                          hipStream_t s) {
    Sun3 sv; sv.v[0] = sun[0]; sv.v[1] = sun[1]; sv.v[2] = sun[2];
    sky_lut_kernel<<<(w * h + 7) / 8, 256, 0, s>>>(w, h, sv, d_trans, tw, th, d_half, d_float);
    return hipGetLastError();
}
hipError_t launch_sky_lut_rows(int w, int h, int row0, int row_stride, const float sun[3], const float4* d_trans, int tw, int th, uint2* d_rows, float4* d_whole_f,
                               hipStream_t s) {
    Sun3 sv; sv.v[0] = sun[0]; sv.v[1] = sun[1]; sv.v[2] = sun[2];
    const int n_rows = row0 < h ? (h - row0 + row_stride - 1) / row_stride : 0;
    if (n_rows) sky_lut_rows_kernel<<<(w * n_rows + 7) / 8, 256, 0, s>>>(w, h, row0, row_stride, n_rows, sv, d_trans, tw, th, d_rows, d_whole_f);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ band interleave (gathering rank, N > 1)
// The gather leaves rank-major compact bands ([member][local band][rows]); the frame wants band k = member k % n, local band k / n.  A plain
// strided copy, deliberately on FEW workgroups: it is HBM-bound (32 MiB per 2048x1024 frame) beside marches that are not, so 48 workgroups
// streaming 16-byte chunks take it off the critical path instead of sweeping the whole chip for 15 us per frame (torch's permute + copy).
__global__ __launch_bounds__(256) void interleave_bands_kernel(const uint4* __restrict__ src, size_t member_stride16, int members, uint32_t band16, uint32_t total_bands,
                                                              uint4* __restrict__ dst) {
    const size_t total = (size_t)band16 * total_bands;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t k = (uint32_t)(i / band16), c = (uint32_t)(i - (size_t)k * band16);
        dst[i] = src[(size_t)(k % members) * member_stride16 + (size_t)(k / members) * band16 + c];
    }
}
hipError_t launch_interleave_bands(const void* d_gathered, size_t member_stride_bytes, int members, size_t band_bytes, int total_bands, void* d_frame, hipStream_t s) {
    const size_t chunks = band_bytes / 16 * (size_t)total_bands;
    if (!chunks) return hipSuccess;
    const unsigned grid = (unsigned)(chunks / 256 + 1 < 48 ? chunks / 256 + 1 : 48);
    interleave_bands_kernel<<<grid, 256, 0, s>>>(reinterpret_cast<const uint4*>(d_gathered), member_stride_bytes / 16, members, (uint32_t)(band_bytes / 16), (uint32_t)total_bands,
                                                 reinterpret_cast<uint4*>(d_frame));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ shape-noise bake
// The stand-in 128^3 RGBA shape volume, one voxel per lane (bit-identical to the host generator: noise_core.h).
__global__ __launch_bounds__(256) void shape_noise_kernel(uint32_t seed, int n, ShapeNoiseParams P, uint32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * n * n) return;
    const int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((size_t)n * n));
    uint8_t o[4];
    shape_voxel(seed, n, x, y, z, P, o);
    out[i] = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
}
hipError_t launch_shape_noise(uint32_t seed, int n, const ShapeNoiseParams& P, uint32_t* d_out, hipStream_t s) {
    const size_t total = (size_t)n * n * n;
    shape_noise_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(seed, n, P, d_out);
    return hipGetLastError();
}

// the 32^3 RGB detail volume (noise_core.h::detail_voxel), one voxel per lane, 3 bytes each
__global__ __launch_bounds__(256) void detail_noise_kernel(uint32_t seed, int n, uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * n * n) return;
    const int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((size_t)n * n));
    uint8_t o[3];
    detail_voxel(seed, n, x, y, z, o);
    out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
}
hipError_t launch_detail_noise(uint32_t seed, int n, uint8_t* d_out, hipStream_t s) {
    const size_t total = (size_t)n * n * n;
    detail_noise_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(seed, n, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ mip chains + texture bake on the device
// csky_set_noise uploads the three 8-bit level-0 textures (9.2 MB) and does everything else here: 2x2x2 box mips (Godot's
// mipmaps/generate=true), then one lane per texel of each device layout (bake_core.h: the same per-texel code as the host bake of
// tests/hostsim, byte-identical).  Replaces ~1.5 s of host loops + 78 MB of pageable uploads per csky_set_noise by < 1 ms of kernels.
__global__ __launch_bounds__(256) void mip_level_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int nd, int ch) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)nd * nd * nd * ch;
    if (i >= total) return;
    const int c = (int)(i % ch); const size_t t = i / ch;
    const int x = (int)(t % nd), y = (int)((t / nd) % nd), z = (int)(t / ((size_t)nd * nd));
    dst[i] = mip_texel(src, nd * 2, ch, x, y, z, c);
}
hipError_t launch_mip_chain(uint8_t* d_chain, int n, int ch, int levels, hipStream_t s) {
    for (int l = 1; l < levels; l++) {
        const int nd = n >> l;
        const size_t total = (size_t)nd * nd * nd * ch;
        mip_level_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(d_chain + chain_offset(n, l - 1, ch), d_chain + chain_offset(n, l, ch), nd, ch);
    }
    return hipGetLastError();
}
__device__ __forceinline__ void bake_tally(unsigned bad, unsigned long long* __restrict__ inexact) {
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(inexact, (unsigned long long)bad);
}
template <int N, int LEVELS> __device__ __forceinline__ bool level_of(size_t i, int& l, int& n, size_t& local) {
    size_t base = 0;
    for (l = 0; l < LEVELS; l++) { n = N >> l; const size_t cnt = (size_t)n * n * n; if (i < base + cnt) { local = i - base; return true; } base += cnt; }
    return false;
}
__global__ __launch_bounds__(256) void bake_shape_kernel(const uint8_t* __restrict__ chain, ShapeTexel* __restrict__ out, unsigned long long* __restrict__ inexact) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    int l, n; size_t local; unsigned bad = 0;
    if (level_of<SHAPE_N, SHAPE_LEVELS>(i, l, n, local)) {
        const int x = (int)(local % n), y = (int)((local / n) % n), z = (int)(local / ((size_t)n * n));
        out[(i - local) + shape_cell_index(n, x, y, z)] = bake_shape_texel(chain + chain_offset(SHAPE_N, l, 4), n, x, y, z, bad);
    }
    bake_tally(bad, inexact);
}
__global__ __launch_bounds__(256) void bake_detail_kernel(const uint8_t* __restrict__ chain, uint4* __restrict__ out, uint16_t* __restrict__ out_h, unsigned long long* __restrict__ inexact) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    int l, n; size_t local; unsigned bad = 0;
    if (level_of<DETAIL_N, DETAIL_LEVELS>(i, l, n, local)) {
        const uint8_t* src = chain + chain_offset(DETAIL_N, l, 3);
        const int x = (int)(local % n), y = (int)((local / n) % n), z = (int)(local / ((size_t)n * n));
        out[i] = bake_detail_texel(src, n, x, y, z, bad);
        out_h[i] = f2h((float)detail_numerator(src, n, x, y, z));                 // unpacked fp16 chain: source of the "lds" variant's LDS copy
    }
    bake_tally(bad, inexact);
}
// also the channel ranges of the map (exact height-window reject, bake.h::height_window): range[0] = min R, [1] = max R, [2] = max B
__global__ __launch_bounds__(256) void bake_weather_kernel(const uint8_t* __restrict__ rgb, uint4* __restrict__ out, unsigned long long* __restrict__ inexact, int* __restrict__ range) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned bad = 0;
    if (i < WEATHER_N * WEATHER_N) {
        out[i] = bake_weather_texel(rgb, i % WEATHER_N, i / WEATHER_N, bad);
        int r = rgb[3 * i], b = rgb[3 * i + 2], rmin = r, rmax = r, bmax = b;
        for (int off = 32; off > 0; off >>= 1) { rmin = min(rmin, __shfl_down(rmin, off)); rmax = max(rmax, __shfl_down(rmax, off)); bmax = max(bmax, __shfl_down(bmax, off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&range[0], rmin); atomicMax(&range[1], rmax); atomicMax(&range[2], bmax); }
    }
    bake_tally(bad, inexact);
}
hipError_t launch_bake(const uint8_t* d_large_chain, const uint8_t* d_small_chain, const uint8_t* d_weather, ShapeTexel* d_shape, uint4* d_detail, uint16_t* d_detail_h,
                       uint4* d_weather_out, unsigned long long* d_inexact, int* d_range, hipStream_t s) {
    size_t shape_total = 0, detail_total = 0;
    for (int l = 0; l < SHAPE_LEVELS; l++) { const size_t n = SHAPE_N >> l; shape_total += n * n * n; }
    for (int l = 0; l < DETAIL_LEVELS; l++) { const size_t n = DETAIL_N >> l; detail_total += n * n * n; }
    bake_shape_kernel<<<(unsigned)((shape_total + 255) / 256), 256, 0, s>>>(d_large_chain, d_shape, d_inexact);
    bake_detail_kernel<<<(unsigned)((detail_total + 255) / 256), 256, 0, s>>>(d_small_chain, d_detail, d_detail_h, d_inexact);
    bake_weather_kernel<<<(WEATHER_N * WEATHER_N + 255) / 256, 256, 0, s>>>(d_weather, d_weather_out, d_inexact, d_range);
    return hipGetLastError();
}

// exact cells (bake_core.h: fp32 coefficients), built only for textures with coefficients fp16 cannot hold (or on request)
__global__ __launch_bounds__(256) void bake_shape32_kernel(const uint8_t* __restrict__ chain, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    int l, n; size_t local;
    if (level_of<SHAPE_N, SHAPE_LEVELS>(i, l, n, local)) {
        const int x = (int)(local % n), y = (int)((local / n) % n), z = (int)(local / ((size_t)n * n));
        float4 c[4];
        bake_shape_texel32(chain + chain_offset(SHAPE_N, l, 4), n, x, y, z, c);
        float4* o = out + 4 * ((i - local) + shape_cell_index(n, x, y, z));
        o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3];
    }
}
__global__ __launch_bounds__(256) void bake_detail32_kernel(const uint8_t* __restrict__ chain, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    int l, n; size_t local;
    if (level_of<DETAIL_N, DETAIL_LEVELS>(i, l, n, local)) {
        const int x = (int)(local % n), y = (int)((local / n) % n), z = (int)(local / ((size_t)n * n));
        float4 c[2];
        bake_detail_texel32(chain + chain_offset(DETAIL_N, l, 3), n, x, y, z, c);
        out[2 * i] = c[0]; out[2 * i + 1] = c[1];
    }
}
__global__ __launch_bounds__(256) void bake_weather32_kernel(const uint8_t* __restrict__ rgb, float4* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < WEATHER_N * WEATHER_N) {
        float4 c[2];
        bake_weather_texel32(rgb, i % WEATHER_N, i / WEATHER_N, c);
        out[2 * i] = c[0]; out[2 * i + 1] = c[1];
    }
}
hipError_t launch_bake32(const uint8_t* d_large_chain, const uint8_t* d_small_chain, const uint8_t* d_weather, float4* d_shape32, float4* d_detail32, float4* d_weather32, hipStream_t s) {
    size_t shape_total = 0, detail_total = 0;
    for (int l = 0; l < SHAPE_LEVELS; l++) { const size_t n = SHAPE_N >> l; shape_total += n * n * n; }
    for (int l = 0; l < DETAIL_LEVELS; l++) { const size_t n = DETAIL_N >> l; detail_total += n * n * n; }
    bake_shape32_kernel<<<(unsigned)((shape_total + 255) / 256), 256, 0, s>>>(d_large_chain, d_shape32);
    bake_detail32_kernel<<<(unsigned)((detail_total + 255) / 256), 256, 0, s>>>(d_small_chain, d_detail32);
    bake_weather32_kernel<<<(WEATHER_N * WEATHER_N + 255) / 256, 256, 0, s>>>(d_weather, d_weather32);
    return hipGetLastError();
}

// test hook (csky_test_sqrt_shell): cloud_core.h::sqrt_shell over an array, for the exhaustive check against the host's sqrtf
__global__ __launch_bounds__(256) void sqrt_shell_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = sqrt_shell(in[i]);
}
hipError_t launch_sqrt_shell(const float* d_in, float* d_out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    sqrt_shell_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(d_in, d_out, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ compositor
// clouds.gdshader sky() on an equirectangular panorama, one pixel per lane
__global__ __launch_bounds__(256) void composite_kernel(CompositeArgs A, uint2* __restrict__ out) {
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), j = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (i >= A.out_w || j >= A.out_h) return;
    const C3 c = composite_pixel(A, i, j);
    out[(size_t)j * A.out_w + i] = make_uint2((uint32_t)f2h(c.x) | ((uint32_t)f2h(c.y) << 16), (uint32_t)f2h(c.z) | ((uint32_t)f2h(1.0f) << 16));
}
hipError_t launch_composite(const CompositeArgs& a, uint2* d_out, hipStream_t s) {
    composite_kernel<<<dim3((a.out_w + 31) / 32, (a.out_h + 7) / 8), 256, 0, s>>>(a, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ clouds
__global__ __launch_bounds__(64) void frame_setup_kernel(CloudParams p, const float4* __restrict__ sky, int sw, int sh, int primary_steps,
                                                         int light_steps, float early_eps, float hf_lo, float hf_hi, int ct_mode, FrameConsts* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        FrameConsts fc;
        frame_setup(p, sky, sw, sh, primary_steps, light_steps, early_eps, hf_lo, hf_hi, fc);
        fc.ct_mode = ct_mode;
        *out = fc;
    }
}
hipError_t launch_frame_setup(const CloudParams& p, const float4* d_sky, int sw, int sh, int primary_steps, int light_steps, float early_eps,
                              float hf_lo, float hf_hi, int ct_mode, FrameConsts* d_fc, hipStream_t s) {
    frame_setup_kernel<<<1, 64, 0, s>>>(p, d_sky, sw, sh, primary_steps, light_steps, early_eps, hf_lo, hf_hi, ct_mode, d_fc);
    return hipGetLastError();
}
// The same for a context that holds no sky LUT of its own (one rank of an N-way frame split renders only its rows of it, straight into the
// gather buffer): the <= 12 texels the three taps of clouds.glsl:163-167 filter are rendered here first, one per half wavefront with the
// per-texel code of sky_lut_kernel (fp16-rounded like the stored LUT), parked in LDS, and lane 0 runs the set-up on them.  The cell
// arithmetic is sky_lut_cell's in both places, so every texel the set-up asks for is one rendered here: the constants are bit-identical to
// those filtered from a whole LUT.
__global__ __launch_bounds__(384) void frame_setup_taps_kernel(CloudParams p, Sun3 sun, const float4* __restrict__ trans, int tw, int th, int sw, int sh,
                                                              int primary_steps, int light_steps, float early_eps, float hf_lo, float hf_hi, int ct_mode,
                                                              FrameConsts* __restrict__ out) {
    __shared__ float steps[12][IN_SCATTERING_STEPS][8];
    __shared__ float4 texel[12];
    const int k = threadIdx.x >> 5, sub = threadIdx.x & 31;      // texel k: corner k % 4 of tap k / 4
    float sx, sy, ax, ay; int x0, x1, y0, y1;
    frame_setup_tap_uv(p.LIGHT_DIRECTION, k >> 2, sx, sy);
    sky_lut_cell(sw, sh, sx, sy, x0, x1, y0, y1, ax, ay);
    sky_texel(steps[k], sub, true, (k & 1) ? x1 : x0, (k & 2) ? y1 : y0, sw, sh, sun, trans, tw, th, [&](int, int, uint16_t hx, uint16_t hy, uint16_t hz, uint16_t hw) {
        texel[k] = make_float4(h2f(hx), h2f(hy), h2f(hz), h2f(hw));
    });
    __syncthreads();
    if (threadIdx.x == 0) {
        FrameConsts fc;
        frame_setup_f(p, [&](int tap, int corner, int, int) { return texel[tap * 4 + corner]; }, sw, sh, primary_steps, light_steps, early_eps, hf_lo, hf_hi, fc);
        fc.ct_mode = ct_mode;
        *out = fc;
    }
}
hipError_t launch_frame_setup_taps(const CloudParams& p, const float sun[3], const float4* d_trans, int tw, int th, int sw, int sh, int primary_steps,
                                   int light_steps, float early_eps, float hf_lo, float hf_hi, int ct_mode, FrameConsts* d_fc, hipStream_t s) {
    Sun3 sv; sv.v[0] = sun[0]; sv.v[1] = sun[1]; sv.v[2] = sun[2];
    frame_setup_taps_kernel<<<1, 384, 0, s>>>(p, sv, d_trans, tw, th, sw, sh, primary_steps, light_steps, early_eps, hf_lo, hf_hi, ct_mode, d_fc);
    return hipGetLastError();
}

// ---- wave-cooperative march (variant "queue") --------------------------------------------------------------
// The lock-step march (cloud_core.h march()) runs the (light_steps+1)-sample light march on all 64 lanes whenever
// ANY lane of the wavefront is inside a cloud.  Measured on the headline frame: a lane is in cloud on 15 % of its
// steps, a wavefront on ~70 % of them, so ~4/5 of the light-march VALU work is masked off.  Here the two loops
// are decoupled through a per-wavefront event queue in LDS:
//   A. all lanes march their primary samples in lock step (uniform loop, clouds.glsl:172-178); every in-cloud
//      sample (t > 0, clouds.glsl:184) is appended to the queue (position, t, height fraction) at
//      slot = running count + mbcnt(ballot);
//   B. when the queue may overflow (or the march ends) the n queued samples need n*(light_steps+1) independent
//      density evaluations (clouds.glsl:186-199).  They are flattened as e = j*n + k and dealt out 64 per round,
//      so every round runs with all lanes busy no matter which rays were in cloud;
//   C. the queue is replayed step by step: the owning lane sums its light samples in the reference's order and
//      composites the sample front to back (clouds.glsl:202-210).
// No __syncthreads: wavefronts are independent; LDS operations of one wavefront complete in issue order and
// wavefront-scope fences keep the compiler from reordering them.  Per-ray arithmetic and its order are unchanged.
constexpr int QCAP = 96;                                    // events per wavefront queue (a flush is forced above QCAP-64)
constexpr int QSTEPS = 48;                                  // steps-with-events per chunk (a flush is forced when full)
constexpr int Q_FLOATS = 5 * QCAP + 7 * QCAP + 3 * QSTEPS;  // pos(3) t hf | lt[7] | per-step mask lo/hi + base  = 5.1 KB per wavefront

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ MarchOut march_queue(const TexSet& T, const FrameConsts& fc, Ray ray, float* __restrict__ q, int step_begin, int step_end) {
    float* __restrict__ ev_px = q;
    float* __restrict__ ev_py = q + QCAP;
    float* __restrict__ ev_pz = q + 2 * QCAP;
    float* __restrict__ ev_t = q + 3 * QCAP;
    float* __restrict__ ev_hf = q + 4 * QCAP;
    float* __restrict__ ev_lt = q + 5 * QCAP;                               // [7][QCAP]
    unsigned* __restrict__ st_lo = reinterpret_cast<unsigned*>(q + 12 * QCAP);   // [QSTEPS]
    unsigned* __restrict__ st_hi = st_lo + QSTEPS;
    unsigned* __restrict__ st_base = st_hi + QSTEPS;

    MarchOut o; o.r = o.g = o.b = o.a = 0.0f; o.t = 1.0f; o.incloud = 0;
    const int lane = threadIdx.x & 63;
    const int ls = fc.light_steps, nl = ls + 1;
    float phase = 0.0f;
    if (ray.above) {
        const float ct = fc.ldir[0] * ray.dx + fc.ldir[1] * ray.dy + fc.ldir[2] * ray.dz;                       // clouds.glsl:158
        phase = fmaxf(fmaxf(henyey_greenstein(ct, 0.6f), henyey_greenstein(ct, fc.hg_g2)), henyey_greenstein(ct, -0.2f));  // :160
    }
    float Tr = 1.0f, alpha = 0.0f, Lr = 0.0f, Lg = 0.0f, Lb = 0.0f;
    float px = ray.px, py = ray.py, pz = ray.pz;
    const float nd = -fc.density;
    bool live = ray.above;
    int count = 0, cs = 0;                                    // queued events / steps-with-events in the current chunk (uniform)
    if (!__any(live)) return o;
    // a ray segment starts where the sequential march would be after step_begin steps: replay the fp32 additions
    // (clouds.glsl:173) so every sample position is bit-identical to the unsegmented march
    for (int i = 0; i < step_begin; i++) advance(px, py, pz, ray.sx, ray.sy, ray.sz);
    for (int i = step_begin; i < step_end; i++) {
        // ---- A: one primary sample per lane
        float t = 0.0f, hf = 0.0f;
        if (live) {
            advance(px, py, pz, ray.sx, ray.sy, ray.sz);                                                       // :173
            hf = height_fraction(length3_shell(px, py, pz));                                                   // :175
            t = sample_density(T, fc, px, py, pz, hf, fc.wpos_x, fc.wpos_y, 0, 0);                             // :174, :177
        }
        const bool have = t > 0.0f;                                                                            // :184
        const unsigned long long m = __ballot(have);
        if (m != 0ull) {
            const int slot = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (have) { ev_px[slot] = px; ev_py[slot] = py; ev_pz[slot] = pz; ev_t[slot] = t; ev_hf[slot] = hf; }
            if (lane == 0) { st_lo[cs] = (unsigned)m; st_hi[cs] = (unsigned)(m >> 32); st_base[cs] = (unsigned)count; }
            count += __popcll(m);
            cs++;
        }
        if (count <= QCAP - 64 && cs < QSTEPS && i + 1 < step_end) continue;
        if (count == 0) continue;
        // ---- B: count*(ls+1) light-march evaluations, 64 per round, all lanes busy
        wave_lds_fence();
        const int total = count * nl;
        const float rn = 1.0f / (float)count;
        for (int e0 = 0; e0 < total; e0 += 64) {
            const int e = e0 + lane;
            if (e < total) {
                const int j = (int)(((float)e + 0.5f) * rn);          // e = j*count + k (exact: e < 672, count <= 96)
                const int k = e - j * count;
                float lx = ev_px[k], ly = ev_py[k], lz = ev_pz[k];
                const bool distant = (j == ls);
                // cone sample j: lp = p + sum_{i<=j} (ldir + RANDOM_VECTORS[i]*i)*lss, added one by one in fp32 like :187.
                // e grows with the lane, so j is non-decreasing across the wavefront: the additions up to the first lane's j
                // are wave-uniform (plain adds under a scalar branch), only the few beyond it need per-lane predication.
                const int j_lo = __builtin_amdgcn_readfirstlane(j);
                if (distant) {
                    advance(lx, ly, lz, fc.ldist[0], fc.ldist[1], fc.ldist[2]);                                // :195
                } else {
#pragma unroll
                    for (int jj = 0; jj < 6; jj++) {
                        if (jj <= j_lo) advance(lx, ly, lz, fc.linc[jj][0], fc.linc[jj][1], fc.linc[jj][2]);
                        else if (jj <= j) advance(lx, ly, lz, fc.linc[jj][0], fc.linc[jj][1], fc.linc[jj][2]);
                    }
                }
                const float lhf = height_fraction(length3_shell(lx, ly, lz));                                  // :188 / :196
                const int lod_s = distant ? 3 : (j > 2 ? j - 2 : 0), lod_d = distant ? 5 : j;                  // textureLod(.., mip-2) / (.., mip)
                float d = sample_density(T, fc, lx, ly, lz, lhf, distant ? 0.0f : fc.wpos_x, distant ? 0.0f : fc.wpos_y, lod_s, lod_d);  // :189-190 / :197-198
                if (distant) d = fast_pow(d, (1.0f - lhf) * 0.8f + 0.5f);                                      // :198 second pow
                ev_lt[j * QCAP + k] = d;
            }
        }
        wave_lds_fence();
        // ---- C: replay the chunk in step order; owners composite (:191,:199 sums in the reference's order, :202-210)
        for (int s = 0; s < cs; s++) {
            const unsigned lo = st_lo[s], hi = st_hi[s];
            const bool mine = lane < 32 ? ((lo >> lane) & 1u) : ((hi >> (lane - 32)) & 1u);
            if (mine) {
                const int slot = (int)st_base[s] + (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                float cd = 0.0f;
                for (int j = 0; j < nl; j++) cd += ev_lt[j * QCAP + slot];
                const float et = ev_t[slot], ehf = ev_hf[slot];
                const float dt = fast_exp(nd * et * ray.ss);                                                   // :178
                shade_sample(fc, phase, et, ehf, dt, cd, Tr, alpha, Lr, Lg, Lb);
                o.incloud++;
            }
        }
        wave_lds_fence();
        count = 0; cs = 0;
        if (fc.early_eps > 0.0f) {                                     // build-side early-out (off by default, bounded error)
            if (Tr < fc.early_eps) live = false;
            if (!__any(live)) break;
        }
    }
    o.r = Lr; o.g = Lg; o.b = Lb; o.a = sat(alpha); o.t = Tr;                                                  // :213-214
    return o;
}

// ---- compacted light march (variant "compact") ------------------------------------------------------------------
// Same decoupling as march_queue, but a flush always takes EXACTLY 64 queued samples, one per lane (k = lane): every lane
// then runs the reference's whole light march (clouds.glsl:186-199) for its sample in registers.  The loop index j is
// wave-uniform, so the LOD, the mip extent and offsets, the cone increment and the distant-sample special case are scalar
// (SALU / immediates) instead of per-lane VALU selects, the cone position is carried in registers (3 adds per sample instead
// of re-adding j+1 increments under predication), and the per-sample densities are summed in place (no lt[7][QCAP] array).
// Samples beyond the 64th stay queued (moved to the front) together with the part of the last step that owns them; only
// the final flush of a ray segment runs with idle lanes.
constexpr int CQ_CAP = 128;                                  // a flush is taken as soon as 64 samples are queued: count <= 63 + 64
constexpr int CQ_STEPS = 66;                                 // steps-with-events between flushes: <= 1 carried + 64 (>= 1 event each)
constexpr int CQ_FLOATS = 5 * CQ_CAP + CQ_CAP / 4 + 3 * CQ_STEPS + 2 + 128 + 320;   // pos(3) t hf (after the light march: D(3) q dt) | owner lane (bytes) | per-step mask lo/hi + base | in-cloud tally | per-ray ss, phase | per-ray T, alpha, L between flushes = 5.2 KB per wavefront

#ifndef CSKY_EAGER_LIGHT
#define CSKY_EAGER_LIGHT 1
#endif
#if CSKY_EAGER_LIGHT
#define CSKY_LIGHT_SAMPLE sample_density_eager<true>
#else
#define CSKY_LIGHT_SAMPLE sample_density
#endif
#ifndef CSKY_EAGER_PRIMARY
#define CSKY_EAGER_PRIMARY 0   // eager fetches in the PRIMARY march measured slower (1: weather + shape together +1 %, 2: all three +4.5 %): only
                               // 56 % / 22 % of its samples need the shape / detail cell, the extra gathers cost more than the latency they hide
#endif
#if CSKY_EAGER_PRIMARY == 1
#define CSKY_PRIMARY_SAMPLE sample_density_eager<false>
#elif CSKY_EAGER_PRIMARY == 2
#define CSKY_PRIMARY_SAMPLE sample_density_eager<true>
#else
#define CSKY_PRIMARY_SAMPLE sample_density
#endif
// Step B of the compact march for ONE queued in-cloud sample: the reference's light march (clouds.glsl:186-199) from its position and the
// state-independent half of its shading (:178, :202-209).  In: position, density t, height fraction, the owner ray's step length and phase
// value.  Out: D.rgb, 1 / max(1e-7, t), dt (shade_terms).  A function of those seven floats alone (round 4's packet exchange ran it on other
// CUs' wavefronts and got byte-identical frames: see the record further down).
// `late(et, ehf, ess, eph)` delivers the four inputs the light march itself does not need AFTER it (the owner reads them from its LDS queue
// then: four registers fewer across the march).
template <class TS, class Late>
__device__ __forceinline__ void light_march_terms(const TS& T, const FrameConsts& fc, const int ls, const float nd, float ex, float ey, float ez, Late&& late,
                                                  float& Dr, float& Dg, float& Db, float& rq, float& dt) {
    float lx = ex, ly = ey, lz = ez, cd = 0.0f;
#pragma unroll 1                                                 // scalar j: one loop body (unrolling measured no faster in rounds 2 and 5, 6x the code: profiles/r05/kernel_experiments.txt)
    for (int j = 0; j < 6; j++) {                                                                      // :186 (light_steps <= 6)
        if (j >= ls) break;
        advance(lx, ly, lz, fc.linc[j][0], fc.linc[j][1], fc.linc[j][2]);                              // :187
        const float lhf = height_fraction(length3_shell(lx, ly, lz));                                  // :188
        cd += CSKY_LIGHT_SAMPLE(T, fc, lx, ly, lz, lhf, fc.wpos_x, fc.wpos_y, j > 2 ? j - 2 : 0, j);   // :189-191
    }
    {   // distant sample, :195-199
        lx = ex; ly = ey; lz = ez;
        advance(lx, ly, lz, fc.ldist[0], fc.ldist[1], fc.ldist[2]);
        const float lhf = height_fraction(length3_shell(lx, ly, lz));
        const float ld = CSKY_LIGHT_SAMPLE(T, fc, lx, ly, lz, lhf, 0.0f, 0.0f, 3, 5);                  // :197 has no weather_pos
        cd += fast_pow(ld, (1.0f - lhf) * 0.8f + 0.5f);                                                // :198 (second pow)
    }
    float et, ehf, ess, eph;
    late(et, ehf, ess, eph);
    dt = fast_exp(nd * et * ess);                                                                      // :178
    shade_terms(fc, eph, et, ehf, dt, cd, Dr, Dg, Db, rq);                                             // :202-209
}

template <class TS>
__device__ __forceinline__ MarchOut march_compact(const TS& T, const FrameConsts& fc, Ray ray, float* __restrict__ q, int step_begin, int step_end) {
    float* __restrict__ ev_px = q;
    float* __restrict__ ev_py = q + CQ_CAP;
    float* __restrict__ ev_pz = q + 2 * CQ_CAP;
    float* __restrict__ ev_t = q + 3 * CQ_CAP;
    float* __restrict__ ev_hf = q + 4 * CQ_CAP;
    unsigned char* __restrict__ ev_owner = reinterpret_cast<unsigned char*>(q + 5 * CQ_CAP);   // the sample's owner lane: its step length and phase value are read from ray_ss / ray_ph
    unsigned* __restrict__ st_lo = reinterpret_cast<unsigned*>(q + 5 * CQ_CAP + CQ_CAP / 4);  // [CQ_STEPS]
    unsigned* __restrict__ st_hi = st_lo + CQ_STEPS;
    unsigned* __restrict__ st_base = st_hi + CQ_STEPS;

    MarchOut o; o.r = o.g = o.b = o.a = 0.0f; o.t = 1.0f; o.incloud = 0;
    const int lane = threadIdx.x & 63;
    const int ls = fc.light_steps;
    float phase = 0.0f;
    if (ray.above) {
        const float ct = fc.ldir[0] * ray.dx + fc.ldir[1] * ray.dy + fc.ldir[2] * ray.dz;                       // clouds.glsl:158
        phase = fmaxf(fmaxf(henyey_greenstein(ct, 0.6f), henyey_greenstein(ct, fc.hg_g2)), henyey_greenstein(ct, -0.2f));  // :160
    }
    float px = ray.px, py = ray.py, pz = ray.pz;
    const float nd = -fc.density;
    bool live = ray.above;
    int count = 0, cs = 0;                                    // queued samples / steps owning them (uniform)
    unsigned* __restrict__ tally = st_base + CQ_STEPS;          // in-cloud samples composited by this wavefront, kept in LDS (round 4: neither a per-lane accumulator register
    if (lane == 0) tally[0] = 0u;                             // in the march loops nor a scalar one in the SGPR-starved persistent form; one ds_add per flush)
    // the two per-ray constants a queued sample carries (step length, phase value) live in LDS, not in registers held across the whole march: the
    // persistent form had spilled `phase` to scratch and re-loaded it, behind a vmcnt(0), at every step with an in-cloud sample (round 4 census)
    float* __restrict__ ray_ss = reinterpret_cast<float*>(tally + 2);
    float* __restrict__ ray_ph = ray_ss + 64;
    ray_ss[lane] = ray.ss; ray_ph[lane] = phase;
    // The running state of the ray (T, alpha, L: clouds.glsl:151-153) is touched only while a flush is composited, ~20 times per tile: between
    // flushes it rests in LDS, not in five registers the allocator had to copy between register sets around every light march (20 M v_mov per
    // C3 frame, round-4 census) and hold across both march loops.
    float* __restrict__ acc = ray_ph + 64;                      // [5][64]
    acc[lane] = 1.0f; acc[64 + lane] = 0.0f; acc[128 + lane] = 0.0f; acc[192 + lane] = 0.0f; acc[256 + lane] = 0.0f;
    if (__builtin_amdgcn_ballot_w64(live) == 0ull) return o;   // (the builtin takes the compare's own lane mask: HIP's __any / __ballot wrappers cost a v_cndmask + v_cmp_ne each)
    for (int i = 0; i < step_begin; i++) advance(px, py, pz, ray.sx, ray.sy, ray.sz);   // segment start: replay the fp32 additions (:173)
    int end = step_end;                                       // shrinks when the whole wavefront has left the height window
    for (int i = step_begin;;) {
        // ---- A: one primary sample per lane (none once the segment is exhausted and only carried samples remain)
        if (i < end) {
            float t, hf;                                     // (meaningful in live lanes only: every use below is guarded by `live` or by `have`)
            bool have = false, below_top = false;
            if (live) {
                advance(px, py, pz, ray.sx, ray.sy, ray.sz);                                                   // :173
                hf = height_fraction(length3_shell(px, py, pz));                                               // :175
                t = CSKY_PRIMARY_SAMPLE(T, fc, px, py, pz, hf, fc.wpos_x, fc.wpos_y, 0, 0);                    // :174, :177
                have = t > 0.0f;                                                                               // :184
                below_top = !(hf >= fc.hf_hi);
            }
            // Exact early end of the march: a ray starts on the inner shell and |p| only grows along it (>= 14 m per step at 128 steps even
            // for a grazing ray, 1.7 m at 1024 steps, against 0.5 m of fp32 noise; every ray of the C3 / C5 frames is walked by
            // tests/test_hostsim_core.py), so once EVERY live ray of the wavefront is above the height window
            // (density() == 0 there, cloud_core.h) all remaining samples are 0 too.  Checked every 4th step: one compare + ballot.
            // 21 % of the wave-steps of the headline view lie above the window (tools/stage_trace).
            if ((i & 3) == 3 && __builtin_amdgcn_ballot_w64(below_top) == 0ull) end = i + 1;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(have);
            if (m != 0ull) {
                const int slot = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (have) { ev_px[slot] = px; ev_py[slot] = py; ev_pz[slot] = pz; ev_t[slot] = t; ev_hf[slot] = hf; ev_owner[slot] = (unsigned char)lane; }
                if (lane == 0) { st_lo[cs] = (unsigned)m; st_hi[cs] = (unsigned)(m >> 32); st_base[cs] = (unsigned)count; }
                count += __popcll(m);
                cs++;
            }
            i++;
        }
        const bool last = i >= end;
        if (count == 0) { if (last) break; continue; }
        if (count < 64 && !last) continue;
        // ---- B: the light march of the first n = min(count, 64) queued samples, one per lane, and the state-independent half of their
        //         shading (clouds.glsl:178, :202-209) while all lanes are busy: the replay below runs with ~1/6 of them (round 3: the census
        //         put 15 % of the kernel's issue time in the replay loop, 43 VALU instructions per step incl. 4 transcendentals)
        wave_lds_fence();
        const int n = count < 64 ? count : 64;
        unsigned long long carry = 0ull;                     // lanes of the last step whose sample is still queued
        if (lane == 0) tally[0] += (unsigned)n;
        if (lane < n) {
            float Dr, Dg, Db, rq, dt;
            light_march_terms(T, fc, ls, nd, ev_px[lane], ev_py[lane], ev_pz[lane],
                              [&](float& et, float& ehf, float& ess, float& eph) { et = ev_t[lane]; ehf = ev_hf[lane]; const int ow = ev_owner[lane]; ess = ray_ss[ow]; eph = ray_ph[ow]; }, Dr, Dg, Db, rq, dt);
            ev_px[lane] = Dr; ev_py[lane] = Dg; ev_pz[lane] = Db; ev_t[lane] = rq; ev_hf[lane] = dt;           // the sample's slot now holds its terms
        }
        wave_lds_fence();
        // ---- C: replay the steps in order; owners of evaluated samples (slot < n) composite (:207-210)
        float Tr = acc[lane], alpha = acc[64 + lane], Lr = acc[128 + lane], Lg = acc[192 + lane], Lb = acc[256 + lane];
        for (int s = 0; s < cs; s++) {
            // the step's lane mask is wave-uniform: as a scalar pair it IS the execution mask of the owners (inverse ballot: no per-lane bit test)
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)st_lo[s]), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)st_hi[s]);
            const bool mine = __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)hi << 32) | lo);
            const int slot = (int)st_base[s] + (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
            if (mine && slot < n) {
                composite_sample(ev_hf[slot], ev_t[slot], ev_px[slot], ev_py[slot], ev_pz[slot], Tr, alpha, Lr, Lg, Lb);
            }
            if (s == cs - 1) carry = __builtin_amdgcn_ballot_w64(mine && slot >= n);
        }
        acc[lane] = Tr; acc[64 + lane] = alpha; acc[128 + lane] = Lr; acc[192 + lane] = Lg; acc[256 + lane] = Lb;
        wave_lds_fence();
        // ---- keep what was not evaluated: samples n..count-1 move to the front, the last step keeps its unevaluated lanes
        const int rem = count - n;
        if (rem > 0) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, a4 = 0.0f; unsigned char a5 = 0;
            if (lane < rem) { a0 = ev_px[n + lane]; a1 = ev_py[n + lane]; a2 = ev_pz[n + lane]; a3 = ev_t[n + lane]; a4 = ev_hf[n + lane]; a5 = ev_owner[n + lane]; }
            wave_lds_fence();
            if (lane < rem) { ev_px[lane] = a0; ev_py[lane] = a1; ev_pz[lane] = a2; ev_t[lane] = a3; ev_hf[lane] = a4; ev_owner[lane] = a5; }
            if (lane == 0) { st_lo[0] = (unsigned)carry; st_hi[0] = (unsigned)(carry >> 32); st_base[0] = 0u; }
            wave_lds_fence();
            count = rem; cs = 1;
        } else {
            count = 0; cs = 0;
        }
        if (fc.early_eps > 0.0f) {                                     // build-side early-out (off by default, bounded error)
            if (Tr < fc.early_eps) live = false;
            if (__builtin_amdgcn_ballot_w64(live) == 0ull) break;
        }
        if (last && count == 0) break;                               // carried samples get one more (partial) flush
    }
    wave_lds_fence();
    o.r = acc[128 + lane]; o.g = acc[192 + lane]; o.b = acc[256 + lane]; o.a = sat(acc[64 + lane]); o.t = acc[lane];   // :213-214
    o.incloud = lane == 0 ? tally[0] : 0u;
    return o;
}

// ---- instruction-level parallelism for small launches (round 3: built as kernel variant "compact-ilp", measured, removed) ------------
// Two primary steps and two light samples in flight per wavefront (their gathers issued together, the exact rejects as selects so that the
// scheduler interleaves the two dependent chains), 116 VGPRs -> 4 waves per SIMD, which is all one GPU's 1/8 share of a frame has anyway.
// Parity green (tight gate, sample counts equal to march_compact for seven march shapes), but ms per frame at 1/8 share x1 / x2 / x4 frames in
// flight: 0.508 / 0.300 / 0.332 against 0.436 / 0.327 / 0.281 for the policy below, whole frame 2.42 vs 2.08: a lone wavefront issues ONE of
// its own VALU instructions per ~5 cycles whether or not they depend on each other, so independent work inside a wavefront buys only the
// overlapped gathers.  profiles/r03/share_matrix_compact_ilp_ab.txt; the code is in the history (commit "compact-ilp kernel variant").

// ---- interleaved ray segments (small launches) ------------------------------------------------------------------
// One workgroup = ONE 8x8 tile; wavefront w marches the primary samples i = 4m + w of every ray of the tile.  In-cloud
// samples cluster in a few step ranges of a ray, so splitting a ray by step RANGE leaves one wavefront with most of the
// light march; dealing the steps out round-robin gives all four SIMDs of the CU a quarter of the tile's light march
// (the heaviest tile of the headline frame has 4.2x the mean in-cloud samples and is the critical path of a launch that
// holds only a few wavefronts per SIMD: one GPU's share of a frame split 8 ways).
// Front-to-back compositing needs T_i = prod_{k<i} dt_k across ALL wavefronts' samples: per chunk of 32 steps every
// wavefront publishes the step transmittances dt of its samples (1 for samples outside cloud), 16 lanes per wavefront
// scan them into exclusive prefix products, and each wavefront then shades its own samples with L_w += T_i * (...).
// The partial L_w are summed in wavefront order and alpha = 1 - T_end (clouds.glsl:207 is the same product).  Sample
// positions and densities are bit-identical to the sequential march; only the compositing sums are re-associated.
constexpr int IL_CHUNK = 32, IL_OWN = IL_CHUNK / 4, IL_BATCH = 96;
constexpr int IL_WAVE_FLOATS = 5 * IL_OWN * 64 + (IL_OWN * 64) / 2 + 7 * IL_BATCH;     // dense px,py,pz,t,hf | idx (u16) | lt
constexpr int IL_BLOCK_FLOATS = 4 * IL_WAVE_FLOATS + 2 * IL_CHUNK * 64 + 4 * 4 * 64;     // + D, P planes + combine

__device__ __forceinline__ void march_interleaved(const TexSet& T, const FrameConsts& fc, const Ray& ray, float* __restrict__ smem, int wave, int lane,
                                                  float& out_r, float& out_g, float& out_b, float& out_a, unsigned& incloud) {
    float* __restrict__ wq = smem + wave * IL_WAVE_FLOATS;
    float* __restrict__ d_px = wq;                              // [IL_OWN][64]; reused for cd after the light march
    float* __restrict__ d_py = wq + IL_OWN * 64;
    float* __restrict__ d_pz = wq + 2 * IL_OWN * 64;
    float* __restrict__ d_t = wq + 3 * IL_OWN * 64;
    float* __restrict__ d_hf = wq + 4 * IL_OWN * 64;
    unsigned short* __restrict__ ev_idx = reinterpret_cast<unsigned short*>(wq + 5 * IL_OWN * 64);   // [IL_OWN*64]
    float* __restrict__ lt = wq + 5 * IL_OWN * 64 + (IL_OWN * 64) / 2;                                  // [7][IL_BATCH]
    float* __restrict__ D = smem + 4 * IL_WAVE_FLOATS;          // [IL_CHUNK][64] step transmittance of every sample of the chunk
    float* __restrict__ P = D + IL_CHUNK * 64;                  // [IL_CHUNK][64] exclusive prefix products
    float* __restrict__ comb = P + IL_CHUNK * 64;               // [4][4][64]

    const int steps = fc.primary_steps, ls = fc.light_steps, nl = ls + 1;
    const float nd = -fc.density;
    float phase = 0.0f;
    if (ray.above) {
        const float ct = fc.ldir[0] * ray.dx + fc.ldir[1] * ray.dy + fc.ldir[2] * ray.dz;
        phase = fmaxf(fmaxf(henyey_greenstein(ct, 0.6f), henyey_greenstein(ct, fc.hg_g2)), henyey_greenstein(ct, -0.2f));
    }
    float Lr = 0.0f, Lg = 0.0f, Lb = 0.0f;
    float px = ray.px, py = ray.py, pz = ray.pz;
    int replay = wave + 1;                                      // additions of dir*ss needed to reach this wavefront's next sample
    float carry = 1.0f;                                         // scan lanes (lane < 16): T of ray 16*wave + lane after the chunks so far
    incloud = 0;
    const int nchunks = (steps + IL_CHUNK - 1) / IL_CHUNK;
    for (int c = 0; c < nchunks; c++) {
        // ---- A: this wavefront's 8 samples of the chunk
        int count = 0;
        for (int q = 0; q < IL_OWN; q++) {
            const int i = c * IL_CHUNK + 4 * q + wave;
            float t = 0.0f, hf = 0.0f;
            if (ray.above && i < steps) {
                for (int a = 0; a < replay; a++) advance(px, py, pz, ray.sx, ray.sy, ray.sz);     // clouds.glsl:173, one add per step
                hf = height_fraction(length3_shell(px, py, pz));
                t = sample_density(T, fc, px, py, pz, hf, fc.wpos_x, fc.wpos_y, 0, 0);
            }
            replay = 4;
            const bool have = t > 0.0f;
            d_t[q * 64 + lane] = t;
            D[(4 * q + wave) * 64 + lane] = have ? fast_exp(nd * t * ray.ss) : 1.0f;               // clouds.glsl:178
            const unsigned long long m = __ballot(have);
            if (m != 0ull) {
                const int slot = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (have) { ev_idx[slot] = (unsigned short)(q * 64 + lane); d_px[q * 64 + lane] = px; d_py[q * 64 + lane] = py; d_pz[q * 64 + lane] = pz; d_hf[q * 64 + lane] = hf; }
                count += __popcll(m);
            }
        }
        wave_lds_fence();
        // ---- B: light march of the chunk's events, IL_BATCH events at a time, 64 evaluations per round
        for (int b0 = 0; b0 < count; b0 += IL_BATCH) {
            const int nb = (count - b0) < IL_BATCH ? (count - b0) : IL_BATCH;
            const int total = nb * nl;
            const float rn = 1.0f / (float)nb;
            for (int e0 = 0; e0 < total; e0 += 64) {
                const int e = e0 + lane;
                if (e < total) {
                    const int j = (int)(((float)e + 0.5f) * rn);
                    const int k = e - j * nb;
                    const int id = ev_idx[b0 + k];
                    float lx = d_px[id], ly = d_py[id], lz = d_pz[id];
                    const bool distant = (j == ls);
                    const int j_lo = __builtin_amdgcn_readfirstlane(j);
                    if (distant) {
                        advance(lx, ly, lz, fc.ldist[0], fc.ldist[1], fc.ldist[2]);
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 6; jj++) {
                            if (jj <= j_lo) advance(lx, ly, lz, fc.linc[jj][0], fc.linc[jj][1], fc.linc[jj][2]);
                            else if (jj <= j) advance(lx, ly, lz, fc.linc[jj][0], fc.linc[jj][1], fc.linc[jj][2]);
                        }
                    }
                    const float lhf = height_fraction(length3_shell(lx, ly, lz));
                    const int lod_s = distant ? 3 : (j > 2 ? j - 2 : 0), lod_d = distant ? 5 : j;
                    float d = sample_density(T, fc, lx, ly, lz, lhf, distant ? 0.0f : fc.wpos_x, distant ? 0.0f : fc.wpos_y, lod_s, lod_d);
                    if (distant) d = fast_pow(d, (1.0f - lhf) * 0.8f + 0.5f);
                    lt[j * IL_BATCH + k] = d;
                }
            }
            wave_lds_fence();
            for (int k = lane; k < nb; k += 64) {               // cd of each event, summed in the reference's order (:191, :199)
                float cd = 0.0f;
                for (int j = 0; j < nl; j++) cd += lt[j * IL_BATCH + k];
                d_px[ev_idx[b0 + k]] = cd;                      // the position is consumed: its x slot now holds cd
            }
            wave_lds_fence();
        }
        __syncthreads();
        // ---- scan: exclusive prefix products of the chunk's step transmittances, 16 rays per wavefront
        if (lane < 16) {
            const int r = wave * 16 + lane;
            float Tp = carry;
            for (int s = 0; s < IL_CHUNK; s++) { P[s * 64 + r] = Tp; Tp *= D[s * 64 + r]; }
            carry = Tp;
        }
        __syncthreads();
        // ---- C: shade this wavefront's in-cloud samples (clouds.glsl:202-209) with T_i from the scan
        for (int q = 0; q < IL_OWN; q++) {
            const float t = d_t[q * 64 + lane];
            if (t > 0.0f) {
                const float hf = d_hf[q * 64 + lane], cd = d_px[q * 64 + lane];
                const float dt = D[(4 * q + wave) * 64 + lane];
                float Tr = P[(4 * q + wave) * 64 + lane], alpha_unused = 0.0f;
                shade_sample(fc, phase, t, hf, dt, cd, Tr, alpha_unused, Lr, Lg, Lb);
                incloud++;
            }
        }
        __syncthreads();                                        // D/P/dense buffers are rewritten by the next chunk
    }
    // ---- combine: L = sum of the wavefronts' partial sums (wavefront order), alpha = 1 - T_end
    comb[(wave * 4 + 0) * 64 + lane] = Lr; comb[(wave * 4 + 1) * 64 + lane] = Lg; comb[(wave * 4 + 2) * 64 + lane] = Lb;
    if (lane < 16) comb[(0 * 4 + 3) * 64 + wave * 16 + lane] = carry;
    __syncthreads();
    out_r = comb[(0 * 4 + 0) * 64 + lane]; out_g = comb[(0 * 4 + 1) * 64 + lane]; out_b = comb[(0 * 4 + 2) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; w++) { out_r += comb[(w * 4 + 0) * 64 + lane]; out_g += comb[(w * 4 + 1) * 64 + lane]; out_b += comb[(w * 4 + 2) * 64 + lane]; }
    out_a = sat(1.0f - comb[(0 * 4 + 3) * 64 + lane]);
}

template <int DUMMY>
__global__ __launch_bounds__(256) void clouds_kernel_interleaved(TexSet T, const FrameConsts* __restrict__ fcp, RenderGeom G, const uint32_t* __restrict__ order,
                                                                 uint2* __restrict__ out, unsigned long long* __restrict__ stats, uint32_t* __restrict__ wg_cost) {
    extern __shared__ __attribute__((aligned(16))) float il_smem[];
    const int tiles_x = (G.tile_w + 7) >> 3;
    const int local_rows = G.n_bands * G.band_rows;
    const uint32_t logical = order[blockIdx.x];
    if (logical == 0xffffffffu) return;
    const int slab = (int)logical / tiles_x, bx = (int)logical - slab * tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gx = bx * 8 + (lane & 7);
    const int lr = slab * 8 + (lane >> 3);
    const bool valid = gx < G.tile_w && lr < local_rows;
    const int band = lr / G.band_rows, rib = lr - band * G.band_rows;
    const int gy = (G.first_band + band * G.band_stride) * G.band_rows + rib;
    const FrameConsts& fc = *fcp;
    T.detail_lds = nullptr;
    Ray ray = ray_setup(fc, valid ? gx : 0, valid ? gy : 0);
    if (!valid) ray.above = false;
    float r, g, b, a; unsigned ic;
    march_interleaved(T, fc, ray, il_smem, wave, lane, r, g, b, a, ic);
    if (valid && wave == 0) {
        const uint32_t lo = (uint32_t)f2h(r) | ((uint32_t)f2h(g) << 16), hi = (uint32_t)f2h(b) | ((uint32_t)f2h(a) << 16);
        out[(size_t)(G.out_full ? gy : lr) * G.pitch_px + gx] = make_uint2(lo, hi);
    }
    if (stats) {
        unsigned ab = (ray.above && wave == 0) ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) { ic += __shfl_down(ic, off); ab += __shfl_down(ab, off); }
        if (lane == 0) { atomicAdd(&stats[0], (unsigned long long)ic); atomicAdd(&stats[1], (unsigned long long)ab); }
    }
}

// ---- "lds" variant: the detail noise volume staged in LDS (north star) ---------------------------------------------
// A 1024-thread workgroup (16 wavefronts = a 128 x 8 pixel strip) copies the whole 32^3 detail mip chain (37 449 fp16
// numerators, 73 KB) into LDS once and every wavefront runs the queue march with its detail taps served from LDS.  LDS:
// 73 KB + 16 x 5.1 KB of event queues = 155 KB, i.e. ONE workgroup per CU (4 wavefronts per SIMD).  Kept as a measured
// alternative; the default keeps the detail volume oct-packed in L2 (one 16-byte gather per tap).
template <int DUMMY>
__global__ __launch_bounds__(1024) void clouds_kernel_lds(TexSet T, const FrameConsts* __restrict__ fcp, RenderGeom G, const uint32_t* __restrict__ order,
                                                          uint2* __restrict__ out, unsigned long long* __restrict__ stats, uint32_t* __restrict__ wg_cost) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    uint16_t* detail_s = reinterpret_cast<uint16_t*>(lds_all);
    constexpr int DETAIL_FLOATS = (DETAIL_CHAIN_TEXELS + 7) / 8 * 4;            // rounded up to 16 bytes
    float* queues = lds_all + DETAIL_FLOATS;
    {   // stage the detail chain: 16-byte copies, all 1024 threads
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(T.detail_h);
        uint4* dst = reinterpret_cast<uint4*>(lds_all);
        for (int i = threadIdx.x; i < DETAIL_FLOATS / 4; i += 1024) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t logical = order[blockIdx.x];
    if (logical == 0xffffffffu) return;
    const int tiles_x = (G.tile_w + 127) >> 7;
    const int local_rows = G.n_bands * G.band_rows;
    const int slab = (int)logical / tiles_x, bx = (int)logical - slab * tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gx = bx * 128 + wave * 8 + (lane & 7);
    const int lr = slab * 8 + (lane >> 3);
    const bool valid = gx < G.tile_w && lr < local_rows;
    const int band = lr / G.band_rows, rib = lr - band * G.band_rows;
    const int gy = (G.first_band + band * G.band_stride) * G.band_rows + rib;
    const FrameConsts& fc = *fcp;
    Ray ray = ray_setup(fc, valid ? gx : 0, valid ? gy : 0);
    if (!valid) ray.above = false;
    TexSet Tl = T;
    Tl.detail_lds = detail_s;
    const MarchOut o = march_queue(Tl, fc, ray, queues + wave * Q_FLOATS, 0, fc.primary_steps);
    if (valid) {
        const uint32_t lo = (uint32_t)f2h(o.r) | ((uint32_t)f2h(o.g) << 16), hi = (uint32_t)f2h(o.b) | ((uint32_t)f2h(o.a) << 16);
        out[(size_t)(G.out_full ? gy : lr) * G.pitch_px + gx] = make_uint2(lo, hi);
    }
    if (stats || wg_cost) {
        unsigned ic = o.incloud, ab = ray.above ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) { ic += __shfl_down(ic, off); ab += __shfl_down(ab, off); }
        if (lane == 0) {
            if (stats) { atomicAdd(&stats[0], (unsigned long long)ic); atomicAdd(&stats[1], (unsigned long long)ab); }
            if (wg_cost) atomicAdd(&wg_cost[logical], ic + 16u * ab);   // cost model of the feedback schedule: light marches + live primary marches
        }
    }
}

// Pixel <-> lane mapping: a wavefront owns one 8x8-pixel tile (lane = ly*8 + lx), the reference's workgroup footprint
// (clouds.glsl:5): its 64 rays are angularly adjacent, so their texture footprints overlap (L1/TA coalescing) and
// they enter/leave cloud together.  A 256-thread workgroup = 4 wavefronts covers 4/SEG tiles side by side:
//   SEG = 1: 4 tiles (32x8 px), every wavefront marches its rays end to end;
//   SEG = 2/4: the primary march of each ray is cut into SEG segments marched by SEG wavefronts in parallel and
//              combined front to back through LDS (L = L0 + T0*L1 + ..., T = prod T_s, 1-alpha = prod (1-alpha_s)):
//              a wavefront's latency (0.65 ms for 128 steps) is what limits small launches (one GPU's 1/8 frame), and
//              segments divide it by SEG.  Sample positions stay bit-identical; the compositing sums are re-associated.
// Workgroup order: physical workgroup b runs on XCD b % 8 (observed, speed only); `order` (api.cpp::build_schedule)
// maps b to a workgroup footprint.
#ifndef CSKY_COMPACT_WAVES
#define CSKY_COMPACT_WAVES 7   // waves/SIMD asked of the "compact" variant.  With the eager light-march fetches (three gathers of a sample in flight
                               // together) 7 waves x 72 VGPRs beat 8 waves x 64 VGPRs + spills: whole frame 1.83 -> 1.80 ms, 1/4 frame 0.49 -> 0.48
#endif
// One workgroup's footprint (4 tiles / SEG): `logical` = slab * tiles_x + bx, `rec` = its position in the launch order (the persistent form: its cost-feedback slot).
template <int VARIANT, int SEG, class TS = TexSet>
__device__ __forceinline__ void render_block(TS T, const FrameConsts* __restrict__ fcp, const RenderGeom& G, const uint32_t logical, const uint32_t rec,
                                             uint2* __restrict__ out, unsigned long long* __restrict__ stats, uint32_t* __restrict__ wg_cost, const int tile_of_wave = -1) {
    constexpr int BW = 32 / SEG;                               // workgroup footprint width in pixels
    // (Round 6, measured and removed, profiles/r06/foot16_ab.txt: 16 x 16-pixel footprints (2 x 2 tiles) for whole-ray workgroups: frame identical, kernel alone
    // +4.4 %, two frames in flight unchanged)
    const int tiles_x = (G.tile_w + BW - 1) / BW;
    const int local_rows = G.n_bands * G.band_rows;
    const int slab = (int)logical / tiles_x, bx = (int)logical - slab * tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = tile_of_wave >= 0 ? tile_of_wave : wave / SEG, seg = tile_of_wave >= 0 ? 0 : wave - tile * SEG;   // tile_of_wave: SEG == 1 only
    // (lanes in Morton order or in 2x2 quads inside the tile instead of rows of 8: 1.701 / 1.700 vs 1.702 ms per frame, no difference)
    const int gx = bx * BW + tile * 8 + (lane & 7);
    const int lr = slab * 8 + (lane >> 3);
    const bool valid = gx < G.tile_w && lr < local_rows;
    const int band = lr / G.band_rows, rib = lr - band * G.band_rows;
    const int gy = (G.first_band + band * G.band_stride) * G.band_rows + rib;

    const FrameConsts& fc = *fcp;
    T.detail_lds = nullptr;                                    // compile-time constant here: the LDS tap path folds away
    Ray ray = ray_setup(fc, valid ? gx : 0, valid ? gy : 0);
    if (!valid) ray.above = false;
    MarchOut o;
    if constexpr (VARIANT == 0) {
        static_assert(SEG == 1, "the lock-step reference variant marches whole rays");
        o = march(T, fc, ray);
    } else {
        __shared__ float lds[4][VARIANT == 3 ? CQ_FLOATS : Q_FLOATS];
        const int s0 = (fc.primary_steps * seg) / SEG, s1 = (fc.primary_steps * (seg + 1)) / SEG;
        if constexpr (VARIANT == 3) o = march_compact(T, fc, ray, &lds[wave][0], s0, s1);
        else o = march_queue(T, fc, ray, &lds[wave][0], s0, s1);
        if constexpr (SEG > 1) {
            __shared__ float comb[4][5][64];
            comb[wave][0][lane] = o.r; comb[wave][1][lane] = o.g; comb[wave][2][lane] = o.b; comb[wave][3][lane] = o.t; comb[wave][4][lane] = o.a;
            __syncthreads();
            if (seg == 0) {
                float Tr = o.t, na = 1.0f - o.a;
#pragma unroll
                for (int s = 1; s < SEG; s++) {
                    const int w = tile * SEG + s;
                    o.r += Tr * comb[w][0][lane]; o.g += Tr * comb[w][1][lane]; o.b += Tr * comb[w][2][lane];
                    Tr *= comb[w][3][lane]; na *= 1.0f - comb[w][4][lane];
                }
                o.a = sat(1.0f - na); o.t = Tr;
            }
        }
    }
    if (valid && seg == 0) {
        const uint32_t lo = (uint32_t)f2h(o.r) | ((uint32_t)f2h(o.g) << 16), hi = (uint32_t)f2h(o.b) | ((uint32_t)f2h(o.a) << 16);
        out[(size_t)(G.out_full ? gy : lr) * G.pitch_px + gx] = make_uint2(lo, hi);  // imageStore, clouds.glsl:264
    }
    if (stats || wg_cost) {
        unsigned ic = o.incloud, ab = (ray.above && seg == 0) ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) { ic += __shfl_down(ic, off); ab += __shfl_down(ab, off); }
        if (lane == 0) {
            if (stats) { atomicAdd(&stats[0], (unsigned long long)ic); atomicAdd(&stats[1], (unsigned long long)ab); }
            if (wg_cost) atomicAdd(&wg_cost[logical], ic + 16u * ab);   // cost model of the feedback schedule: light marches + live primary marches
        }
    }
}

template <int VARIANT, int SEG, class TS = TexSet>
__global__ __launch_bounds__(256, VARIANT == 3 ? CSKY_COMPACT_WAVES : 7) void clouds_kernel(TS T, const FrameConsts* __restrict__ fcp, RenderGeom G, const uint32_t* __restrict__ order,
                                                     uint2* __restrict__ out, unsigned long long* __restrict__ stats, uint32_t* __restrict__ wg_cost) {
    const uint32_t logical = order[blockIdx.x];
    if (logical == 0xffffffffu) return;                        // workgroup-uniform
    render_block<VARIANT, SEG, TS>(T, fcp, G, logical, blockIdx.x, out, stats, wg_cost);
}

// Persistent form of the whole-ray kernel: the launch is only as large as the chip holds (CUs x resident workgroups) and its
// wavefronts pull work from the launch order until it is empty.  The order is read as eight interleaved sequences (entry 8j + x
// belongs to XCD x, the same assignment the hardware's round-robin gives a plain launch, so the per-XCD L2 locality of mode 5 is
// kept); a footprint is popped from the workgroup's own XCD's sequence (one returning device-scope atomic) and, when that is
// empty, from the others in turn, so that XCDs that finish early take over work of the ones that run late.
//   * The four wavefronts of a workgroup never meet at a barrier: each draws a workgroup-local ticket t from LDS = tile t & 3 of
//     the workgroup's footprint number t >> 2.  The drawer of tile 0 pops that footprint and publishes it in one of two LDS slots;
//     the others wait for the publication (a global atomic's latency at most).  A slot is rewritten only after the three readers
//     of its previous footprint are through (slot_reads).  So a workgroup's wavefronts stay on neighbouring tiles (shared L1
//     lines) without waiting for each other.  Measured alternatives: one pop per workgroup behind a barrier 1.757 ms per C3
//     frame (tickets: 1.724; plain launches 1.806); popping single tiles globally per wavefront scatters a footprint over CUs
//     (2.12 -> 2.34 ms).  Everything that steers the loop is made wave-uniform with readfirstlane: with per-lane values the
//     structurizer wrapped the body in a lane loop that re-entered with ticket 0.
//   * heads[0..7] = the sequences' pop counters, heads[8] = wavefronts that have left; all nine are zero at launch and the last
//     wavefront out zeroes them again (a memset node in front of every launch cost 46 us on a busy chip).
// Used for launches of 12 Ki - 64 Ki wavefronts while two frames are in flight (api.cpp::clouds_dev has the policy and the
// numbers; profiles/r02/persistent_launch_ab.txt).
template <int VARIANT>
__global__ __launch_bounds__(256, VARIANT == 3 ? CSKY_COMPACT_WAVES : 7) void clouds_kernel_persistent(TexSet T, const FrameConsts* __restrict__ fcp, RenderGeom G,
        const uint32_t* __restrict__ order, const uint32_t n_items, uint32_t* __restrict__ heads, uint2* __restrict__ out, unsigned long long* __restrict__ stats,
        uint32_t* __restrict__ wg_cost) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;                                                 // eight sequences: MI355X has 8 XCDs; a part with fewer just leaves sequences to be stolen, one with more folds two XCDs onto one sequence (both still correct: every entry is popped exactly once)
    const uint32_t per_xcd = (n_items + 7u) >> 3;
    __shared__ uint32_t ticket, slot_ready[2], slot_reads[2], slot_entry[2], slot_rec[2];
    if (threadIdx.x == 0) { ticket = 0; slot_ready[0] = slot_ready[1] = 0; slot_reads[0] = slot_reads[1] = 0; }
    __syncthreads();
    const bool lane0 = (threadIdx.x & 63) == 0;
    for (;;) {
        uint32_t tv = 0;
        if (lane0) tv = __hip_atomic_fetch_add(&ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tv);
        const uint32_t f = t >> 2, sl = f & 1u;
        const int tile = (int)(t & 3u);
        uint32_t logical = 0xfffffffeu, rec = 0;               // 0xfffffffe: every sequence is empty
        if (tile == 0) {
            for (unsigned k = 0; k < 8u; k++) {
                const unsigned y = (xcc + k) & 7u;             // (own, then xcc ^ k = the XCD on the same IO die first: 0.3 % slower)
                uint32_t jv = 0;
                if (lane0) jv = atomicAdd(&heads[y], 1u);
                const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)jv);
                const uint32_t i = 8u * j + y;
                if (j < per_xcd && i < n_items) { logical = order[i]; rec = i; break; }
            }
            for (;;) {                                         // the slot's previous footprint (f - 2) had three readers
                const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&slot_reads[sl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (r == 3u * (f >> 1)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane0) {
                slot_entry[sl] = logical; slot_rec[sl] = rec;
                __hip_atomic_store(&slot_ready[sl], f + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            for (;;) {
                const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&slot_ready[sl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (r == f + 1u) break;
                __builtin_amdgcn_s_sleep(1);
            }
            logical = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot_entry[sl]);
            rec = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot_rec[sl]);
            if (lane0) __hip_atomic_fetch_add(&slot_reads[sl], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (logical == 0xfffffffeu) {
            if (lane0 && atomicAdd(&heads[8], 1u) == gridDim.x * 4u - 1u) {
                for (unsigned k = 0; k < 9u; k++) atomicExch(&heads[k], 0u);
            }
            return;
        }
        if (logical != 0xffffffffu) render_block<VARIANT, 1>(T, fcp, G, logical, rec, out, stats, wg_cost, tile);
    }
}

// ---- light-march packet exchange (round 4: built, measured, removed; evidence profiles/r04/exchange_ab.txt; code: commit "exchange: ...") ----
// the persistent form above whose wavefronts, once every sequence is empty, stay and serve the light marches of the tiles
// still being marched.  An owner published a flush (the 7 floats x 64 samples march_compact parks in LDS) to a per-XCD ticket queue in global
// memory instead of running step B, kept marching, and composited the helper's five results per sample IN ORDER later (readlane / ds_bpermute
// replay, 4-8 packets outstanding per wavefront); helpers ran the same light_march_terms().  Transport per cdna_hip_programming.md G16: 8-byte
// {tag, value} granules, relaxed agent-scope (sc1) stores and loads, a per-launch epoch in the tags: no fence, no L1 invalidate.
//   * Frames BYTE-IDENTICAL to clouds_kernel<3,1> in every build (512x256, C3, a rank's 1/8 share), no spin ever hit its bound.
//   * Slower everywhere.  ms per C3 frame alone / two in flight / 1/8 share x1: product 2.03 / 1.69 / 0.41; four protocols 144.8 -> 65.5
//     -> 5.94 -> 2.88 / 2.78 / 0.96 (idle helpers scanning shared words; compare-exchange pops: one winner per round trip with hundreds in
//     flight; owners reading a policy word per flush; finally static per-XCD queues where no shared word is polled at all).  The kernel's own
//     cost with publication compiled out: 2.23 / 2.22 / 0.86 (helpers hold their slots to the end, so the next frame cannot fill the tail).
//   * Why: publication -> results read takes ~43 us (six memory hops at 2-5 us each on a loaded chip) against 6.8 us of work per packet;
//     with four packets outstanding an owner waits ~11 us per flush where running it costs 6.  Hiding it needs ~20 packets in flight per
//     wavefront plus prefetched results in a kernel at its 72-VGPR budget (the prefetch build: 51 spilled registers, 4.0 ms).  The unit of
//     work the compact march can hand over is an order of magnitude too small for a cross-CU hand-off on this chip.

// ---- launch-tail / share experiments of round 2 (measured, removed; evidence under profiles/r02/) -----------------------------
// A whole-frame launch drains for the last ~27 % of its span with the chip 3/4 empty (time-integral of occupancy 72-76 %).  Three
// ways of filling that tail were built and measured on MI355X, none made the frame faster, and the code was removed again:
//   * deadline order: static order, previous-frame costs pull late heavy workgroups forward             2.10 vs 2.07 ms (timeline_static_vs_deadline_order.txt)
//   * mixed-segment launch: the order's last 10-25 % as 2-/4-segment workgroups in the SAME launch: occupancy integral 72 -> 85-87 %,
//     frame 2.06-2.28 vs 2.09 ms: segments add 17 % wave-time and the launch is VALU/L1-throughput bound     (timeline_static_vs_mixed_segment_tail.txt)
//   * adaptive segments per workgroup from previous-frame costs, for one GPU's 1/4..1/16 share             0.446 vs 0.434 ms at 1/8 (share_matrix_adaptive_segments.txt)
// What does fill the tail is the NEXT frame's workgroups (two frames in flight, api.cpp): 2.12 -> 1.80 ms per frame, and with them
// in flight the persistent form above (cross-XCD stealing at the end of a launch): 1.81 -> 1.72 ms per frame.

// ---- cost-feedback schedule (api.cpp, schedule mode 7) -----------------------------------------------------------------
// Workgroups differ 10x in cost (in-cloud samples per tile) and a C3 frame is only ~4 waves of resident workgroups deep, so
// the order they start in decides the tail.  Every launch records a cost per workgroup (wg_cost: in-cloud samples + live rays);
// these three kernels turn it into the NEXT launch's order, heaviest first (longest-processing-time-first list scheduling;
// consecutive blocks land on consecutive XCDs, so each XCD also receives a descending sequence).  Counting sort on 1024 cost
// buckets; ties are placed in arrival order (the schedule may differ run to run, the frame cannot: every ray's arithmetic is
// independent of where and when its workgroup runs, tests/test_gpu_parity.py::test_variants_and_schedules_agree).
constexpr int LPT_BUCKETS = 1024;
__device__ __forceinline__ int lpt_bucket(uint32_t cost, int shift) {
    const uint32_t b = cost >> shift;
    return LPT_BUCKETS - 1 - (int)(b > (uint32_t)(LPT_BUCKETS - 1) ? (uint32_t)(LPT_BUCKETS - 1) : b);   // bucket 0 = heaviest
}
__global__ __launch_bounds__(256) void lpt_hist_kernel(const uint32_t* __restrict__ cost, int n, int shift, uint32_t* __restrict__ hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&hist[lpt_bucket(cost[i], shift)], 1u);
}
__global__ __launch_bounds__(LPT_BUCKETS) void lpt_scan_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ offsets) {
    __shared__ uint32_t sh[LPT_BUCKETS];                       // counts -> exclusive offsets; the histogram is left zeroed for the next launch
    const int t = threadIdx.x;
    const uint32_t own = hist[t];
    hist[t] = 0u;
    sh[t] = own;
    __syncthreads();
    for (int off = 1; off < LPT_BUCKETS; off <<= 1) {
        const uint32_t v = t >= off ? sh[t - off] : 0u;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    offsets[t] = sh[t] - own;
}
__global__ __launch_bounds__(256) void lpt_scatter_kernel(uint32_t* __restrict__ cost, int n, int shift, uint32_t* __restrict__ offsets,
                                                          uint32_t* __restrict__ order) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        order[atomicAdd(&offsets[lpt_bucket(cost[i], shift)], 1u)] = (uint32_t)i;
        cost[i] = 0u;                                          // the next launch accumulates into a clean array: no memset nodes per frame
    }
}
// d_cost[n] and d_scratch[2 * LPT_BUCKETS] must be zero before their first use (api.cpp clears them at allocation); both are left zeroed
hipError_t launch_lpt_order(uint32_t* d_cost, int n, int shift, uint32_t* d_scratch, uint32_t* d_order, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    lpt_hist_kernel<<<(n + 255) / 256, 256, 0, s>>>(d_cost, n, shift, d_scratch);
    lpt_scan_kernel<<<1, LPT_BUCKETS, 0, s>>>(d_scratch, d_scratch + LPT_BUCKETS);
    lpt_scatter_kernel<<<(n + 255) / 256, 256, 0, s>>>(d_cost, n, shift, d_scratch + LPT_BUCKETS, d_order);
    return hipGetLastError();
}

// ---- static workgroup orders, generated on the device (api.cpp::ensure_order) -------------------------------------------
// Physical workgroup b runs on XCD b % 8 (observed placement, used for speed only).  A "slab" is one 32 x 8 pixel workgroup
// footprint (bw x 8 for segmented launches); `grid` entries, 0xffffffff = idle padding.
//   mode 2: natural order
//   mode 1: contiguous eighths of the launch per XCD
//   mode 5: slab ROWS dealt round-robin to the XCDs, every XCD walks its rows left to right: all XCDs see the same mix of
//           elevations and concurrently running workgroups are neighbours (shared cache lines)
// Written by a kernel on the launch's own stream: no host table, no pageable copy, no device-wide synchronisation when the
// geometry changes (a 64-tile walk re-uses the table anyway: these orders depend on the launch geometry only).
__global__ __launch_bounds__(256) void static_order_kernel(int mode, int tiles_x, int slabs, int grid, uint32_t* __restrict__ out) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= grid) return;
    const int nblocks = tiles_x * slabs;
    uint32_t l = 0xffffffffu;
    if (mode == 2) { if (b < nblocks) l = (uint32_t)b; }
    else if (mode == 1) { const int per = (nblocks + 7) >> 3, v = (b & 7) * per + (b >> 3); if ((b >> 3) < per && v < nblocks) l = (uint32_t)v; }
    else { const int x = b & 7, j = b >> 3, k = j / tiles_x, bx = j - k * tiles_x, i = 8 * k + x; if (i < slabs) l = (uint32_t)(i * tiles_x + bx); }
    out[b] = l;
}
hipError_t launch_static_order(int mode, int tiles_x, int slabs, int grid, uint32_t* d_order, hipStream_t s) {
    if (grid <= 0) return hipSuccess;
    static_order_kernel<<<(grid + 255) / 256, 256, 0, s>>>(mode, tiles_x, slabs, grid, d_order);
    return hipGetLastError();
}

static const char* const kVariantNames[] = {"lockstep", "queue", "queue-lds", "compact"};
int cloud_resident_workgroups_per_cu() { return CSKY_COMPACT_WAVES; }
int cloud_variant_count() { return (int)(sizeof(kVariantNames) / sizeof(kVariantNames[0])); }
const char* cloud_variant_name(int v) { return (v >= 0 && v < cloud_variant_count()) ? kVariantNames[v] : nullptr; }

hipError_t launch_clouds(int variant, int seg, const TexSet& t, const FrameConsts* d_fc, const RenderGeom& g, const uint32_t* d_order, int grid,
                         uint2* d_out, unsigned long long* d_stats, uint32_t* d_wg_cost, hipStream_t s, uint32_t* d_heads, int resident, const TexSet32* t32) {
    if (grid <= 0) return hipSuccess;
    if (t32) {                                                 // exact fp32-coefficient cells: the compact whole-ray kernel on the other texture-set type
        if (variant != 3 || seg != 1) return hipErrorInvalidValue;
        clouds_kernel<3, 1, TexSet32><<<grid, 256, 0, s>>>(*t32, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
        return hipGetLastError();
    }
    if (d_heads && variant == 3 && seg == 1) {                 // persistent form, see clouds_kernel_persistent
        clouds_kernel_persistent<3><<<grid < resident ? grid : resident, 256, 0, s>>>(t, d_fc, g, d_order, (uint32_t)grid, d_heads, d_out, d_stats, d_wg_cost);
        return hipGetLastError();
    }
    if (variant == 0 && seg == 1) clouds_kernel<0, 1><<<grid, 256, 0, s>>>(t, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
    else if (variant == 1 && seg == 1) clouds_kernel<1, 1><<<grid, 256, 0, s>>>(t, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
    else if (variant == 1 && seg == 2) clouds_kernel<1, 2><<<grid, 256, 0, s>>>(t, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
    else if (variant == 1 && seg == 4) clouds_kernel<1, 4><<<grid, 256, 0, s>>>(t, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
    else if (variant == 3 && seg == 1) clouds_kernel<3, 1><<<grid, 256, 0, s>>>(t, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
    else if (variant == 3 && seg == 2) clouds_kernel<3, 2><<<grid, 256, 0, s>>>(t, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
    else if (variant == 3 && seg == 4) clouds_kernel<3, 4><<<grid, 256, 0, s>>>(t, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
    else if ((variant == 1 || variant == 3) && seg == 5) {                      // 5 = 4 interleaved segments, one tile per workgroup, 76 KB of LDS
        // > 64 KB of dynamic LDS needs the opt-in attribute; it is per device, so set it on every launch (cheap, idempotent)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&clouds_kernel_interleaved<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(IL_BLOCK_FLOATS * sizeof(float)));
        if (e != hipSuccess) return e;
        clouds_kernel_interleaved<0><<<grid, 256, IL_BLOCK_FLOATS * sizeof(float), s>>>(t, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
    }
    else if (variant == 2) {                                  // detail noise staged in LDS, 16 wavefronts per workgroup
        constexpr size_t bytes = ((DETAIL_CHAIN_TEXELS + 7) / 8 * 4 + 16 * Q_FLOATS) * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&clouds_kernel_lds<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        clouds_kernel_lds<0><<<grid, 1024, bytes, s>>>(t, d_fc, g, d_order, d_out, d_stats, d_wg_cost);
    }
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace csky
