// kernels.h -- host-callable launchers of the gfx950 kernels (kernels.hip).  Internal to libcloudsky.
#pragma once
#include <hip/hip_runtime_api.h>
#include "csky_common.h"
#include "composite_core.h"
#include "noise_core.h"

namespace csky {

// transmittance-lut.glsl main(): writes the RGBA16F image and a float4 copy of the fp16-ROUNDED values
// (what a sampler would read back), so later kernels sample floats without per-tap half unpacking.
hipError_t launch_transmittance(int w, int h, uint16_t* d_half, float4* d_float, hipStream_t s);
// sky-lut.glsl main()
hipError_t launch_sky_lut(int w, int h, const float sun[3], const float4* d_trans, int tw, int th, uint16_t* d_half,
                          float4* d_float, hipStream_t s);
// rows row0, row0 + row_stride, ... of that LUT (one rank / device of an N-way frame split): d_whole_f == nullptr: compact RGBA16F into d_rows;
// otherwise at their own place in the whole LUT d_rows (RGBA16F) + d_whole_f (the float copy)
hipError_t launch_sky_lut_rows(int w, int h, int row0, int row_stride, const float sun[3], const float4* d_trans, int tw, int th, uint2* d_rows, float4* d_whole_f,
                               hipStream_t s);
// per-frame constants of clouds.glsl:143-170 (one wave)
hipError_t launch_frame_setup(const CloudParams& p, const float4* d_sky, int sw, int sh, int primary_steps, int light_steps,
                              float early_eps, float hf_lo, float hf_hi, int ct_mode, FrameConsts* d_fc, hipStream_t s);
// the same without a sky LUT in memory: renders the <= 12 texels the set-up filters itself (sw x sh LUT of the sun `sun`)
hipError_t launch_frame_setup_taps(const CloudParams& p, const float sun[3], const float4* d_trans, int tw, int th, int sw, int sh, int primary_steps,
                                   int light_steps, float early_eps, float hf_lo, float hf_hi, int ct_mode, FrameConsts* d_fc, hipStream_t s);
// clouds.glsl main() over the rows described by `g`.  d_stats (may be null): [0] += in-cloud samples,
// [1] += rays above the horizon.
// seg = ray segments per ray (1, 2 or 4; variant 1 only): a workgroup covers 4/seg tiles of 8x8 pixels.
// d_order[grid]: physical workgroup -> workgroup-footprint id (0xffffffff = idle), see api.cpp::build_schedule.
hipError_t launch_clouds(int variant, int seg, const TexSet& t, const FrameConsts* d_fc, const RenderGeom& g, const uint32_t* d_order, int grid,
                         uint2* d_out, unsigned long long* d_stats, uint32_t* d_wg_cost, hipStream_t s, uint32_t* d_heads = nullptr, int resident = 0,
                         const TexSet32* t32 = nullptr);   // t32: march on the exact fp32-coefficient cells (variant 3, seg 1 only)
// resident 256-thread workgroups per CU of the "compact" kernel (its launch bound): the size of a persistent launch
int cloud_resident_workgroups_per_cu();
// next launch's workgroup order from this launch's per-workgroup costs, heaviest first.  d_cost[n] and d_scratch[2048] must be zero
// before their first use and are left zeroed (the cloud kernel accumulates the next costs into d_cost).
hipError_t launch_lpt_order(uint32_t* d_cost, int n, int shift, uint32_t* d_scratch, uint32_t* d_order, hipStream_t s);

// static workgroup orders 1, 2, 5 written on the device (kernels.hip); grid = padded number of physical workgroups
hipError_t launch_static_order(int mode, int tiles_x, int slabs, int grid, uint32_t* d_order, hipStream_t s);
// clouds.gdshader sky() on an equirectangular panorama (all pointers in `a` are device pointers)
hipError_t launch_composite(const CompositeArgs& a, uint2* d_out, hipStream_t s);

// stand-in shape noise bake: n^3 RGBA8 voxels (little-endian u32 = r | g<<8 | b<<16 | a<<24)
hipError_t launch_shape_noise(uint32_t seed, int n, const ShapeNoiseParams& P, uint32_t* d_out, hipStream_t s);

// generated 32^3 RGB detail volume (noise_core.h::detail_voxel), 3 bytes per voxel
hipError_t launch_detail_noise(uint32_t seed, int n, uint8_t* d_out, hipStream_t s);
// frame band k (band_bytes each, total_bands of them) = member k % members, local band k / members of a gathered rank-major buffer
hipError_t launch_interleave_bands(const void* d_gathered, size_t member_stride_bytes, int members, size_t band_bytes, int total_bands, void* d_frame, hipStream_t s);
// BC7 (BPTC) blocks of n_img images of w x h RGBA8 texels (bc7enc.hip; what compress/mode=2 of the *.import files asks the importer for)
hipError_t launch_bc7_encode(const uint8_t* d_img, int w, int h, int n_img, int quality, uint4* d_blocks, hipStream_t s);
// 2x2x2 box mips of a device chain whose level 0 is filled (level l at chain_offset(n, l, ch))
hipError_t launch_mip_chain(uint8_t* d_chain, int n, int ch, int levels, hipStream_t s);
// the three device texture layouts from the 8-bit chains; *d_inexact += coefficients not exact in fp16; d_range = {min R, max R, max B} of the
// weather map (initialise to {255, 0, 0})
hipError_t launch_bake(const uint8_t* d_large_chain, const uint8_t* d_small_chain, const uint8_t* d_weather, ShapeTexel* d_shape, uint4* d_detail, uint16_t* d_detail_h,
                       uint4* d_weather_out, unsigned long long* d_inexact, int* d_range, hipStream_t s);

// exact cells (bake_core.h): fp32-coefficient layouts of the same three textures
hipError_t launch_bake32(const uint8_t* d_large_chain, const uint8_t* d_small_chain, const uint8_t* d_weather, float4* d_shape32, float4* d_detail32, float4* d_weather32, hipStream_t s);

// test hook: cloud_core.h::sqrt_shell over an array
hipError_t launch_sqrt_shell(const float* d_in, float* d_out, size_t n, hipStream_t s);

int cloud_variant_count();
const char* cloud_variant_name(int v);

}  // namespace csky
