/*
 * cloudsky_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar fp32 CPU restatement of the three compute shaders of
 * clayjohn/godot-volumetric-cloud-demo-v2 (cloud_sky/clouds.glsl,
 * cloud_sky/sky-lut.glsl, cloud_sky/transmittance-lut.glsl).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / reported baseline.  The product (libcloudsky.so) never
 * links, loads or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference holds no golden vectors, no tests and no dumped
 * frame for this path, and it cannot be executed here (GLSL-for-Vulkan +
 * GDScript; no Godot, Vulkan or GLSL compiler in the image).  The oracle is
 * therefore checked only against (i) an independent numpy fp32 restatement
 * (oracle/numpy_restatement.py -> tests/golden/), (ii) structural known-answer
 * properties derivable from the GLSL text (tests/test_oracle_*.py) and, since
 * round 6, (iii) the reference's OWN shader text (the three .glsl files and
 * clouds.gdshader, read from /root/reference when the fixture is generated,
 * never stored here) compiled under a C++ GLSL-subset shim and executed on the
 * CPU (oracle/glsl_exec -> tests/golden/glslexec.npz): this file is
 * BIT-IDENTICAL to it on every fixture (tests/test_oracle_glslexec.py), which
 * takes the hand transcription out of the trusted base.  The shim is a
 * builder-written stand-in for the GLSL runtime, so by the task's rules the
 * parity stays UNPINNED: what it cannot vouch for is listed next.
 * Sampler filtering, fp16 store rounding, 3-D mip generation and BC7 texture
 * compression are supplied by Godot Engine (>=4.2, un-vendored) and are restated
 * from the Vulkan specification / defined here, see DESIGN.md.
 */
#ifndef CLOUDSKY_ORACLE_H
#define CLOUDSKY_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Mip-chained 8-bit textures, level 0 first, tightly packed. */
typedef struct {
    const uint8_t *large_rgba8;   /* 128^3 RGBA8 + mips (level l at csko_mip_offset(128,l,4)); index ((z*N+y)*N+x)*4 */
    int large_levels;             /* number of levels present (>=4 needed: LODs 0..3 are hit) */
    const uint8_t *small_rgb8;    /* 32^3 RGB8 + mips (levels 0..5)                                              */
    int small_levels;             /* 6                                                                          */
    const uint8_t *weather_rgb8;  /* 512x512 RGB8, row-major, row 0 = top row of the bitmap, no mips             */
} csko_textures;

typedef struct {
    uint64_t primary_samples;     /* rays_above_horizon * primary_steps                     */
    uint64_t incloud_samples;     /* primary samples with density > 0 (light march executed) */
    uint64_t rays;                /* pixels rendered                                         */
    uint64_t rays_marched;        /* pixels with dir.y > 0                                   */
} csko_stats;

/* byte offset of mip level `level` of an N^3 volume with `ch` bytes per texel */
size_t csko_mip_offset(int n, int level, int ch);
size_t csko_mip_total(int n, int levels, int ch);
/* 2x2x2 box filter, round-half-up in integers ((sum+4)>>3); writes levels 1..levels-1 after level 0 */
void csko_build_mips(uint8_t *vol, int n, int ch, int levels);

/* float <-> half (IEEE binary16, round-to-nearest-even) */
uint16_t csko_f2h(float f);
float csko_h2f(uint16_t h);

/* transmittance-lut.glsl:157-196.  out: w*h*4 halfs, row-major, row 0 = pos.y==0 */
void csko_transmittance_lut(int w, int h, uint16_t *out_rgba16f);

/* sky-lut.glsl:278-315.  trans: the (tw x th) RGBA16F transmittance LUT (sampled bilinear CLAMP) */
void csko_sky_lut(int w, int h, const float sun_dir[3], const uint16_t *trans, int tw, int th,
                  uint16_t *out_rgba16f);

/* clouds.glsl:258-266 over the pixel rectangle [x0,x0+w) x [y0,y0+h) (gl_GlobalInvocationID = local
 * pixel; params[2..3] = update_position is ADDED on top, exactly like the shader).  params = the 28-float
 * push-constant block (clouds.glsl:18-40).  primary_steps/light_steps generalise the literals 128
 * (clouds.glsl:228) and 6 (clouds.glsl:186).  out row pitch in bytes; pixel (gx,gy) of the rectangle is
 * written at out + gy*pitch + gx*8.  nthreads > 1 uses OpenMP over rows.  stats may be NULL. */
void csko_clouds(const csko_textures *tex, const float params[28], int primary_steps, int light_steps,
                 const uint16_t *sky_lut, int sw, int sh, int gx0, int gy0, int w, int h,
                 uint16_t *out_rgba16f, size_t pitch_bytes, int nthreads, csko_stats *stats);

/* csko_clouds over a band set (same meaning as the product's csky_bands), compact output rows */
void csko_clouds_bands(const csko_textures *tex, const float params[28], int primary_steps, int light_steps,
                       const uint16_t *sky_lut, int sw, int sh, int tile_w, int band_rows, int first_band, int band_stride,
                       int n_bands, uint16_t *out_rgba16f, int nthreads, csko_stats *stats);

/* clouds.gdshader:105-116 sky() evaluated on an equirectangular panorama (build-side EYEDIR mapping, see the .c file).
 * cloud_from/to: RGBA16F cloud textures (cw x ch); sky_from/to: RGBA16F sky LUTs; trans: transmittance LUT. */
void csko_composite(int out_w, int out_h, const uint16_t *cloud_from, const uint16_t *cloud_to, int cw, int ch, const uint16_t *sky_from,
                    const uint16_t *sky_to, int sw, int sh, const uint16_t *trans, int tw, int th, float blend_amount,
                    float sun_disk_scale, const float light_dir[3], uint16_t *out_rgba16f);
void csko_panorama_eyedir(int out_w, int out_h, int i, int j, float eye[3]);   /* EYEDIR of panorama pixel (i, j): csko_composite's mapping */
void csko_composite_view(int out_w, int out_h, const float basis[9], float fov_y_degrees, const uint16_t *cloud_from, const uint16_t *cloud_to, int cw, int ch,
                         const uint16_t *sky_from, const uint16_t *sky_to, int sw, int sh, const uint16_t *trans, int tw, int th, float blend_amount,
                         float sun_disk_scale, const float light_dir[3], uint16_t *out_rgba16f);

/* one-tap probes of the samplers above (REPEAT/LINEAR 3-D at an integer LOD; REPEAT/LINEAR weather; CLAMP/LINEAR RGBA16F):
 * the bindings of oracle/glsl_exec's texture()/textureLod() */
void csko_tap3d_repeat(const uint8_t *chain, int n0, int levels, int ch, float lod, const float s[3], float out[4]);
void csko_tap_weather(const uint8_t *weather_rgb8, float sx, float sy, float out[3]);
void csko_tap_rgba16f_clamp(const uint16_t *t, int w, int h, float sx, float sy, float out[4]);

/* probes used by the structural tests */
float csko_hash_probe(float px, float py, float pz);                     /* clouds.glsl:60-64 on pos*10 */
void csko_pixel_dir(const float params[28], int px, int py, float dir[3]); /* clouds.glsl:260-262        */
void csko_sky_lut_lookup(const uint16_t *sky_lut, int sw, int sh, const float dir[3], float rgb[3]); /* :49-57 */
float csko_density_probe(const csko_textures *tex, const float params[28], const float p[3],
                         const float weather[3], float mip);              /* clouds.glsl:109-137         */
int csko_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
