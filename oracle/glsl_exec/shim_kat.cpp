/*
 * shim_kat.cpp -- TEST INFRASTRUCTURE ONLY.  Known-answer tests of glsl_shim.hpp against the GLSL 4.50 SPECIFICATION (not against the oracle, and not
 * against the reference): the oracle is bit-identical to the reference's shader text executed under the shim, so a misreading of GLSL shared by the shim
 * and the oracle (both written by one author) would go unseen there.  Every expectation below is a value the language specification fixes:
 * constructor and matrix layout (5.4.2: column-major), swizzle read / write (5.5), built-in definitions (8.1-8.5), texel-centre sampling and
 * CLAMP_TO_EDGE / REPEAT addressing (Vulkan 1.3 spec 16.5-16.6), image stores outside the image being discarded, and the constant-folding model of the
 * header's comment (values computed by hand).  The GLSL-syntax part is written in GLSL -- compiled under the same six #defines the reference's text is
 * compiled under -- so the syntax path (swizzle assignment, `out` references, array constructors, uniform blocks) is what is tested.
 * Build + run: tests/test_glsl_shim_kat.py (g++ -std=c++17 -ffp-contract=off, fold and float variants).  Runs anywhere: needs no reference, no GPU.
 */
#include <cstdio>
#include <cstring>
#include <vector>

#include "glsl_shim.hpp"

static int g_fail = 0;
typedef float real_t;                                       /* (`float` is redefined for the GLSL part below) */
static unsigned bits(real_t f) { unsigned u; memcpy(&u, &f, 4); return u; }
static real_t val(const gx::F &f) { return f.f(); }
static real_t val(const gx::swz<1> &f) { return f.p[0]->f(); }
static real_t val(double f) { return (real_t)f; }
#define EXPECT_BITS(expr, want)                                                                                                       \
    do { real_t got_ = val(expr); real_t want_ = val(want); if (bits(got_) != bits(want_)) { g_fail++; printf("FAIL line %d  %s = %.9g (0x%08x), want %.9g (0x%08x)\n", __LINE__, #expr, got_, bits(got_), want_, bits(want_)); } } while (0)
#define EXPECT_TRUE(expr) do { if (!(expr)) { g_fail++; printf("FAIL line %d  %s\n", __LINE__, #expr); } } while (0)

/* GLSL-BEGIN / GLSL-END mark the part tests/test_glsl_shim_kat.py passes through the SAME mechanical rewrites as the reference's text (REWRITES of
 * make_glsl_fixtures.py: `out` parameters -> references, float-literal suffix stripped) before compiling: the rewrites are under test too. */
namespace kat {
using namespace gx;
#define float gx::F
#define in
#define uniform struct
#define layout(...)
#define restrict
#define writeonly
/* GLSL-BEGIN ------------------------------------------------------- (own text, GLSL 4.50 subset) */
layout(local_size_x = 8, local_size_y = 8, local_size_z = 1) in;
layout(rgba16f, set = 0, binding = 0) uniform restrict writeonly image2D target;
layout(set = 1, binding = 0) uniform sampler2D clamp_tex;
layout(set = 1, binding = 1) uniform sampler2D repeat_tex;
layout(set = 1, binding = 2) uniform sampler3D volume;
layout(push_constant, std430) uniform Block {
	vec2 size;
	vec3 dir;
	float k;
} pc;

const float g = 0.8;
const float gg = g * g;                       // a constant expression: folded in double by glslang (0.64 -> 0x3f23d70a), 0.8f * 0.8f = 0x3f23d70b in float
const vec4 weights = vec4(1.0, 2.0, 3.0, 4.0) * 0.1;
const mat4x3 M = mat4x3(1.0, 2.0, 3.0,   4.0, 5.0, 6.0,   7.0, 8.0, 9.0,   10.0, 11.0, 12.0);   // four COLUMNS of three

void two_outputs(in float x, out vec4 a, out float b) {
	a = vec4(x, x * 2.0, x * 3.0, x * 4.0);
	b = x + 1.0;
}

vec2 perp(vec2 v) {
	vec2 s;
	s.x = v.x >= 0.0 ? 1.0 : -1.0;
	s.y = v.y >= 0.0 ? 1.0 : -1.0;
	return (1.0 - abs(v.yx)) * s;
}

float run_time_square(float x) { return x * x; }      // a parameter is never a constant expression, whatever the caller passes

void glsl_tests(float one, float third, float sq08, float w3, float ss07) {   // run-time values: 1.0, float(1.0 / 3.0), and three expectations computed in plain C float arithmetic
	// ---- constructors, component names, swizzle reads
	vec4 v = vec4(1.0, 2.0, 3.0, 4.0);
	EXPECT_BITS(v.x, 1.0f); EXPECT_BITS(v.g, 2.0f); EXPECT_BITS(v.b, 3.0f); EXPECT_BITS(v.w, 4.0f);
	vec3 s = v.xyz;
	vec3 t = s.xzy;
	EXPECT_BITS(t.x, 1.0f); EXPECT_BITS(t.y, 3.0f); EXPECT_BITS(t.z, 2.0f);
	vec2 yx = t.yx;
	EXPECT_BITS(yx.x, 3.0f); EXPECT_BITS(yx.y, 1.0f);
	vec3 splat = vec3(0.5);
	EXPECT_BITS(splat.z, 0.5f);
	vec4 ext = vec4(s, 9.0);
	EXPECT_BITS(ext.z, 3.0f); EXPECT_BITS(ext.w, 9.0f);
	// ---- swizzle writes
	vec3 p = vec3(10.0, 20.0, 30.0);
	p.xz += vec2(1.0, 2.0) * 0.5;
	EXPECT_BITS(p.x, 10.5f); EXPECT_BITS(p.y, 20.0f); EXPECT_BITS(p.z, 31.0f);
	p.xz -= vec2(0.5, 1.0);
	p.y -= 5.0;
	EXPECT_BITS(p.x, 10.0f); EXPECT_BITS(p.y, 15.0f); EXPECT_BITS(p.z, 30.0f);
	vec3 n = vec3(-0.25, 0.5, -1.0);
	n.xy = n.z >= 0.0 ? n.xy : perp(n.xy);             // conditional with a swizzle on one side, swizzle assignment
	EXPECT_BITS(n.x, -0.5f); EXPECT_BITS(n.y, 0.75f);
	vec3 q = vec3(1.0, 2.0, 3.0);
	q.xy = q.yx;                                       // all components are read before any is written
	EXPECT_BITS(q.x, 2.0f); EXPECT_BITS(q.y, 1.0f);
	q *= 2.0; q /= vec3(4.0, 2.0, 1.0);
	EXPECT_BITS(q.x, 1.0f); EXPECT_BITS(q.y, 1.0f); EXPECT_BITS(q.z, 6.0f);
	// ---- out parameters, arrays, int <-> float
	vec4 a; float b;
	two_outputs(1.5, a, b);
	EXPECT_BITS(a.w, 6.0f); EXPECT_BITS(b, 2.5f);
	const vec3 table[3] = {vec3(1.0, 0.0, 0.0), vec3(0.0, 2.0, 0.0), vec3(0.0, 0.0, 3.0)};
	float acc = 0.0;
	for (int j = 0; j < 3; j++) acc += dot(table[j] * float(j + 1), vec3(1.0));
	EXPECT_BITS(acc, 14.0f);
	float steps = 128.0;
	EXPECT_TRUE(int(steps) == 128 && int(-2.75 * one) == -2);                       // float -> int truncates towards zero
	ivec2 px = ivec2(vec2(7.9, -0.5) * one) + ivec2(1, 1);
	EXPECT_TRUE(px.x == 8 && px.y == 1);
	vec2 fpx = vec2(px) / vec2(16.0, 2.0);
	EXPECT_BITS(fpx.x, 0.5f); EXPECT_BITS(fpx.y, 0.5f);
	// ---- matrix: column-major constructor, M * v = sum of columns
	vec3 c0 = M * vec4(1.0, 0.0, 0.0, 0.0), c3 = M * vec4(0.0, 0.0, 0.0, 1.0), all = M * vec4(one);
	EXPECT_BITS(c0.x, 1.0f); EXPECT_BITS(c0.y, 2.0f); EXPECT_BITS(c0.z, 3.0f);
	EXPECT_BITS(c3.x, 10.0f); EXPECT_BITS(c3.z, 12.0f);
	EXPECT_BITS(all.x, 22.0f); EXPECT_BITS(all.y, 26.0f); EXPECT_BITS(all.z, 30.0f);
	// ---- built-ins (GLSL 4.50 chapter 8)
	EXPECT_BITS(mix(2.0 * one, 6.0, 0.25), 3.0f);                                   // x (1 - a) + y a
	EXPECT_BITS(clamp(5.0 * one, 0.0, 1.0), 1.0f); EXPECT_BITS(clamp(-5.0 * one, 0.0, 1.0), 0.0f);
	EXPECT_BITS(smoothstep(0.0, 1.0, 0.5 * one), 0.5f); EXPECT_BITS(smoothstep(2.0, 4.0, 5.0 * one), 1.0f); EXPECT_BITS(smoothstep(2.0, 4.0, one), 0.0f);
	EXPECT_BITS(smoothstep(0.0, 4.0, one), 0.15625f);                               // t = 0.25: t t (3 - 2 t)
	EXPECT_BITS(fract(-0.25 * one), 0.75f); EXPECT_BITS(fract(2.5 * one), 0.5f);    // x - floor(x)
	EXPECT_BITS(sign(-3.0 * one), -1.0f); EXPECT_BITS(sign(0.0 * one), 0.0f); EXPECT_BITS(sign(2.0 * one), 1.0f);
	EXPECT_BITS(abs(-2.0 * one), 2.0f); EXPECT_BITS(max(one, 2.0), 2.0f); EXPECT_BITS(min(one, 2.0), 1.0f);
	EXPECT_BITS(atan(one, 0.0 * one), 1.57079637f);                                 // atan(y, x): the angle of (x, y) = (0, 1)
	EXPECT_BITS(atan(0.0 * one, -one), 3.14159274f);
	EXPECT_BITS(asin(one), 1.57079637f); EXPECT_BITS(pow(2.0 * one, 10.0), 1024.0f); EXPECT_BITS(exp(0.0 * one), 1.0f); EXPECT_BITS(sqrt(16.0 * one), 4.0f);
	EXPECT_BITS(length(vec3(3.0, 0.0, 4.0) * one), 5.0f);
	vec3 nn = normalize(vec3(3.0, 0.0, 4.0) * one);
	EXPECT_BITS(nn.x, 0.6f); EXPECT_BITS(nn.z, 0.8f);
	EXPECT_BITS(dot(vec3(1.0, 2.0, 3.0) * one, vec3(4.0, 5.0, 6.0)), 32.0f);
	vec4 e = exp(vec4(0.0) * one), mx = max(vec4(-1.0, 2.0, -3.0, 4.0) * one, 0.0);
	EXPECT_BITS(e.w, 1.0f); EXPECT_BITS(mx.x, 0.0f); EXPECT_BITS(mx.w, 4.0f);
	vec3 cl = clamp(vec3(-1.0, 50.0, 200.0) * one, vec3(0.0), vec3(100.0)), sm = smoothstep(0.0, 2.0, vec3(0.0, 1.0, 2.0) * one);
	EXPECT_BITS(cl.x, 0.0f); EXPECT_BITS(cl.y, 50.0f); EXPECT_BITS(cl.z, 100.0f); EXPECT_BITS(sm.y, 0.5f);
	EXPECT_BITS(-vec2(one, 2.0).y, -2.0f);
	// ---- constant folding (the model of glsl_shim.hpp's header comment; expected bits computed by hand)
	float g_rt = g;                                                                // a run-time copy of the constant: float(0.8)
	if (GX_FOLD_DOUBLE) {
		EXPECT_BITS(gg * one, 0.64f);                                              // 0.8 * 0.8 in double -> 0x3f23d70a
		EXPECT_BITS((1.0 - gg) * one, 0.36f);                                      // 1 - 0.64 in double -> float(0.36)
		EXPECT_BITS(weights.z * one, 0.3f);                                        // 3.0 * 0.1 in double = 0.30000000000000004 -> float(0.3)
	} else {
		EXPECT_BITS(gg * one, sq08);                                               // 0.8f * 0.8f = 0x3f23d70b
		EXPECT_BITS(weights.z * one, w3);                                          // 3.0f * 0.1f
	}
	EXPECT_BITS(g_rt * g_rt, sq08);                                                // run-time operands: float arithmetic in BOTH variants
	EXPECT_BITS(run_time_square(0.8), sq08);                                       // a literal handed to a parameter is a run-time value inside the function
	EXPECT_BITS(run_time_square(g), sq08);
	EXPECT_BITS(third * 3.0, 1.0f);                                                // float(1/3) * 3 rounds to 1 in float
	EXPECT_BITS(smoothstep(0.6, 1.0, 0.7 * one), ss07);                            // built-in with a run-time argument: (0.7f - 0.6f) / (1.0f - 0.6f), all in FLOAT
	// ---- uniform block + push constants are plain run-time data
	vec3 d2 = pc.dir.xzy;
	EXPECT_BITS(pc.size.x * pc.k, 32.0f); EXPECT_BITS(d2.y, 3.0f);
	// ---- textures: texel centres, CLAMP_TO_EDGE, REPEAT, integer LOD; image stores
	EXPECT_BITS(texture(clamp_tex, vec2(0.5 / 4.0, 0.5 / 2.0)).r, 0.0f);            // centre of texel (0, 0)
	EXPECT_BITS(texture(clamp_tex, vec2(1.5 / 4.0, 0.5 / 2.0)).r, 1.0f);            // centre of texel (1, 0)
	EXPECT_BITS(texture(clamp_tex, vec2(1.0 / 4.0, 0.5 / 2.0)).r, 0.5f);            // half way between them
	EXPECT_BITS(texture(clamp_tex, vec2(-3.0, 0.5 / 2.0)).r, 0.0f);                 // clamped to the edge texel
	EXPECT_BITS(texture(clamp_tex, vec2(7.0, 1.5 / 2.0)).g, 13.0f);                 // texel (3, 1) holds g = 10 + 3
	EXPECT_BITS(textureLod(clamp_tex, vec2(2.5 / 4.0, 1.5 / 2.0), 0.0).a, 1.0f);
	EXPECT_BITS(texture(repeat_tex, vec2(0.5 / 512.0, 0.5 / 512.0) + vec2(3.0, -2.0)).b, 0.2f);   // REPEAT: whole periods vanish; texel (0,0) = 51/255
	EXPECT_BITS(textureLod(volume, vec3(0.5 / 4.0), 0.0).r, 16.0f / 255.0f);        // level 0 texel (0,0,0)
	EXPECT_BITS(textureLod(volume, vec3(0.5 / 4.0) + vec3(1.0, 2.0, -1.0), 0.0).r, 16.0f / 255.0f);
	EXPECT_BITS(textureLod(volume, vec3(0.25), 1.0).g, 32.0f / 255.0f);             // level 1 (2^3): the centre of its texel (0,0,0)
	EXPECT_BITS(textureLod(volume, vec3(0.25), -3.0).r, textureLod(volume, vec3(0.25), 0.0).r);   // LOD clamps at 0 ...
	EXPECT_BITS(textureLod(volume, vec3(0.5), 9.0).g, 64.0f / 255.0f);              // ... and at the last level (1^3)
	imageStore(target, ivec2(1, 1), vec4(1.0, 0.5, 65520.0, -2.0));
	imageStore(target, ivec2(2, 0), vec4(0.1, 0.0, 0.0, 0.0));
	imageStore(target, ivec2(-1, 0), vec4(7.0)); imageStore(target, ivec2(4, 0), vec4(7.0)); imageStore(target, ivec2(0, 2), vec4(7.0));   // outside: discarded
}
/* GLSL-END */
#undef float
#undef in
#undef uniform
#undef layout
#undef restrict
#undef writeonly
} /* namespace kat */

int main() {
    using namespace kat;
    /* a 4 x 2 RGBA16F image: r = x, g = 10 + x (row 1), a = 1 */
    std::vector<uint16_t> img(4 * 2 * 4);
    for (int y = 0; y < 2; y++) for (int x = 0; x < 4; x++) {
        uint16_t *t = &img[(y * 4 + x) * 4];
        t[0] = csko_f2h((float)x); t[1] = csko_f2h(y ? 10.0f + (float)x : 0.0f); t[2] = csko_f2h(0.0f); t[3] = csko_f2h(1.0f);
    }
    std::vector<uint8_t> weather(512 * 512 * 3, 0);
    weather[2] = 51;                                                   /* texel (0,0).b = 51/255 = 0.2 */
    for (int x = 1; x < 512; x++) weather[x * 3 + 2] = 51;             /* the whole first row and ... */
    for (int y = 1; y < 512; y++) weather[(size_t)y * 512 * 3 + 2] = 51, weather[((size_t)y * 512 + 511) * 3 + 2] = 51, weather[(size_t)511 * 512 * 3 + y * 3 + 2] = 51;
    weather[((size_t)511 * 512 + 511) * 3 + 2] = 51;                   /* ... its wrap-around neighbours, so that the tap at the texel centre is exact */
    /* a 4^3 one-channel... stored as RGB: level 0 all r = 16 / g = 0, level 1 (2^3) g = 32, level 2 (1^3) g = 64 */
    const int n = 4, ch = 3;
    std::vector<uint8_t> vol(csko_mip_total(n, 3, ch), 0);
    for (size_t i = 0; i < (size_t)n * n * n; i++) vol[i * ch + 0] = 16;
    { uint8_t *l1 = vol.data() + csko_mip_offset(n, 1, ch); for (int i = 0; i < 8; i++) l1[i * ch + 1] = 32; }
    { uint8_t *l2 = vol.data() + csko_mip_offset(n, 2, ch); l2[1] = 64; }
    clamp_tex = gx::sampler2D{1, img.data(), 4, 2};
    repeat_tex = gx::sampler2D{0, weather.data(), 512, 512};
    volume = gx::sampler3D{vol.data(), n, 3, ch};
    std::vector<uint16_t> out(4 * 2 * 4, 0x1234);
    target = gx::image2D{out.data(), 4, 2, 0, 0, 16};
    pc.size = gx::vec2(gx::F(16.0f), gx::F(8.0f)); pc.dir = gx::vec3(gx::F(1.0f), gx::F(2.0f), gx::F(3.0f)); pc.k = gx::F(2.0f);
    real_t t07 = (0.7f - 0.6f) / (1.0f - 0.6f); t07 = t07 < 0.0f ? 0.0f : (t07 > 1.0f ? 1.0f : t07);
    glsl_tests(gx::F(1.0f), gx::F(1.0f / 3.0f), gx::F(0.8f * 0.8f), gx::F(3.0f * 0.1f), gx::F(t07 * t07 * (3.0f - 2.0f * t07)));
    /* image stores: RGBA16F, round to nearest even, 65520 -> +inf, outside writes discarded, untouched texels untouched */
    const uint16_t *t11 = &out[(1 * 4 + 1) * 4], *t20 = &out[(0 * 4 + 2) * 4];
    EXPECT_TRUE(t11[0] == 0x3c00 && t11[1] == 0x3800 && t11[2] == 0x7c00 && t11[3] == 0xc000);
    EXPECT_TRUE(t20[0] == 0x2e66);                                     /* 0.1 -> 0x2e66 (RNE) */
    int touched = 0; for (size_t i = 0; i < out.size(); i += 4) touched += out[i] != 0x1234;
    EXPECT_TRUE(touched == 2);
    printf("%s: %d failure(s), variant %s\n", g_fail ? "FAILED" : "shim KAT ok", g_fail, GX_FOLD_DOUBLE ? "fold" : "float");
    return g_fail ? 1 : 0;
}
