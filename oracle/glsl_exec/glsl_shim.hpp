/*
 * glsl_shim.hpp -- TEST INFRASTRUCTURE ONLY (build container only; nothing here travels to the GPU box or into the product).
 *
 * A C++17 stand-in for the subset of the GLSL 4.50 *language and built-in library* that the reference's three compute shaders
 * and its sky shader use (cloud_sky/clouds.glsl, cloud_sky/sky-lut.glsl, cloud_sky/transmittance-lut.glsl, cloud_sky/clouds.gdshader), so that their TEXT -- read from
 * /root/reference at run time by make_glsl_fixtures.py, never stored in this repository -- compiles with g++ and runs on the CPU.
 * The purpose: take the two hand restatements (oracle/cloudsky_oracle.c, oracle/numpy_restatement.py) out of the trusted base.
 * What stays builder-defined, and is therefore NOT pinned by this exercise, is exactly what this header defines:
 *
 *   - the built-in functions: mix = a*(1-t)+b*t, smoothstep = t*t*(3-2t) on clamp((x-e0)/(e1-e0),0,1), clamp = min(max(x,lo),hi),
 *     fract = x-floor(x), length = sqrt(x*x+y*y+..) summed left to right, normalize = v/length(v), dot summed left to right,
 *     mat*vec = column sum left to right; pow/exp/log/sin/cos/asin/atan/sqrt = glibc's float functions (a GPU uses its own);
 *   - texture()/textureLod()/imageStore(): bound to the C oracle's sampler and fp16-store functions (csko_tap_*, csko_f2h);
 *   - no FMA contraction (-ffp-contract=off), IEEE fp32 with round-to-nearest for every run-time operation.
 *
 * Constant folding.  glslang (Godot's GLSL front end) evaluates constant expressions -- literals, `const` variables initialised
 * from constant expressions, and operators / constructors / built-ins applied to them -- at compile time in DOUBLE and narrows
 * the result to float where it meets a run-time value [recalled: glslang Constant.cpp folds on TConstUnion::dConst].  The shim
 * models that with a per-value flag `k` ("is a constant expression"): k-flagged operands combine in double and stay flagged,
 * anything else is narrowed to float first and combines in float (a built-in call folds only if ALL its arguments are flagged).  "Is a constant expression" is decided the way the language
 * does, from the expression's form, which C++ exposes as its value category and constness:
 *     literal                      -> flagged
 *     const-qualified variable     -> keeps the flag of its initialiser      (C++: const lvalue)
 *     non-const variable/parameter -> NEVER constant, whatever was stored    (C++: non-const lvalue: flag dropped on read)
 *     temporary                    -> keeps its flag                         (C++: rvalue)
 * GX_FOLD_DOUBLE=0 turns the model off: every literal is narrowed to float at once and nothing ever folds in double (build that
 * variant with -fsingle-precision-constant so literal-op-literal is float too).  The two variants bracket what a GLSL compiler
 * may legitimately do with constants; fixtures are written for both.
 */
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <utility>

#include "../cloudsky_oracle.h"

#ifndef GX_FOLD_DOUBLE
#define GX_FOLD_DOUBLE 1
#endif

namespace gx {

/* ------------------------------------------------------------------------------------------------ scalar */
struct F { /* GLSL `float` */
    double d;
    bool k;
    F() = default;
    F(double v) { if (GX_FOLD_DOUBLE) { d = v; k = true; } else { d = (double)(float)v; k = false; } } /* literal */
    F(float v) : d((double)v), k(false) {} /* run-time value (driver inputs; with -fsingle-precision-constant: literals) */
    F(int v) : d((double)(float)v), k(false) {}
    F(unsigned v) : d((double)(float)v), k(false) {}
    static F raw(double v, bool kk) { F r; r.d = v; r.k = kk; return r; }
    float f() const { return (float)d; }
    explicit operator int() const { return (int)(float)d; }
    explicit operator float() const { return (float)d; }
};
inline F narrow(F a) { return F::raw((double)(float)a.d, false); }

#define GX_S2(name, opf, opd)                                                    \
    inline F name(F a, F b) {                                                    \
        if (GX_FOLD_DOUBLE && a.k && b.k) return F::raw(opd, true);              \
        float x = (float)a.d, y = (float)b.d; (void)x; (void)y;                  \
        return F::raw((double)(float)(opf), false);                              \
    }
GX_S2(s_add, x + y, a.d + b.d)
GX_S2(s_sub, x - y, a.d - b.d)
GX_S2(s_mul, x * y, a.d * b.d)
GX_S2(s_div, x / y, a.d / b.d)
GX_S2(s_pow, ::powf(x, y), ::pow(a.d, b.d))
GX_S2(s_atan2, ::atan2f(x, y), ::atan2(a.d, b.d))
GX_S2(s_max, (x < y ? y : x), (a.d < b.d ? b.d : a.d)) /* GLSL: max(x,y) = x < y ? y : x */
GX_S2(s_min, (y < x ? y : x), (b.d < a.d ? b.d : a.d)) /* GLSL: min(x,y) = y < x ? y : x */
#undef GX_S2
#define GX_S1(name, opf, opd)                                                    \
    inline F name(F a) {                                                         \
        if (GX_FOLD_DOUBLE && a.k) return F::raw(opd, true);                     \
        float x = (float)a.d;                                                    \
        return F::raw((double)(float)(opf), false);                              \
    }
GX_S1(s_neg, -x, -a.d)
GX_S1(s_sqrt, ::sqrtf(x), ::sqrt(a.d))
GX_S1(s_exp, ::expf(x), ::exp(a.d))
GX_S1(s_log, ::logf(x), ::log(a.d))
GX_S1(s_sin, ::sinf(x), ::sin(a.d))
GX_S1(s_cos, ::cosf(x), ::cos(a.d))
GX_S1(s_asin, ::asinf(x), ::asin(a.d))
GX_S1(s_abs, ::fabsf(x), ::fabs(a.d))
GX_S1(s_floor, ::floorf(x), ::floor(a.d))
GX_S1(s_sign, (x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f)), (a.d > 0.0 ? 1.0 : (a.d < 0.0 ? -1.0 : 0.0)))
#undef GX_S1
inline int s_cmp_lt(F a, F b) { return (GX_FOLD_DOUBLE && a.k && b.k) ? a.d < b.d : (float)a.d < (float)b.d; }
inline int s_cmp_le(F a, F b) { return (GX_FOLD_DOUBLE && a.k && b.k) ? a.d <= b.d : (float)a.d <= (float)b.d; }

/* ------------------------------------------------------------------------------------------------ vectors and swizzles */
struct vec2; struct vec3; struct vec4; struct ivec2;
template <int N> struct vec_of;
template <> struct vec_of<0> { typedef F type; };
template <> struct vec_of<2> { typedef vec2 type; };
template <> struct vec_of<3> { typedef vec3 type; };
template <> struct vec_of<4> { typedef vec4 type; };

/* swizzle proxy: K pointers into the owner's components (K == 1: a scalar view such as .r) */
template <int K> struct swz {
    F *p[K];
    swz() {}
    swz(const swz &) = delete; /* proxies never move between owners; the owner rebuilds them */
    template <class B> swz &operator=(B &&b);
    swz &operator=(const swz &b);
    template <int KK = K, class = std::enable_if_t<KK == 1>> operator F() const { return *p[0]; } /* .r passed as a float argument */
};

template <class T> struct tr { static constexpr int dim = -1; static constexpr bool gx = false; static constexpr int kind = 0; };
template <> struct tr<F> { static constexpr int dim = 0; static constexpr bool gx = true; static constexpr int kind = 1; };
template <> struct tr<double> { static constexpr int dim = 0; static constexpr bool gx = false; static constexpr int kind = 2; };
template <> struct tr<float> { static constexpr int dim = 0; static constexpr bool gx = false; static constexpr int kind = 2; };
template <> struct tr<int> { static constexpr int dim = 0; static constexpr bool gx = false; static constexpr int kind = 2; };
template <> struct tr<unsigned> { static constexpr int dim = 0; static constexpr bool gx = false; static constexpr int kind = 2; };
template <> struct tr<vec2> { static constexpr int dim = 2; static constexpr bool gx = true; static constexpr int kind = 3; };
template <> struct tr<vec3> { static constexpr int dim = 3; static constexpr bool gx = true; static constexpr int kind = 3; };
template <> struct tr<vec4> { static constexpr int dim = 4; static constexpr bool gx = true; static constexpr int kind = 3; };
template <int K> struct tr<swz<K>> { static constexpr int dim = (K == 1 ? 0 : K); static constexpr bool gx = true; static constexpr int kind = 4; };
template <class A> using dec = std::decay_t<A>;
template <class A> constexpr int dim_v = tr<dec<A>>::dim;
/* a non-const lvalue is a run-time variable: never a constant expression */
template <class A> constexpr bool drop_v = std::is_lvalue_reference<A>::value && !std::is_const<std::remove_reference_t<A>>::value;

struct vec2 {
    F x, y;
    swz<1> r, g;
    swz<2> xy, yx, rg;
    void bind() { r.p[0] = &x; g.p[0] = &y; xy.p[0] = &x; xy.p[1] = &y; yx.p[0] = &y; yx.p[1] = &x; rg.p[0] = &x; rg.p[1] = &y; }
    F &at(int i) { return i == 0 ? x : y; }
    const F &at(int i) const { return i == 0 ? x : y; }
    vec2() { bind(); }
    vec2(const vec2 &o) : x(o.x), y(o.y) { bind(); }
    vec2 &operator=(const vec2 &o) { x = o.x; y = o.y; return *this; }
    explicit vec2(const ivec2 &p); /* vec2(ivec2): int -> float conversion per component */
    template <class A, class = std::enable_if_t<dim_v<A> == 0>> explicit vec2(A &&a);
    template <class A, class B, class = std::enable_if_t<dim_v<A> == 0 && dim_v<B> == 0>> vec2(A &&a, B &&b);
    template <class A, class = std::enable_if_t<dim_v<A> == 2 && tr<dec<A>>::kind == 4>, class = void> vec2(A &&a);
    template <class B, class = std::enable_if_t<(dim_v<B> >= 0)>> vec2 &operator=(B &&b);
};
struct vec3 {
    F x, y, z;
    swz<1> r, g, b;
    swz<2> xy, xz, yx, yz, zx, zy;
    swz<3> xyz, xzy, yxz, yzx, zxy, zyx, rgb;
    void bind() {
        r.p[0] = &x; g.p[0] = &y; b.p[0] = &z;
        F *c[3] = {&x, &y, &z};
#define GX_B2(n, i, j) n.p[0] = c[i]; n.p[1] = c[j];
#define GX_B3(n, i, j, l) n.p[0] = c[i]; n.p[1] = c[j]; n.p[2] = c[l];
        GX_B2(xy, 0, 1) GX_B2(xz, 0, 2) GX_B2(yx, 1, 0) GX_B2(yz, 1, 2) GX_B2(zx, 2, 0) GX_B2(zy, 2, 1)
        GX_B3(xyz, 0, 1, 2) GX_B3(xzy, 0, 2, 1) GX_B3(yxz, 1, 0, 2) GX_B3(yzx, 1, 2, 0) GX_B3(zxy, 2, 0, 1) GX_B3(zyx, 2, 1, 0) GX_B3(rgb, 0, 1, 2)
    }
    F &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
    const F &at(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    vec3() { bind(); }
    vec3(const vec3 &o) : x(o.x), y(o.y), z(o.z) { bind(); }
    vec3 &operator=(const vec3 &o) { x = o.x; y = o.y; z = o.z; return *this; }
    template <class A, class = std::enable_if_t<dim_v<A> == 0>> explicit vec3(A &&a);
    template <class A, class B, class C, class = std::enable_if_t<dim_v<A> == 0 && dim_v<B> == 0 && dim_v<C> == 0>> vec3(A &&a, B &&b, C &&c);
    template <class A, class = std::enable_if_t<dim_v<A> == 3 && tr<dec<A>>::kind == 4>, class = void> vec3(A &&a);
    template <class B, class = std::enable_if_t<(dim_v<B> >= 0)>> vec3 &operator=(B &&b);
};
struct vec4 {
    F x, y, z, w;
    swz<1> r, g, b, a;
    swz<2> xy, xz, zw;
    swz<3> xyz, rgb;
    void bind() {
        r.p[0] = &x; g.p[0] = &y; b.p[0] = &z; a.p[0] = &w;
        F *c[4] = {&x, &y, &z, &w};
        GX_B2(xy, 0, 1) GX_B2(xz, 0, 2) GX_B2(zw, 2, 3) GX_B3(xyz, 0, 1, 2) GX_B3(rgb, 0, 1, 2)
#undef GX_B2
#undef GX_B3
    }
    F &at(int i) { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    const F &at(int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    vec4() { bind(); }
    vec4(const vec4 &o) : x(o.x), y(o.y), z(o.z), w(o.w) { bind(); }
    vec4 &operator=(const vec4 &o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
    template <class A, class = std::enable_if_t<dim_v<A> == 0>> explicit vec4(A &&a);
    template <class A, class B, class C, class D, class = std::enable_if_t<dim_v<A> == 0 && dim_v<B> == 0 && dim_v<C> == 0 && dim_v<D> == 0>>
    vec4(A &&a, B &&b, C &&c, D &&d);
    template <class A, class B, class = std::enable_if_t<dim_v<A> == 3 && dim_v<B> == 0>> vec4(A &&a, B &&b); /* vec4(vec3, float) */
    template <class A, class = std::enable_if_t<dim_v<A> == 4 && tr<dec<A>>::kind == 4>, class = void> vec4(A &&a);
    template <class B, class = std::enable_if_t<(dim_v<B> >= 0)>> vec4 &operator=(B &&b);
};

/* component i of any participating operand, with the constant-expression rule applied (scalars broadcast) */
template <class A> inline F comp(A &&a, int i) {
    typedef dec<A> T;
    constexpr int kind = tr<T>::kind;
    static_assert(kind != 0, "type does not take part in GLSL arithmetic");
    if constexpr (kind == 2) { (void)i; return F(a); }
    else if constexpr (kind == 1) { (void)i; return drop_v<A> ? narrow(a) : F(a); }
    else if constexpr (kind == 3) { return drop_v<A> ? narrow(a.at(i)) : a.at(i); }
    else { const F &v = *a.p[tr<T>::dim == 0 ? 0 : i]; return drop_v<A> ? narrow(v) : v; }
}
template <int N, class Fn> inline typename vec_of<N>::type build(Fn fn) {
    if constexpr (N == 0) return fn(0);
    else { typename vec_of<N>::type r; for (int i = 0; i < N; i++) r.at(i) = fn(i); return r; }
}

template <class A, class> vec2::vec2(A &&a) { bind(); x = y = comp(std::forward<A>(a), 0); }
template <class A, class B, class> vec2::vec2(A &&a, B &&b) { bind(); x = comp(std::forward<A>(a), 0); y = comp(std::forward<B>(b), 0); }
template <class A, class, class> vec2::vec2(A &&a) { bind(); for (int i = 0; i < 2; i++) at(i) = comp(std::forward<A>(a), i); }
template <class A, class> vec3::vec3(A &&a) { bind(); x = y = z = comp(std::forward<A>(a), 0); }
template <class A, class B, class C, class> vec3::vec3(A &&a, B &&b, C &&c) {
    bind(); x = comp(std::forward<A>(a), 0); y = comp(std::forward<B>(b), 0); z = comp(std::forward<C>(c), 0);
}
template <class A, class, class> vec3::vec3(A &&a) { bind(); for (int i = 0; i < 3; i++) at(i) = comp(std::forward<A>(a), i); }
template <class A, class> vec4::vec4(A &&a) { bind(); x = y = z = w = comp(std::forward<A>(a), 0); }
template <class A, class B, class C, class D, class> vec4::vec4(A &&a, B &&b, C &&c, D &&d) {
    bind(); x = comp(std::forward<A>(a), 0); y = comp(std::forward<B>(b), 0); z = comp(std::forward<C>(c), 0); w = comp(std::forward<D>(d), 0);
}
template <class A, class B, class> vec4::vec4(A &&a, B &&b) { bind(); for (int i = 0; i < 3; i++) at(i) = comp(std::forward<A>(a), i); w = comp(std::forward<B>(b), 0); }
template <class A, class, class> vec4::vec4(A &&a) { bind(); for (int i = 0; i < 4; i++) at(i) = comp(std::forward<A>(a), i); }
/* assignment from a proxy / scalar-free vector expression: evaluate all components first (p.xy = p.yx must not alias) */
#define GX_ASSIGN(V, N)                                                                                   \
    template <class B, class> V &V::operator=(B &&b) {                                                     \
        static_assert(dim_v<B> == N, "vector size mismatch in assignment");                                \
        F t[N]; for (int i = 0; i < N; i++) t[i] = comp(std::forward<B>(b), i);                            \
        for (int i = 0; i < N; i++) at(i) = t[i]; return *this;                                            \
    }
GX_ASSIGN(vec2, 2) GX_ASSIGN(vec3, 3) GX_ASSIGN(vec4, 4)
#undef GX_ASSIGN
template <int K> template <class B> swz<K> &swz<K>::operator=(B &&b) {
    static_assert(dim_v<B> == (K == 1 ? 0 : K), "vector size mismatch in swizzle assignment");
    F t[K]; for (int i = 0; i < K; i++) t[i] = comp(std::forward<B>(b), i);
    for (int i = 0; i < K; i++) *p[i] = t[i]; return *this;
}
template <int K> swz<K> &swz<K>::operator=(const swz<K> &b) {
    F t[K]; for (int i = 0; i < K; i++) t[i] = *b.p[i];
    for (int i = 0; i < K; i++) *p[i] = t[i]; return *this;
}

/* result dimension of a component-wise operation; -1 = these operands are none of our business */
template <class A, class B> struct bdim {
    static constexpr int a = dim_v<A>, b = dim_v<B>;
    static constexpr bool ok = (tr<dec<A>>::gx || tr<dec<B>>::gx) && a >= 0 && b >= 0 && (a == 0 || b == 0 || a == b);
    static constexpr int value = ok ? (a > b ? a : b) : -1;
};
template <class A, class B, class C> struct tdim {
    static constexpr int ab = bdim<A, B>::value >= 0 ? bdim<A, B>::value : (dim_v<A> == 0 && dim_v<B> == 0 ? 0 : -1);
    static constexpr int c = dim_v<C>;
    static constexpr bool any = tr<dec<A>>::gx || tr<dec<B>>::gx || tr<dec<C>>::gx;
    static constexpr bool ok = any && ab >= 0 && c >= 0 && (ab == 0 || c == 0 || ab == c);
    static constexpr int value = ok ? (ab > c ? ab : c) : -1;
};

#define GX_BINOP(op, fn)                                                                                               \
    template <class A, class B, int N = bdim<A, B>::value, class = std::enable_if_t<(N >= 0)>>                         \
    inline typename vec_of<N>::type operator op(A &&a, B &&b) {                                                        \
        return build<N>([&](int i) { return fn(comp(std::forward<A>(a), i), comp(std::forward<B>(b), i)); });          \
    }
GX_BINOP(+, s_add) GX_BINOP(-, s_sub) GX_BINOP(*, s_mul) GX_BINOP(/, s_div)
#undef GX_BINOP
#define GX_CMP(op, expr)                                                                                               \
    template <class A, class B, int N = bdim<A, B>::value, class = std::enable_if_t<(N == 0)>>                         \
    inline bool operator op(A &&a, B &&b) { F x = comp(std::forward<A>(a), 0), y = comp(std::forward<B>(b), 0); return expr; }
GX_CMP(<, s_cmp_lt(x, y)) GX_CMP(>, s_cmp_lt(y, x)) GX_CMP(<=, s_cmp_le(x, y)) GX_CMP(>=, s_cmp_le(y, x))
#undef GX_CMP
/* compound assignment: the left side is a variable (or a swizzle of one), so it is read as a run-time value */
#define GX_CASSIGN(op, fn)                                                                                             \
    template <class A, class B, int N = bdim<A &, B>::value, class = std::enable_if_t<(N >= 0 && tr<dec<A>>::gx)>>     \
    inline A &operator op(A &a, B &&b) {                                                                               \
        static_assert(dim_v<A> == N, "compound assignment cannot widen its left side");                                \
        a = build<N>([&](int i) { return fn(comp(a, i), comp(std::forward<B>(b), i)); });                              \
        return a;                                                                                                      \
    }
GX_CASSIGN(+=, s_add) GX_CASSIGN(-=, s_sub) GX_CASSIGN(*=, s_mul) GX_CASSIGN(/=, s_div)
#undef GX_CASSIGN
template <class A, int N = dim_v<A>, class = std::enable_if_t<(N >= 0 && tr<dec<A>>::gx)>>
inline typename vec_of<N>::type operator-(A &&a) { return build<N>([&](int i) { return s_neg(comp(std::forward<A>(a), i)); }); }

/* ------------------------------------------------------------------------------------------------ built-in functions */
#define GX_FN1(name, fn)                                                                                               \
    template <class A, int N = dim_v<A>, class = std::enable_if_t<(N >= 0 && tr<dec<A>>::gx)>>                         \
    inline typename vec_of<N>::type name(A &&a) { return build<N>([&](int i) { return fn(comp(std::forward<A>(a), i)); }); }
GX_FN1(sqrt, s_sqrt) GX_FN1(exp, s_exp) GX_FN1(log, s_log) GX_FN1(sin, s_sin) GX_FN1(cos, s_cos) GX_FN1(asin, s_asin)
GX_FN1(abs, s_abs) GX_FN1(floor, s_floor) GX_FN1(sign, s_sign)
#undef GX_FN1
#define GX_FN2(name, fn)                                                                                               \
    template <class A, class B, int N = bdim<A, B>::value, class = std::enable_if_t<(N >= 0)>>                         \
    inline typename vec_of<N>::type name(A &&a, B &&b) {                                                               \
        return build<N>([&](int i) { return fn(comp(std::forward<A>(a), i), comp(std::forward<B>(b), i)); });          \
    }
GX_FN2(pow, s_pow) GX_FN2(atan, s_atan2) GX_FN2(max, s_max) GX_FN2(min, s_min)
#undef GX_FN2
/* A built-in is folded only when ALL of its arguments are constant expressions (then glslang evaluates the whole call in double); otherwise the GPU runs
 * it, in float, on the narrowed arguments: smoothstep(0.6, 1.0, x) computes 1.0f - 0.6f = 0.39999998 at run time, not the double 0.4 (found by executing
 * clouds.gdshader: 3 halfs of a panorama moved by 1 ulp while this header still folded `e1 - e0` inside the call). */
inline void unfold3(F &a, F &b, F &c) { if (!(a.k && b.k && c.k)) { a = narrow(a); b = narrow(b); c = narrow(c); } }
inline F s_fract(F a) { return s_sub(a, s_floor(a)); }
inline F s_clamp(F x, F lo, F hi) { unfold3(x, lo, hi); return s_min(s_max(x, lo), hi); }
inline F s_mix(F a, F b, F t) { unfold3(a, b, t); const F one = a.k ? F(1.0) : F(1.0f); return s_add(s_mul(a, s_sub(one, t)), s_mul(b, t)); }
inline F s_smoothstep(F e0, F e1, F x) {
    unfold3(e0, e1, x);
    const bool k = e0.k;
    F t = s_div(s_sub(x, e0), s_sub(e1, e0));
    t = s_min(s_max(t, k ? F(0.0) : F(0.0f)), k ? F(1.0) : F(1.0f));
    return s_mul(s_mul(t, t), s_sub(k ? F(3.0) : F(3.0f), s_mul(k ? F(2.0) : F(2.0f), t)));
}
template <class A, int N = dim_v<A>, class = std::enable_if_t<(N >= 0 && tr<dec<A>>::gx)>>
inline typename vec_of<N>::type fract(A &&a) { return build<N>([&](int i) { return s_fract(comp(std::forward<A>(a), i)); }); }
#define GX_FN3(name, fn)                                                                                               \
    template <class A, class B, class C, int N = tdim<A, B, C>::value, class = std::enable_if_t<(N >= 0)>>             \
    inline typename vec_of<N>::type name(A &&a, B &&b, C &&c) {                                                        \
        return build<N>([&](int i) { return fn(comp(std::forward<A>(a), i), comp(std::forward<B>(b), i), comp(std::forward<C>(c), i)); }); \
    }
GX_FN3(clamp, s_clamp) GX_FN3(mix, s_mix) GX_FN3(smoothstep, s_smoothstep)
#undef GX_FN3
/* the components of a vector argument, with the all-or-nothing rule: one run-time component makes the whole call a run-time call */
template <int N, class A> inline void load_vec(A &&a, F (&v)[N], bool &allk) { for (int i = 0; i < N; i++) { v[i] = comp(std::forward<A>(a), i); allk = allk && v[i].k; } }
template <int N> inline void unfold(F (&v)[N], bool allk) { if (!allk) for (int i = 0; i < N; i++) v[i] = narrow(v[i]); }
template <class A, class B, int N = bdim<A, B>::value, class = std::enable_if_t<(N >= 2 && dim_v<A> == dim_v<B>)>>
inline F dot(A &&a, B &&b) {
    F x[N], y[N]; bool k = true;
    load_vec<N>(std::forward<A>(a), x, k); load_vec<N>(std::forward<B>(b), y, k); unfold<N>(x, k); unfold<N>(y, k);
    F s = s_mul(x[0], y[0]);
    for (int i = 1; i < N; i++) s = s_add(s, s_mul(x[i], y[i]));
    return s;
}
template <int N> inline F length_of(const F (&x)[N]) {
    F s = s_mul(x[0], x[0]);
    for (int i = 1; i < N; i++) s = s_add(s, s_mul(x[i], x[i]));
    return s_sqrt(s);
}
template <class A, int N = dim_v<A>, class = std::enable_if_t<(N >= 2)>>
inline F length(A &&a) { F x[N]; bool k = true; load_vec<N>(std::forward<A>(a), x, k); unfold<N>(x, k); return length_of<N>(x); }
template <class A, int N = dim_v<A>, class = std::enable_if_t<(N >= 2)>>
inline typename vec_of<N>::type normalize(A &&a) {
    F x[N]; bool k = true; load_vec<N>(std::forward<A>(a), x, k); unfold<N>(x, k);
    const F l = length_of<N>(x);
    return build<N>([&](int i) { return s_div(x[i], l); });
}

/* ------------------------------------------------------------------------------------------------ integer vectors, matrix */
struct uvec2 { unsigned x, y; };
struct uvec3 { union { struct { unsigned x, y, z; }; uvec2 xy; }; };
struct ivec2 {
    int x, y;
    ivec2() {}
    ivec2(int a, int b) : x(a), y(b) {}
    explicit ivec2(const uvec2 &u) : x((int)u.x), y((int)u.y) {}
    template <class A, class = std::enable_if_t<dim_v<A> == 2>> explicit ivec2(A &&a) : x((int)comp(std::forward<A>(a), 0)), y((int)comp(std::forward<A>(a), 1)) {}
};
inline ivec2 operator+(const ivec2 &a, const ivec2 &b) { return ivec2(a.x + b.x, a.y + b.y); }
inline vec2::vec2(const ivec2 &p) : x(F(p.x)), y(F(p.y)) { bind(); }
struct mat4x3 { /* 4 columns of 3 rows, constructor arguments in column-major order (GLSL 4.50 section 5.4.2) */
    F c[4][3];
    template <class... A, class = std::enable_if_t<sizeof...(A) == 12>> mat4x3(A &&...a) {
        F t[12] = {comp(std::forward<A>(a), 0)...};
        for (int i = 0; i < 12; i++) c[i / 3][i % 3] = t[i];
    }
};
template <class V, class = std::enable_if_t<dim_v<V> == 4>> inline vec3 operator*(const mat4x3 &m, V &&v) {
    vec3 r;
    for (int row = 0; row < 3; row++) {
        F s = s_mul(m.c[0][row], comp(std::forward<V>(v), 0));
        for (int col = 1; col < 4; col++) s = s_add(s, s_mul(m.c[col][row], comp(std::forward<V>(v), col)));
        r.at(row) = s;
    }
    return r;
}

inline uvec3 gl_GlobalInvocationID; /* set by the drivers before each main() */

/* ------------------------------------------------------------------------------------------------ opaque types */
struct sampler3D { const uint8_t *chain; int n0, levels, ch; };               /* REPEAT, LINEAR, integer LOD (cloud_sky.gd:301-309) */
struct sampler2D { int kind; const void *data; int w, h; };                    /* kind 0: weather RGB8 512^2 REPEAT; 1: RGBA16F CLAMP */
struct image2D { uint16_t *out; int w, h; int x0, y0; size_t pitch_halfs; }; /* rgba16f, stores outside [x0,x0+w)x[y0,y0+h) discarded */

template <class P> inline vec4 textureLod(const sampler3D &s, P &&pnt, F lod) {
    vec3 q(std::forward<P>(pnt));
    float c[3] = {q.x.f(), q.y.f(), q.z.f()}, o[4];
    csko_tap3d_repeat(s.chain, s.n0, s.levels, s.ch, lod.f(), c, o);
    return vec4(F(o[0]), F(o[1]), F(o[2]), F(o[3]));
}
inline vec4 texture(const sampler2D &s, const vec2 &uv);
inline vec4 textureLod(const sampler2D &s, const vec2 &uv, F) { return texture(s, uv); }   /* single-level 2-D images: every LOD is level 0 */
inline vec4 texture(const sampler2D &s, const vec2 &uv) {
    float o[4] = {0, 0, 0, 1};
    if (s.kind == 0) csko_tap_weather((const uint8_t *)s.data, uv.x.f(), uv.y.f(), o);
    else csko_tap_rgba16f_clamp((const uint16_t *)s.data, s.w, s.h, uv.x.f(), uv.y.f(), o);
    return vec4(F(o[0]), F(o[1]), F(o[2]), F(o[3]));
}
inline void imageStore(const image2D &im, const ivec2 &p, const vec4 &v) {
    int lx = p.x - im.x0, ly = p.y - im.y0;
    if (lx < 0 || ly < 0 || lx >= im.w || ly >= im.h) return; /* Vulkan discards out-of-bounds image stores */
    uint16_t *o = im.out + (size_t)ly * im.pitch_halfs + (size_t)lx * 4;
    o[0] = csko_f2h(v.x.f()); o[1] = csko_f2h(v.y.f()); o[2] = csko_f2h(v.z.f()); o[3] = csko_f2h(v.w.f());
}

} /* namespace gx */
