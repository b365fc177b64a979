// TEST INFRASTRUCTURE -- part of the oracle/_ref recipe (see oracle/ref_glue/README.md). Not linked into the product.
//
// A GLSL front-end in a header: the vector / matrix types, swizzles, constructors and built-in functions the
// reference's shaders use, so that the reference's OWN shader sources (mechanically rewritten by glsl2cpp.py:
// parameter qualifiers, float literal suffixes, layout declarations -- nothing else) compile as C++20.
//
// GLSL leaves the evaluation order and accuracy of its built-ins to the implementation ("the driver").  This
// header plays the driver and fixes them -- written from the GLSL 4.60 specification, independently of
// oracle/glsl_math.h, with the same stated conventions so the two can be compared bit for bit:
//   dot(a,b) = ((a.x*b.x + a.y*b.y) + a.z*b.z) [+ a.w*b.w]     normalize(v) = v * (1 / sqrt(dot(v,v)))
//   mix(a,b,t) = a*(1-t) + b*t          min(a,b) = b<a ? b : a          max(a,b) = a<b ? b : a
//   M*v = ((c0*v.x + c1*v.y) + c2*v.z) + c3*v.w                  v*M = (dot(v,c0), dot(v,c1), ...)
//   transcendental functions = include/pt_fpmath.h (the fp32 contract); fp32 throughout, no contraction.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include "../../include/pt_fpmath.h"

namespace glslc {

typedef unsigned int uint;

template <class T> struct tvec2;
template <class T> struct tvec3;
template <class T> struct tvec4;
struct vec2;
struct vec3;
struct vec4;

// ---- swizzle proxies: live inside the anonymous union of their parent vector (P = parent's component count) ----
template <int P, int A, int B> struct swz2 {
  float d[P];
  inline operator vec2() const;
  inline swz2& operator=(const vec2& v);
  inline swz2& operator=(const swz2& o);
};
template <int P, int A, int B, int C> struct swz3 {
  float d[P];
  inline operator vec3() const;
  inline swz3& operator=(const vec3& v);
  inline swz3& operator=(const swz3& o);
};
template <int P, int A, int B, int C, int D> struct swz4 {
  float d[P];
  inline operator vec4() const;
};

struct vec2 {
  union {
    struct { float x, y; };
    struct { float r, g; };
    float d[2];
    swz2<2, 0, 1> xy;
  };
  vec2() {}
  vec2(const vec2& o) : x(o.x), y(o.y) {}
  vec2& operator=(const vec2& o) { x = o.x; y = o.y; return *this; }
  vec2(float a, float b) : x(a), y(b) {}
  template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> vec2(S a) : x(float(a)), y(float(a)) {}
  template <class S, class U, class = std::enable_if_t<std::is_arithmetic<S>::value && std::is_arithmetic<U>::value>> vec2(S a, U b) : x(float(a)), y(float(b)) {}
  template <class T> explicit vec2(const tvec2<T>& v);
  float&       operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
};
struct vec3 {
  union {
    struct { float x, y, z; };
    struct { float r, g, b; };
    float d[3];
    swz2<3, 0, 1>    xy;
    swz3<3, 0, 1, 2> xyz;
    swz3<3, 0, 1, 2> rgb;
  };
  vec3() {}
  vec3(const vec3& o) : x(o.x), y(o.y), z(o.z) {}
  vec3& operator=(const vec3& o) { x = o.x; y = o.y; z = o.z; return *this; }
  template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> vec3(S a) : x(float(a)), y(float(a)), z(float(a)) {}
  template <class S, class U, class V, class = std::enable_if_t<std::is_arithmetic<S>::value && std::is_arithmetic<U>::value && std::is_arithmetic<V>::value>>
  vec3(S a, U b, V c) : x(float(a)), y(float(b)), z(float(c)) {}
  template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> vec3(const vec2& v, S c) : x(v.x), y(v.y), z(float(c)) {}
  explicit inline vec3(const vec4& v);  // GLSL: drops w
  template <class T> explicit vec3(const tvec3<T>& v);
  float&       operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
};
struct vec4 {
  union {
    struct { float x, y, z, w; };
    struct { float r, g, b, a; };
    float d[4];
    swz2<4, 0, 1>       xy;
    swz3<4, 0, 1, 2>    xyz;
    swz3<4, 0, 1, 2>    rgb;
    swz4<4, 0, 1, 2, 3> rgba;
  };
  vec4() {}
  vec4(const vec4& o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
  vec4& operator=(const vec4& o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
  template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> vec4(S s) : x(float(s)), y(float(s)), z(float(s)), w(float(s)) {}
  template <class S, class U, class V, class W,
            class = std::enable_if_t<std::is_arithmetic<S>::value && std::is_arithmetic<U>::value && std::is_arithmetic<V>::value && std::is_arithmetic<W>::value>>
  vec4(S a, U b, V c, W e) : x(float(a)), y(float(b)), z(float(c)), w(float(e)) {}
  template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> vec4(const vec3& v, S e) : x(v.x), y(v.y), z(v.z), w(float(e)) {}
  template <class S, class U, class = std::enable_if_t<std::is_arithmetic<S>::value && std::is_arithmetic<U>::value>>
  vec4(const vec2& v, S c, U e) : x(v.x), y(v.y), z(float(c)), w(float(e)) {}
  float&       operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
};
inline vec3::vec3(const vec4& v) : x(v.x), y(v.y), z(v.z) {}

template <int P, int A, int B> inline swz2<P, A, B>::operator vec2() const { return vec2(d[A], d[B]); }
template <int P, int A, int B> inline swz2<P, A, B>& swz2<P, A, B>::operator=(const vec2& v) { d[A] = v.x; d[B] = v.y; return *this; }
template <int P, int A, int B> inline swz2<P, A, B>& swz2<P, A, B>::operator=(const swz2& o) { return *this = vec2(o); }
template <int P, int A, int B, int C> inline swz3<P, A, B, C>::operator vec3() const { return vec3(d[A], d[B], d[C]); }
template <int P, int A, int B, int C> inline swz3<P, A, B, C>& swz3<P, A, B, C>::operator=(const vec3& v) { d[A] = v.x; d[B] = v.y; d[C] = v.z; return *this; }
template <int P, int A, int B, int C> inline swz3<P, A, B, C>& swz3<P, A, B, C>::operator=(const swz3& o) { return *this = vec3(o); }
template <int P, int A, int B, int C, int D> inline swz4<P, A, B, C, D>::operator vec4() const { return vec4(d[A], d[B], d[C], d[D]); }

// ---- integer / boolean vectors -------------------------------------------------------------------------------
template <class T> struct tvec2 {
  T x, y;
  tvec2() {}
  tvec2(T a, T b) : x(a), y(b) {}
  template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> tvec2(S a) : x(T(a)), y(T(a)) {}
  template <class S, class U, class = std::enable_if_t<std::is_arithmetic<S>::value && std::is_arithmetic<U>::value>> tvec2(S a, U b) : x(T(a)), y(T(b)) {}
  template <class U> tvec2(const tvec2<U>& o) : x(T(o.x)), y(T(o.y)) {}  // GLSL implicit int -> uint conversions, explicit constructors
  explicit tvec2(const vec2& v) : x(T(v.x)), y(T(v.y)) {}
  tvec2<T> xy_() const { return *this; }
};
template <class T> struct tvec3 {
  T x, y, z;
  tvec3() {}
  tvec3(T a, T b, T c) : x(a), y(b), z(c) {}
  template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> tvec3(S a) : x(T(a)), y(T(a)), z(T(a)) {}
  template <class S, class U, class V, class = std::enable_if_t<std::is_arithmetic<S>::value && std::is_arithmetic<U>::value && std::is_arithmetic<V>::value>>
  tvec3(S a, U b, V c) : x(T(a)), y(T(b)), z(T(c)) {}
  template <class U> tvec3(const tvec3<U>& o) : x(T(o.x)), y(T(o.y)), z(T(o.z)) {}
  template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> tvec3(const vec2& v, S c) : x(T(v.x)), y(T(v.y)), z(T(c)) {}
  explicit tvec3(const vec3& v) : x(T(v.x)), y(T(v.y)), z(T(v.z)) {}
};
typedef tvec2<int>  ivec2;
typedef tvec3<int>  ivec3;
typedef tvec2<uint> uvec2;
typedef tvec3<uint> uvec3;
typedef tvec3<bool> bvec3;
template <class T> inline vec2::vec2(const tvec2<T>& v) : x(float(v.x)), y(float(v.y)) {}
template <class T> inline vec3::vec3(const tvec3<T>& v) : x(float(v.x)), y(float(v.y)), z(float(v.z)) {}

#define GLSLC_IVEC_OP(op)                                                                                                                     \
  template <class T> inline tvec2<T> operator op(tvec2<T> a, tvec2<T> b) { return tvec2<T>(T(a.x op b.x), T(a.y op b.y)); }                   \
  template <class T> inline tvec3<T> operator op(tvec3<T> a, tvec3<T> b) { return tvec3<T>(T(a.x op b.x), T(a.y op b.y), T(a.z op b.z)); }    \
  template <class T, class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> inline tvec2<T> operator op(tvec2<T> a, S s) { return tvec2<T>(T(a.x op T(s)), T(a.y op T(s))); } \
  template <class T, class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> inline tvec3<T> operator op(tvec3<T> a, S s) { return tvec3<T>(T(a.x op T(s)), T(a.y op T(s)), T(a.z op T(s))); } \
  template <class T, class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> inline tvec2<T> operator op(S s, tvec2<T> a) { return tvec2<T>(T(T(s) op a.x), T(T(s) op a.y)); } \
  template <class T, class S, class = std::enable_if_t<std::is_arithmetic<S>::value>> inline tvec3<T> operator op(S s, tvec3<T> a) { return tvec3<T>(T(T(s) op a.x), T(T(s) op a.y), T(T(s) op a.z)); }
GLSLC_IVEC_OP(+)
GLSLC_IVEC_OP(-)
GLSLC_IVEC_OP(*)
GLSLC_IVEC_OP(/)
GLSLC_IVEC_OP(^)
GLSLC_IVEC_OP(|)
GLSLC_IVEC_OP(&)
GLSLC_IVEC_OP(>>)
GLSLC_IVEC_OP(<<)
#undef GLSLC_IVEC_OP
template <class T> inline tvec2<T>& operator+=(tvec2<T>& a, tvec2<T> b) { a = a + b; return a; }
template <class T> inline tvec3<T>& operator+=(tvec3<T>& a, tvec3<T> b) { a = a + b; return a; }
template <class T> inline tvec3<T>& operator^=(tvec3<T>& a, tvec3<T> b) { a = a ^ b; return a; }
template <class T> inline tvec2<T>& operator^=(tvec2<T>& a, tvec2<T> b) { a = a ^ b; return a; }

// ---- float vector arithmetic (component-wise; concrete overloads so that swizzle proxies convert implicitly) --
#define GLSLC_VEC_OP(V, op, ...)                                                    \
  inline V operator op(const V& a, const V& b) { return V(__VA_ARGS__); }
#define GLSLC_VEC_OPS(op)                                                                                                            \
  inline vec2 operator op(const vec2& a, const vec2& b) { return vec2(a.x op b.x, a.y op b.y); }                                      \
  inline vec3 operator op(const vec3& a, const vec3& b) { return vec3(a.x op b.x, a.y op b.y, a.z op b.z); }                          \
  inline vec4 operator op(const vec4& a, const vec4& b) { return vec4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w); }              \
  inline vec2 operator op(const vec2& a, float s) { return vec2(a.x op s, a.y op s); }                                                \
  inline vec3 operator op(const vec3& a, float s) { return vec3(a.x op s, a.y op s, a.z op s); }                                      \
  inline vec4 operator op(const vec4& a, float s) { return vec4(a.x op s, a.y op s, a.z op s, a.w op s); }                            \
  inline vec2 operator op(float s, const vec2& a) { return vec2(s op a.x, s op a.y); }                                                \
  inline vec3 operator op(float s, const vec3& a) { return vec3(s op a.x, s op a.y, s op a.z); }                                      \
  inline vec4 operator op(float s, const vec4& a) { return vec4(s op a.x, s op a.y, s op a.z, s op a.w); }                            \
  inline vec2& operator op##=(vec2& a, const vec2& b) { a = a op b; return a; }                                                       \
  inline vec3& operator op##=(vec3& a, const vec3& b) { a = a op b; return a; }                                                       \
  inline vec4& operator op##=(vec4& a, const vec4& b) { a = a op b; return a; }                                                       \
  inline vec2& operator op##=(vec2& a, float s) { a = a op s; return a; }                                                             \
  inline vec3& operator op##=(vec3& a, float s) { a = a op s; return a; }                                                             \
  inline vec4& operator op##=(vec4& a, float s) { a = a op s; return a; }
GLSLC_VEC_OPS(+)
GLSLC_VEC_OPS(-)
GLSLC_VEC_OPS(*)
GLSLC_VEC_OPS(/)
#undef GLSLC_VEC_OPS
#undef GLSLC_VEC_OP
inline vec2 operator-(const vec2& a) { return vec2(-a.x, -a.y); }
inline vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
inline vec4 operator-(const vec4& a) { return vec4(-a.x, -a.y, -a.z, -a.w); }
// GLSL converts int / uint operands implicitly (e.g. vec3 * uint)
#define GLSLC_VEC_INT(V)                                                                                                  \
  inline V operator*(const V& a, int s) { return a * float(s); }                                                          \
  inline V operator*(const V& a, uint s) { return a * float(s); }                                                         \
  inline V operator*(int s, const V& a) { return float(s) * a; }                                                          \
  inline V operator/(const V& a, int s) { return a / float(s); }                                                          \
  inline V operator+(const V& a, int s) { return a + float(s); }                                                          \
  inline V operator-(const V& a, int s) { return a - float(s); }                                                          \
  inline V& operator/=(V& a, int s) { a = a / float(s); return a; }                                                        \
  inline V& operator*=(V& a, int s) { a = a * float(s); return a; }
GLSLC_VEC_INT(vec2)
GLSLC_VEC_INT(vec3)
GLSLC_VEC_INT(vec4)
#undef GLSLC_VEC_INT

// ---- scalar built-ins ------------------------------------------------------------------------------------------
template <class A, class B> using glslc_common = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value, std::common_type_t<A, B>>;
template <class A, class B> inline glslc_common<A, B> min(A a, B b) { typedef std::common_type_t<A, B> T; T x = T(a), y = T(b); return y < x ? y : x; }
template <class A, class B> inline glslc_common<A, B> max(A a, B b) { typedef std::common_type_t<A, B> T; T x = T(a), y = T(b); return x < y ? y : x; }
template <class A, class B, class C2, class = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value && std::is_arithmetic<C2>::value>>
inline std::common_type_t<A, B, C2> clamp(A x, B lo, C2 hi) { typedef std::common_type_t<A, B, C2> T; return min(max(T(x), T(lo)), T(hi)); }
inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline float step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
inline float smoothstep(float e0, float e1, float x)
{
  float t = clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return t * t * (3.0f - 2.0f * t);
}
inline float abs(float x) { return ptf_abs(x); }
inline int   abs(int x) { return x < 0 ? -x : x; }
inline float sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
inline float floor(float x) { return ::floorf(x); }
inline float ceil(float x) { return ::ceilf(x); }
inline float fract(float x) { return x - ::floorf(x); }
inline float mod(float x, float y) { return x - y * ::floorf(x / y); }
inline float sqrt(float x) { return ::sqrtf(x); }
inline float inversesqrt(float x) { return 1.0f / ::sqrtf(x); }
inline float roundEven(float x) { return ::rintf(x); }
inline bool  isnan(float x) { return ptf_isnan(x) != 0; }
inline bool  isinf(float x) { return ptf_isinf(x) != 0; }
inline float sin(float x) { return pt_sin(x); }
inline float cos(float x) { return pt_cos(x); }
inline float tan(float x) { return pt_tan(x); }
inline float asin(float x) { return pt_asin(x); }
inline float acos(float x) { return pt_acos(x); }
inline float atan(float y, float x) { return pt_atan2(y, x); }
inline float atan(float x) { return pt_atan(x); }
inline float exp(float x) { return pt_exp(x); }
inline float log(float x) { return pt_log(x); }
template <class A, class B, class = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>> inline float pow(A x, B y) { return pt_pow(float(x), float(y)); }
inline float length(float x) { return ptf_abs(x); }
inline uint  floatBitsToUint(float f) { return ptf_bits(f); }
inline int   floatBitsToInt(float f) { return (int)ptf_bits(f); }
inline float uintBitsToFloat(uint u) { return ptf_from_bits(u); }
inline float intBitsToFloat(int i) { return ptf_from_bits((uint)i); }
inline vec3  uintBitsToFloat(const uvec3& u) { return vec3(ptf_from_bits(u.x), ptf_from_bits(u.y), ptf_from_bits(u.z)); }

// ---- vector built-ins ------------------------------------------------------------------------------------------
inline float dot(const vec2& a, const vec2& b) { return a.x * b.x + a.y * b.y; }
inline float dot(const vec3& a, const vec3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float dot(const vec4& a, const vec4& b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
inline vec3  cross(const vec3& a, const vec3& b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline float length(const vec2& a) { return ::sqrtf(dot(a, a)); }
inline float length(const vec3& a) { return ::sqrtf(dot(a, a)); }
inline float distance(const vec3& a, const vec3& b) { return length(a - b); }
inline vec2  normalize(const vec2& a) { return a * (1.0f / ::sqrtf(dot(a, a))); }
inline vec3  normalize(const vec3& a) { return a * (1.0f / ::sqrtf(dot(a, a))); }
inline vec3  reflect(const vec3& I, const vec3& N) { return I - (2.0f * dot(N, I)) * N; }
inline vec3  refract(const vec3& I, const vec3& N, float eta)
{
  float d = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  if(k < 0.0f)
    return vec3(0.0f);
  return eta * I - (eta * d + ::sqrtf(k)) * N;
}
#define GLSLC_MAP1(fn)                                                                     \
  inline vec2 fn(const vec2& a) { return vec2(fn(a.x), fn(a.y)); }                         \
  inline vec3 fn(const vec3& a) { return vec3(fn(a.x), fn(a.y), fn(a.z)); }                \
  inline vec4 fn(const vec4& a) { return vec4(fn(a.x), fn(a.y), fn(a.z), fn(a.w)); }
GLSLC_MAP1(abs)
GLSLC_MAP1(sign)
GLSLC_MAP1(floor)
GLSLC_MAP1(fract)
GLSLC_MAP1(sqrt)
GLSLC_MAP1(sin)
GLSLC_MAP1(cos)
GLSLC_MAP1(exp)
GLSLC_MAP1(log)
#undef GLSLC_MAP1
#define GLSLC_MAP2(fn)                                                                                          \
  inline vec2 fn(const vec2& a, const vec2& b) { return vec2(fn(a.x, b.x), fn(a.y, b.y)); }                      \
  inline vec3 fn(const vec3& a, const vec3& b) { return vec3(fn(a.x, b.x), fn(a.y, b.y), fn(a.z, b.z)); }        \
  inline vec4 fn(const vec4& a, const vec4& b) { return vec4(fn(a.x, b.x), fn(a.y, b.y), fn(a.z, b.z), fn(a.w, b.w)); } \
  inline vec2 fn(const vec2& a, float b) { return vec2(fn(a.x, b), fn(a.y, b)); }                                \
  inline vec3 fn(const vec3& a, float b) { return vec3(fn(a.x, b), fn(a.y, b), fn(a.z, b)); }                    \
  inline vec4 fn(const vec4& a, float b) { return vec4(fn(a.x, b), fn(a.y, b), fn(a.z, b), fn(a.w, b)); }
GLSLC_MAP2(min)
GLSLC_MAP2(max)
GLSLC_MAP2(pow)
GLSLC_MAP2(mod)
#undef GLSLC_MAP2
inline vec3 step(const vec3& e, const vec3& x) { return vec3(step(e.x, x.x), step(e.y, x.y), step(e.z, x.z)); }
inline vec2 mix(const vec2& a, const vec2& b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 mix(const vec3& a, const vec3& b, float t) { return a * (1.0f - t) + b * t; }
inline vec4 mix(const vec4& a, const vec4& b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 mix(const vec3& a, const vec3& b, const vec3& t) { return vec3(mix(a.x, b.x, t.x), mix(a.y, b.y, t.y), mix(a.z, b.z, t.z)); }
inline vec3 mix(const vec3& a, const vec3& b, const bvec3& t) { return vec3(t.x ? b.x : a.x, t.y ? b.y : a.y, t.z ? b.z : a.z); }
template <class B, class C2, class = std::enable_if_t<std::is_arithmetic<B>::value && std::is_arithmetic<C2>::value>>
inline vec3 clamp(const vec3& v, B lo, C2 hi) { return vec3(clamp(v.x, float(lo), float(hi)), clamp(v.y, float(lo), float(hi)), clamp(v.z, float(lo), float(hi))); }
template <class B, class C2, class = std::enable_if_t<std::is_arithmetic<B>::value && std::is_arithmetic<C2>::value>>
inline vec2 clamp(const vec2& v, B lo, C2 hi) { return vec2(clamp(v.x, float(lo), float(hi)), clamp(v.y, float(lo), float(hi))); }
inline bvec3 lessThan(const vec3& a, const vec3& b) { return bvec3(a.x < b.x, a.y < b.y, a.z < b.z); }
inline vec4  unpackUnorm4x8(uint p) { return vec4(float(p & 0xffu) / 255.0f, float((p >> 8) & 0xffu) / 255.0f, float((p >> 16) & 0xffu) / 255.0f, float(p >> 24) / 255.0f); }

// ---- matrices (column-major) -------------------------------------------------------------------------------------
struct mat3 {
  vec3 c[3];
  mat3() {}
  mat3(const vec3& a, const vec3& b, const vec3& d) { c[0] = a; c[1] = b; c[2] = d; }
  mat3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2)
  {
    c[0] = vec3(a0, a1, a2); c[1] = vec3(b0, b1, b2); c[2] = vec3(c0, c1, c2);
  }
  vec3&       operator[](int i) { return c[i]; }
  const vec3& operator[](int i) const { return c[i]; }
};
struct mat4x3 {  // 4 columns of vec3
  vec3 c[4];
  vec3&       operator[](int i) { return c[i]; }
  const vec3& operator[](int i) const { return c[i]; }
};
struct mat4 {
  vec4 c[4];
  mat4() {}
  explicit mat4(const mat4x3& m)
  {
    c[0] = vec4(m.c[0], 0.0f); c[1] = vec4(m.c[1], 0.0f); c[2] = vec4(m.c[2], 0.0f); c[3] = vec4(m.c[3], 1.0f);
  }
  vec4&       operator[](int i) { return c[i]; }
  const vec4& operator[](int i) const { return c[i]; }
};
inline vec3 operator*(const mat3& m, const vec3& v) { return (m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z; }
inline vec4 operator*(const mat4& m, const vec4& v) { return ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3] * v.w; }
inline vec4 operator*(const vec4& v, const mat4& m) { return vec4(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2]), dot(v, m.c[3])); }
inline vec3 operator*(const mat4x3& m, const vec4& v) { return ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3] * v.w; }
inline vec4 operator*(const vec3& v, const mat4x3& m) { return vec4(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2]), dot(v, m.c[3])); }

}  // namespace glslc

// <cmath> macros that collide with identifiers of the shaders (globals.glsl declares `const float M_PI`, `#define INFINITY`)
#ifndef GLSLC_KEEP_MATH_DEFINES
#undef M_PI
#undef M_PI_2
#undef M_PI_4
#undef M_1_PI
#undef M_2_PI
#undef INFINITY
#endif
