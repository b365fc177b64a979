// TEST INFRASTRUCTURE -- part of the CPU oracle (see oracle/README.md). Not linked into the product.
//
// Minimal GLSL-semantics vector maths for the CPU restatement of the reference shaders.
// Every operation is spelled out in fp32 with a fixed evaluation order (no FMA contraction: the
// Makefile passes -ffp-contract=off) so that the HIP kernels can reproduce the arithmetic bit for
// bit wherever no transcendental function is involved.
//
// Conventions fixed here (GLSL leaves them to the implementation; both sides of the parity tests
// use exactly these):
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   normalize(v)  = v * (1.0f / sqrt(dot(v,v)))
//   mix(a,b,t)    = a*(1-t) + b*t                      (GLSL spec formula)
//   min(a,b)      = b < a ? b : a ;  max(a,b) = a < b ? b : a      (GLSL spec formulas)
//   M * v (mat4)  = ((c0*v.x + c1*v.y) + c2*v.z) + c3*v.w
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include "../include/pt_fpmath.h"  // the fp32 transcendental contract (one fixed IEEE operation sequence per function)

namespace orc {

// Transcendental functions go through these wrappers.  Mode 0 (default): the fp32 transcendental contract of
// include/pt_fpmath.h -- GLSL leaves the accuracy of these functions to the implementation, the contract fixes
// one legal implementation as a sequence of IEEE operations, and the HIP kernels run the same sequence, so
// path-traced frames of the oracle and of the product are comparable bit for bit.  Mode 1: the double-precision
// libm function rounded to fp32 (practically correctly rounded).  Mode 2: that result moved by one ulp in ~30 %
// of the calls (deterministic hash of the argument).  Modes 1/2 are calibration tools: rendering the same input
// in two modes measures how sensitive an image is to ~1-ulp differences in sin/cos/pow/... (what a driver with
// a different libm would produce); no parity test depends on them any more.
extern int g_math_mode;
inline float perturb_ulp(float r, float x)
{
  uint32_t h;
  std::memcpy(&h, &x, 4);
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  if((h % 10u) < 3u && std::isfinite(r) && r != 0.0f)
    r = std::nextafter(r, (h & 0x80000u) ? INFINITY : -INFINITY);
  return r;
}
#define ORC_MATH1(name, fn)                                                                                         \
  inline float name(float x)                                                                                        \
  {                                                                                                                 \
    if(g_math_mode == 0) return ::pt_##fn(x);                                                                       \
    float r = (float)::fn((double)x);                                                                               \
    return g_math_mode == 2 ? perturb_ulp(r, x) : r;                                                                \
  }
ORC_MATH1(msin, sin)
ORC_MATH1(mcos, cos)
ORC_MATH1(mtan, tan)
ORC_MATH1(macos, acos)
ORC_MATH1(masin, asin)
ORC_MATH1(mexp, exp)
ORC_MATH1(mlog, log)
inline float mpow(float x, float y)
{
  if(g_math_mode == 0) return ::pt_pow(x, y);
  float r = (float)::pow((double)x, (double)y);
  return g_math_mode == 2 ? perturb_ulp(r, x + y) : r;
}
inline float matan2(float y, float x)
{
  if(g_math_mode == 0) return ::pt_atan2(y, x);
  float r = (float)::atan2((double)y, (double)x);
  return g_math_mode == 2 ? perturb_ulp(r, x - y) : r;
}

struct vec2 {
  float x, y;
  vec2() : x(0), y(0) {}
  vec2(float a, float b) : x(a), y(b) {}
  explicit vec2(float a) : x(a), y(a) {}
};
struct vec3 {
  float x, y, z;
  vec3() : x(0), y(0), z(0) {}
  vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  explicit vec3(float a) : x(a), y(a), z(a) {}
  float&       operator[](int i) { return (&x)[i]; }
  const float& operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
  float x, y, z, w;
  vec4() : x(0), y(0), z(0), w(0) {}
  vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
  vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
  vec3 xyz() const { return vec3(x, y, z); }
};

// ---- scalar helpers ------------------------------------------------------------------------
inline float gmin(float a, float b) { return b < a ? b : a; }
inline float gmax(float a, float b) { return a < b ? b : a; }
inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
inline float gmix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline float gstep(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
inline float gsmoothstep(float e0, float e1, float x)
{
  float t = gclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return t * t * (3.0f - 2.0f * t);
}
inline float    uintBitsToFloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t floatBitsToUint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline int32_t  floatBitsToInt(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }
inline float    intBitsToFloat(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

// ---- vec2 -----------------------------------------------------------------------------------
inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
inline vec2 operator*(vec2 a, vec2 b) { return vec2(a.x * b.x, a.y * b.y); }
inline vec2 operator/(vec2 a, vec2 b) { return vec2(a.x / b.x, a.y / b.y); }
inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }

// ---- vec3 -----------------------------------------------------------------------------------
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec3 operator+(vec3 a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
inline vec3 operator-(vec3 a, float s) { return vec3(a.x - s, a.y - s, a.z - s); }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
inline vec3& operator/=(vec3& a, float s) { a = a / s; return a; }
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline vec3  cross(vec3 a, vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3  normalize(vec3 a)
{
  float inv = 1.0f / std::sqrt(dot(a, a));
  return a * inv;
}
inline vec3 gmix(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 gmix(vec3 a, vec3 b, vec3 t) { return vec3(gmix(a.x, b.x, t.x), gmix(a.y, b.y, t.y), gmix(a.z, b.z, t.z)); }
inline vec3 gmax(vec3 a, vec3 b) { return vec3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
inline vec3 gclamp(vec3 a, float lo, float hi) { return vec3(gclamp(a.x, lo, hi), gclamp(a.y, lo, hi), gclamp(a.z, lo, hi)); }
inline vec3 gpow(vec3 a, float e) { return vec3(mpow(a.x, e), mpow(a.y, e), mpow(a.z, e)); }
inline vec3 gpow(vec3 a, vec3 e) { return vec3(mpow(a.x, e.x), mpow(a.y, e.y), mpow(a.z, e.z)); }
inline vec3 gexp(vec3 a) { return vec3(mexp(a.x), mexp(a.y), mexp(a.z)); }
inline vec3 glog(vec3 a) { return vec3(mlog(a.x), mlog(a.y), mlog(a.z)); }
inline vec3 gsqrt(vec3 a) { return vec3(std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)); }
inline vec3 gfloor(vec3 a) { return vec3(std::floor(a.x), std::floor(a.y), std::floor(a.z)); }
// GLSL reflect / refract (spec formulas)
inline vec3 reflect(vec3 I, vec3 N) { return I - N * (2.0f * dot(N, I)); }
inline vec3 refract(vec3 I, vec3 N, float eta)
{
  float d = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  if(k < 0.0f)
    return vec3(0.0f);
  return I * eta - N * (eta * d + std::sqrt(k));
}

// ---- vec4 -----------------------------------------------------------------------------------
inline vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline vec4 operator*(vec4 a, vec4 b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline float dot(vec4 a, vec4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }

// ---- mat4 (column-major, GLSL) -----------------------------------------------------------------
struct mat4 {
  vec4 c[4];
  mat4() {}
  explicit mat4(const float* m)
  {
    for(int i = 0; i < 4; ++i)
      c[i] = vec4(m[i * 4 + 0], m[i * 4 + 1], m[i * 4 + 2], m[i * 4 + 3]);
  }
};
inline vec4 operator*(const mat4& m, vec4 v) { return ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3] * v.w; }
// row-vector * matrix: component i = dot(v, column i)
inline vec4 operator*(vec4 v, const mat4& m) { return vec4(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2]), dot(v, m.c[3])); }

// mat4x3: 4 columns of vec3 (an affine object<->world transform)
struct mat4x3 {
  vec3 c[4];
};
inline vec3 mul_point(const mat4x3& m, vec3 p) { return ((m.c[0] * p.x + m.c[1] * p.y) + m.c[2] * p.z) + m.c[3] * 1.0f; }  // M * vec4(p,1)
inline vec3 mul_dir(const mat4x3& m, vec3 d) { return (m.c[0] * d.x + m.c[1] * d.y) + m.c[2] * d.z; }                     // mat4(M) * vec4(d,0), the "+ c3*0" dropped
inline vec3 mul_rowvec(vec3 n, const mat4x3& m) { return vec3(dot(n, m.c[0]), dot(n, m.c[1]), dot(n, m.c[2])); }           // vec3(n * M)

// mat3 from three column vectors, times vec3
inline vec3 mul_mat3(vec3 c0, vec3 c1, vec3 c2, vec3 v) { return (c0 * v.x + c1 * v.y) + c2 * v.z; }

}  // namespace orc
