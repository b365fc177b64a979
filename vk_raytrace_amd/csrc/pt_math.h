// Device-side fp32 vector maths for the gfx950 path tracer.
//
// The evaluation order of every operation is fixed (and the build passes -ffp-contract=off) because
// image parity with the reference's arithmetic is checked bit-for-bit wherever no transcendental
// function is involved (DESIGN.md "Numerical contract"):
//   dot3(a,b)   = (a.x*b.x + a.y*b.y) + a.z*b.z
//   unit(v)     = v * (1 / sqrt(dot3(v,v)))
//   lerp(a,b,t) = a*(1-t) + b*t               (GLSL mix)
//   fmin2/fmax2 = GLSL min/max formulas (b<a?b:a / a<b?b:a), not IEEE minNum/maxNum
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pt_fpmath.h"  // the fp32 transcendental contract: one fixed IEEE operation sequence per function, same bits on host and device

#define PT_DEV __device__ __forceinline__

struct f2 {
  float x, y;
};
struct f3 {
  float x, y, z;
};
struct f4 {
  float x, y, z, w;
};

PT_DEV f2 mk2(float x, float y) { return f2{x, y}; }
PT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PT_DEV f3 splat3(float s) { return f3{s, s, s}; }
PT_DEV f4 mk4(float x, float y, float z, float w) { return f4{x, y, z, w}; }
PT_DEV f3 xyz(const f4& v) { return f3{v.x, v.y, v.z}; }
PT_DEV f3 xyz(const float4& v) { return f3{v.x, v.y, v.z}; }

PT_DEV float fmin2(float a, float b) { return b < a ? b : a; }
PT_DEV float fmax2(float a, float b) { return a < b ? b : a; }
PT_DEV float clampf(float x, float lo, float hi) { return fmin2(fmax2(x, lo), hi); }
PT_DEV float lerp(float a, float b, float t) { return a * (1.0f - t) + b * t; }
PT_DEV float smooth(float e0, float e1, float x)
{
  float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
  return t * t * (3.0f - 2.0f * t);
}

PT_DEV f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
PT_DEV f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
PT_DEV f2 operator*(f2 a, float s) { return f2{a.x * s, a.y * s}; }

PT_DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
PT_DEV f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
PT_DEV f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
PT_DEV f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
PT_DEV f3 operator/(f3 a, f3 b) { return f3{a.x / b.x, a.y / b.y, a.z / b.z}; }
PT_DEV f3 operator+(f3 a, float s) { return f3{a.x + s, a.y + s, a.z + s}; }
PT_DEV f3 operator-(f3 a, float s) { return f3{a.x - s, a.y - s, a.z - s}; }
PT_DEV f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
PT_DEV f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
PT_DEV f3& operator*=(f3& a, float s) { a = a * s; return a; }
PT_DEV f3& operator/=(f3& a, float s) { a = a / s; return a; }

PT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_DEV f3 cross3(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_DEV float len3(f3 a) { return sqrtf(dot3(a, a)); }
PT_DEV f3 unit(f3 a)
{
  float inv = 1.0f / sqrtf(dot3(a, a));
  return a * inv;
}
PT_DEV f3 lerp(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
PT_DEV f3 pow3(f3 a, float e) { return f3{pt_pow(a.x, e), pt_pow(a.y, e), pt_pow(a.z, e)}; }
PT_DEV f3 exp3(f3 a) { return f3{pt_exp(a.x), pt_exp(a.y), pt_exp(a.z)}; }
PT_DEV f3 log3(f3 a) { return f3{pt_log(a.x), pt_log(a.y), pt_log(a.z)}; }
PT_DEV f3 sqrt3(f3 a) { return f3{sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)}; }
PT_DEV f3 mirror(f3 I, f3 N) { return I - N * (2.0f * dot3(N, I)); }  // GLSL reflect
PT_DEV f3 bend(f3 I, f3 N, float eta)                                   // GLSL refract
{
  float d = dot3(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  if(k < 0.0f)
    return splat3(0.0f);
  return I * eta - N * (eta * d + sqrtf(k));
}

PT_DEV f4 operator+(f4 a, f4 b) { return f4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
PT_DEV f4 operator*(f4 a, float s) { return f4{a.x * s, a.y * s, a.z * s, a.w * s}; }
PT_DEV f4 operator*(f4 a, f4 b) { return f4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
PT_DEV float dot4(f4 a, f4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }

// Affine 3x4 transform stored as 3 rows of float4 (row r = (m[0][r], m[1][r], m[2][r], m[3][r]) of the
// column-major GLSL matrix), so one point transform is three dot-like chains on aligned 16-byte loads.
struct Affine {
  float4 r0, r1, r2;
};
// M * vec4(p,1):  ((c0*x + c1*y) + c2*z) + c3
PT_DEV f3 xform_point(const Affine& m, f3 p)
{
  return f3{((m.r0.x * p.x + m.r0.y * p.y) + m.r0.z * p.z) + m.r0.w * 1.0f,  //
            ((m.r1.x * p.x + m.r1.y * p.y) + m.r1.z * p.z) + m.r1.w * 1.0f,  //
            ((m.r2.x * p.x + m.r2.y * p.y) + m.r2.z * p.z) + m.r2.w * 1.0f};
}
// mat4(M) * vec4(d,0)
PT_DEV f3 xform_dir(const Affine& m, f3 d)
{
  return f3{(m.r0.x * d.x + m.r0.y * d.y) + m.r0.z * d.z, (m.r1.x * d.x + m.r1.y * d.y) + m.r1.z * d.z, (m.r2.x * d.x + m.r2.y * d.y) + m.r2.z * d.z};
}
// vec3(n * M): component i = dot(n, column i)
PT_DEV f3 xform_rowvec(f3 n, const Affine& m)
{
  return f3{(n.x * m.r0.x + n.y * m.r1.x) + n.z * m.r2.x, (n.x * m.r0.y + n.y * m.r1.y) + n.z * m.r2.y, (n.x * m.r0.z + n.y * m.r1.z) + n.z * m.r2.z};
}
// mat3(c0,c1,c2) * v
PT_DEV f3 basis_mul(f3 c0, f3 c1, f3 c2, f3 v) { return (c0 * v.x + c1 * v.y) + c2 * v.z; }

// ---- RNG (reference: shaders/random.glsl:34-48 tea, :59-65 pcg, :98-102 rand) -----------------------
PT_DEV uint32_t rng_tea(uint32_t v0, uint32_t v1)
{
  uint32_t s0 = 0;
#pragma unroll
  for(int n = 0; n < 16; ++n)
  {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}
PT_DEV float rng_next(uint32_t& state)
{
  state         = state * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  word          = (word >> 22u) ^ word;
  return __uint_as_float(0x3f800000u | (word >> 9)) - 1.0f;
}

// constants (reference: shaders/globals.glsl:27-43); PI_MACRO is the `#define PI`, PI_CONST the `const float M_PI`
#define PT_PI 3.14159265358979323f
#define PT_TWO_PI 6.28318530717958648f
#define PT_1_OVER_PI 0.318309886183790671538f
#define PT_INFINITY 1e32f
