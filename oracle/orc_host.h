// TEST INFRASTRUCTURE -- part of the CPU oracle (see oracle/README.md). Not linked into the product.
//
// Restatement of the reference's HOST-side arithmetic that feeds the hot path:
//   unit-vector codec            shaders/compress.glsl:31-139 (host shims + compress_unit_vec)
//   vertex packing               src/scene.cpp:219-242
//   alias table for the HDR      src/hdr_sampling.cpp:107-248
//   camera matrices              src/scene.cpp:629-640 (+ glm::lookAt / perspectiveRH_ZO / inverse)
//   sampler translation          src/scene.cpp:447-482,561-571
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>
#include "../include/pt_types.h"
#include "glsl_math.h"

namespace orc {

// shaders/compress.glsl:70-94 (the host stand-in for GLSL roundEven)
inline float host_roundEven(float x)
{
  int   ipart = static_cast<int>(x);
  float fpart = static_cast<float>(ipart);
  float frac  = x - std::floor(x);
  if(frac > 0.5f || frac < 0.5f)
    return std::round(x);
  if((ipart % 2) == 0)
    return fpart;
  if(x <= 0)
    return fpart - 1;
  return fpart + 1;
}

// shaders/compress.glsl:111-139
inline uint32_t compress_unit_vec(vec3 nv)
{
  const float stack_max = 3.402823466e+38f;
  if((nv.x < stack_max) && !std::isinf(nv.x))
  {
    const float d = 32767.0f / (std::fabs(nv.x) + std::fabs(nv.y) + std::fabs(nv.z));
    int         x = int(host_roundEven(nv.x * d));
    int         y = int(host_roundEven(nv.y * d));
    if(nv.z < 0.0f)
    {
      const int mx = x >> 31;
      const int my = y >> 31;
      const int t  = 32767 + mx + my;
      const int ox = x;
      x            = (t - (y ^ my)) ^ mx;
      y            = (t - (ox ^ mx)) ^ my;
    }
    uint32_t packed = (uint32_t(y + 32767) << 16) | uint32_t(x + 32767);
    if(packed == ~0u)
      return ~0x1u;
    return packed;
  }
  return ~0u;
}

// shaders/compress.glsl:58-72 / glm::packUnorm4x8 used at src/scene.cpp:226,375
inline uint32_t packUnorm4x8(vec4 v)
{
  auto     q = [](float c) { return (uint32_t)(unsigned char)std::round(std::min(std::max(c, 0.0f), 1.0f) * 255.f); };
  return q(v.x) | (q(v.y) << 8) | (q(v.z) << 16) | (q(v.w) << 24);
}

// src/scene.cpp:219-242
inline void pack_vertex(const float* pos, const float* nrm, const float* tan4, const float* uv, const float* col4, pt_VertexAttributes* out)
{
  out->position[0] = pos[0];
  out->position[1] = pos[1];
  out->position[2] = pos[2];
  out->normal      = compress_unit_vec(vec3(nrm[0], nrm[1], nrm[2]));
  out->tangent     = compress_unit_vec(vec3(tan4[0], tan4[1], tan4[2]));
  out->texcoord[0] = uv[0];
  out->color       = packUnorm4x8(vec4(col4[0], col4[1], col4[2], col4[3]));
  uint32_t bits    = floatBitsToUint(uv[1]);
  if(tan4[3] > 0)
    bits |= 1u;
  else
    bits &= ~1u;
  out->texcoord[1] = uintBitsToFloat(bits);
}

// src/hdr_sampling.cpp:107-176
inline float build_aliasmap(const std::vector<float>& data, std::vector<pt_EnvAccel>& accel)
{
  uint32_t size = (uint32_t)data.size();
  float    sum  = std::accumulate(data.begin(), data.end(), 0.f);
  float    inverseAverage = float(size) / sum;
  for(uint32_t i = 0; i < size; ++i)
  {
    accel[i].q     = data[i] * inverseAverage;
    accel[i].alias = i;
  }
  std::vector<uint32_t> part(size);
  uint32_t              s = 0u, large = size;
  for(uint32_t i = 0; i < size; ++i)
  {
    if(accel[i].q < 1.f)
      part[s++] = i;
    else
      part[--large] = i;
  }
  for(s = 0; s < large && large < size; ++s)
  {
    const uint32_t lo = part[s];
    const uint32_t hi = part[large];
    accel[lo].alias   = hi;
    const float diff  = 1.f - accel[lo].q;
    accel[hi].q -= diff;
    if(accel[hi].q < 1.0f)
      large++;
  }
  return sum;
}

// src/hdr_sampling.cpp:187-248
inline void create_environment_accel(const float* pixels, uint32_t rx, uint32_t ry, std::vector<pt_EnvAccel>& envAccel, float& integral, float& average)
{
  envAccel.assign(size_t(rx) * ry, pt_EnvAccel{});
  std::vector<float> importance(size_t(rx) * ry);
  float              cosTheta0 = 1.0f;
  const float        stepPhi   = float(2.0 * M_PI) / float(rx);
  const float        stepTheta = float(M_PI) / float(ry);
  double             total     = 0;
  for(uint32_t y = 0; y < ry; ++y)
  {
    const float theta1    = float(y + 1) * stepTheta;
    const float cosTheta1 = std::cos(theta1);
    const float area      = (cosTheta0 - cosTheta1) * stepPhi;
    cosTheta0             = cosTheta1;
    for(uint32_t x = 0; x < rx; ++x)
    {
      const uint32_t idx  = y * rx + x;
      const float*   p    = pixels + size_t(idx) * 4;
      float          lum  = p[0] * 0.2126f + p[1] * 0.7152f + p[2] * 0.0722f;
      importance[idx]     = area * std::max(p[0], std::max(p[1], p[2]));
      total += lum;
    }
  }
  average  = static_cast<float>(total) / static_cast<float>(rx * ry);
  integral = build_aliasmap(importance, envAccel);
  const float invInt = 1.0f / integral;
  for(uint32_t i = 0; i < rx * ry; ++i)
  {
    const float* p  = pixels + size_t(i) * 4;
    envAccel[i].pdf = std::max(p[0], std::max(p[1], p[2])) * invInt;
  }
  for(uint32_t i = 0; i < rx * ry; ++i)
    envAccel[i].aliasPdf = envAccel[envAccel[i].alias].pdf;
}

// 4x4 inverse in double (general cofactor expansion), column-major in/out.  The reference uses
// glm::inverse in fp32 (src/scene.cpp:634-635); rounding the double result to fp32 pins the value
// independent of the cofactor evaluation order.
inline bool invert4x4(const double* m, double* inv)
{
  double t[16];
  t[0]  = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  t[4]  = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  t[8]  = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  t[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  t[1]  = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  t[5]  = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  t[9]  = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  t[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  t[2]  = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  t[6]  = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  t[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  t[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  t[3]  = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  t[7]  = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  t[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  t[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * t[0] + m[1] * t[4] + m[2] * t[8] + m[3] * t[12];
  if(det == 0.0)
    return false;
  double r = 1.0 / det;
  for(int i = 0; i < 16; ++i)
    inv[i] = t[i] * r;
  return true;
}

// src/scene.cpp:629-640: view = lookAt(eye, center, up) (glm RH), proj = perspectiveRH_ZO(fov, aspect, 0.001, 1e5),
// proj[1][1] *= -1, then both inverted; focalDist = |center - eye|.
inline void camera_lookat(const float* eye, const float* center, const float* up, float fovDeg, float aspect, pt_SceneCamera* out)
{
  double e[3] = {eye[0], eye[1], eye[2]}, c[3] = {center[0], center[1], center[2]}, u0[3] = {up[0], up[1], up[2]};
  double f[3] = {c[0] - e[0], c[1] - e[1], c[2] - e[2]};
  double fl   = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for(double& x : f)
    x /= fl;
  double s[3] = {f[1] * u0[2] - f[2] * u0[1], f[2] * u0[0] - f[0] * u0[2], f[0] * u0[1] - f[1] * u0[0]};
  double sl   = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
  for(double& x : s)
    x /= sl;
  double u[3]  = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
  double view[16] = {s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0,
                     -(s[0] * e[0] + s[1] * e[1] + s[2] * e[2]), -(u[0] * e[0] + u[1] * e[1] + u[2] * e[2]), (f[0] * e[0] + f[1] * e[1] + f[2] * e[2]), 1};
  const double zn = 0.001, zf = 100000.0;
  double       th = std::tan((double)fovDeg * M_PI / 180.0 / 2.0);
  double       proj[16] = {0};
  proj[0]  = 1.0 / ((double)aspect * th);
  proj[5]  = -(1.0 / th);
  proj[10] = zf / (zn - zf);
  proj[11] = -1.0;
  proj[14] = -(zf * zn) / (zf - zn);
  double vi[16], pi[16];
  invert4x4(view, vi);
  invert4x4(proj, pi);
  for(int i = 0; i < 16; ++i)
  {
    out->viewInverse[i] = (float)vi[i];
    out->projInverse[i] = (float)pi[i];
  }
  out->focalDist = (float)fl;
  out->aperture  = 0.0f;
  out->nbLights  = 0;
}

// src/scene.cpp:447-482 (std::map::operator[] on an unknown key yields enum value 0) and :561-571
inline void sampler_from_gltf(int has_sampler, int mag, int min, int wrapS, int wrapT, pt_TextureDesc* io)
{
  if(!has_sampler)
  {
    io->magFilter = io->minFilter = PT_FILTER_LINEAR;
    io->wrapS = io->wrapT = PT_WRAP_REPEAT;
    return;
  }
  auto filt = [](int code) { return (code == 9729 || code == 9985 || code == 9987) ? PT_FILTER_LINEAR : PT_FILTER_NEAREST; };
  auto wrap = [](int code) { return code == 33071 ? PT_WRAP_CLAMP_TO_EDGE : (code == 33648 ? PT_WRAP_MIRRORED_REPEAT : PT_WRAP_REPEAT); };
  io->magFilter = filt(mag);
  io->minFilter = filt(min);
  io->wrapS     = wrap(wrapS);
  io->wrapT     = wrap(wrapT);
}

}  // namespace orc
