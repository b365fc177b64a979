// Host-side scene packing helpers of libptmi.so (no GPU involved).  They produce the exact buffers the
// reference's Scene / HdrSampling classes upload, so a caller that holds raw glTF attributes can feed
// pt_set_scene / pt_set_env without linking nvpro_core.
//
// Behavioural contract (file:line = reference):
//   pt_compress_unit_vec   shaders/compress.glsl:70-94 (roundEven shim), :111-139
//   pt_pack_vertices       src/scene.cpp:219-242
//   pt_build_env_accel     src/hdr_sampling.cpp:107-176 (alias map), :187-248
//   pt_camera_lookat       src/scene.cpp:629-640 (+ glm::lookAt, glm::perspectiveRH_ZO, glm::inverse)
//   pt_sampler_from_gltf   src/scene.cpp:447-482, :561-571
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>
#include "../../include/pt_api.h"

namespace {

inline uint32_t f2u(float f)
{
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float u2f(uint32_t u)
{
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// round-half-to-even the way the reference's C++ shim does it
float round_half_even(float x)
{
  const int   whole = static_cast<int>(x);
  const float frac  = x - std::floor(x);
  if(frac != 0.5f)
    return std::round(x);
  if(whole % 2 == 0)
    return static_cast<float>(whole);
  return x <= 0 ? static_cast<float>(whole) - 1 : static_cast<float>(whole) + 1;
}

uint32_t quantize_unorm8(float c)
{
  c = std::min(std::max(c, 0.0f), 1.0f);
  return static_cast<uint32_t>(static_cast<unsigned char>(std::round(c * 255.f)));
}

// Gauss-Jordan inverse of a 4x4 (column-major storage is irrelevant for inversion) in double.
bool invert_4x4(const double in[16], double out[16])
{
  double a[4][8];
  for(int r = 0; r < 4; ++r)
    for(int c = 0; c < 4; ++c)
    {
      a[r][c]     = in[c * 4 + r];
      a[r][c + 4] = (r == c) ? 1.0 : 0.0;
    }
  for(int col = 0; col < 4; ++col)
  {
    int piv = col;
    for(int r = col + 1; r < 4; ++r)
      if(std::fabs(a[r][col]) > std::fabs(a[piv][col]))
        piv = r;
    if(a[piv][col] == 0.0)
      return false;
    if(piv != col)
      for(int c = 0; c < 8; ++c)
        std::swap(a[piv][c], a[col][c]);
    const double inv = 1.0 / a[col][col];
    for(int c = 0; c < 8; ++c)
      a[col][c] *= inv;
    for(int r = 0; r < 4; ++r)
      if(r != col)
      {
        const double f = a[r][col];
        if(f != 0.0)
          for(int c = 0; c < 8; ++c)
            a[r][c] -= f * a[col][c];
      }
  }
  for(int r = 0; r < 4; ++r)
    for(int c = 0; c < 4; ++c)
      out[c * 4 + r] = a[r][c + 4];
  return true;
}

}  // namespace

extern "C" {

uint32_t pt_compress_unit_vec(const float v[3])
{
  const float big = 3.402823466e+38f;
  if(!(v[0] < big) || std::isinf(v[0]))
    return ~0u;
  const float scale = 32767.0f / (std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]));
  int         qx    = int(round_half_even(v[0] * scale));
  int         qy    = int(round_half_even(v[1] * scale));
  if(v[2] < 0.0f)
  {  // fold the lower hemisphere over the diagonals
    const int sx = qx >> 31, sy = qy >> 31;
    const int t  = 32767 + sx + sy;
    const int fx = (t - (qy ^ sy)) ^ sx;
    const int fy = (t - (qx ^ sx)) ^ sy;
    qx           = fx;
    qy           = fy;
  }
  const uint32_t packed = (uint32_t(qy + 32767) << 16) | uint32_t(qx + 32767);
  return packed == ~0u ? ~0x1u : packed;
}

int pt_pack_vertices(uint32_t n, const float* positions, const float* normals, const float* tangents, const float* uvs, const float* colors,
                     pt_VertexAttributes* out)
{
  if(n && (!positions || !normals || !tangents || !uvs || !colors || !out))
    return PT_ERR_INVALID;
  for(uint32_t i = 0; i < n; ++i)
  {
    pt_VertexAttributes& o = out[i];
    std::memcpy(o.position, positions + 3 * size_t(i), 12);
    o.normal      = pt_compress_unit_vec(normals + 3 * size_t(i));
    o.tangent     = pt_compress_unit_vec(tangents + 4 * size_t(i));
    o.texcoord[0] = uvs[2 * size_t(i)];
    // handedness of the tangent goes into the LSB of v
    uint32_t vb   = f2u(uvs[2 * size_t(i) + 1]);
    vb            = tangents[4 * size_t(i) + 3] > 0 ? (vb | 1u) : (vb & ~1u);
    o.texcoord[1] = u2f(vb);
    const float* c = colors + 4 * size_t(i);
    o.color        = quantize_unorm8(c[0]) | (quantize_unorm8(c[1]) << 8) | (quantize_unorm8(c[2]) << 16) | (quantize_unorm8(c[3]) << 24);
  }
  return PT_OK;
}

int pt_build_env_accel(const float* px, int width, int height, pt_EnvAccel* accel, float* out_integral, float* out_average)
{
  if(!px || !accel || width <= 0 || height <= 0)
    return PT_ERR_INVALID;
  const uint32_t     rx = uint32_t(width), ry = uint32_t(height), count = rx * ry;
  std::vector<float> energy(count);
  // importance = solid angle of the texel row x max(r,g,b); also the CIE luminance average
  const float stepPhi = float(2.0 * M_PI) / float(rx), stepTheta = float(M_PI) / float(ry);
  float       cosPrev  = 1.0f;
  double      lumTotal = 0;
  for(uint32_t y = 0; y < ry; ++y)
  {
    const float cosNext = std::cos(float(y + 1) * stepTheta);
    const float area    = (cosPrev - cosNext) * stepPhi;
    cosPrev             = cosNext;
    for(uint32_t x = 0; x < rx; ++x)
    {
      const float* p   = px + 4 * size_t(y * rx + x);
      energy[y * rx + x] = area * std::max(p[0], std::max(p[1], p[2]));
      lumTotal += p[0] * 0.2126f + p[1] * 0.7152f + p[2] * 0.0722f;
    }
  }
  const float average = static_cast<float>(lumTotal) / static_cast<float>(count);

  // alias map
  const float sum = std::accumulate(energy.begin(), energy.end(), 0.f);
  const float toRatio = float(count) / sum;
  for(uint32_t i = 0; i < count; ++i)
  {
    accel[i].q     = energy[i] * toRatio;
    accel[i].alias = i;
  }
  std::vector<uint32_t> order(count);
  uint32_t              small = 0, large = count;
  for(uint32_t i = 0; i < count; ++i)
  {
    if(accel[i].q < 1.f)
      order[small++] = i;
    else
      order[--large] = i;
  }
  for(small = 0; small < large && large < count; ++small)
  {
    const uint32_t lo = order[small], hi = order[large];
    accel[lo].alias = hi;
    accel[hi].q -= 1.f - accel[lo].q;
    if(accel[hi].q < 1.0f)
      ++large;
  }
  const float invSum = 1.0f / sum;
  for(uint32_t i = 0; i < count; ++i)
  {
    const float* p = px + 4 * size_t(i);
    accel[i].pdf   = std::max(p[0], std::max(p[1], p[2])) * invSum;
  }
  for(uint32_t i = 0; i < count; ++i)
    accel[i].aliasPdf = accel[accel[i].alias].pdf;
  if(out_integral)
    *out_integral = sum;
  if(out_average)
    *out_average = average;
  return PT_OK;
}

int pt_camera_lookat(const float eye[3], const float center[3], const float up[3], float fov_degrees, float aspect, pt_SceneCamera* out)
{
  if(!eye || !center || !up || !out || !(aspect > 0))
    return PT_ERR_INVALID;
  auto sub   = [](const double* a, const double* b, double* r) { for(int i = 0; i < 3; ++i) r[i] = a[i] - b[i]; };
  auto dot   = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  auto cross = [](const double* a, const double* b, double* r) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
  };
  auto norm = [&](double* a) {
    double l = std::sqrt(dot(a, a));
    for(int i = 0; i < 3; ++i) a[i] /= l;
    return l;
  };
  double e[3] = {eye[0], eye[1], eye[2]}, c[3] = {center[0], center[1], center[2]}, u0[3] = {up[0], up[1], up[2]};
  double f[3], s[3], u[3];
  sub(c, e, f);
  const double dist = norm(f);
  if(!(dist > 0))
    return PT_ERR_INVALID;
  cross(f, u0, s);
  norm(s);
  cross(s, f, u);
  // glm::lookAt (right handed), column-major
  const double view[16] = {s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0, -dot(s, e), -dot(u, e), dot(f, e), 1};
  // glm::perspectiveRH_ZO(fovy, aspect, 0.001, 100000) with [1][1] negated
  const double zn = 0.001, zf = 100000.0, th = std::tan(double(fov_degrees) * M_PI / 180.0 * 0.5);
  double       proj[16] = {0};
  proj[0]  = 1.0 / (double(aspect) * th);
  proj[5]  = -1.0 / th;
  proj[10] = zf / (zn - zf);
  proj[11] = -1.0;
  proj[14] = -(zf * zn) / (zf - zn);
  double vi[16], pi[16];
  if(!invert_4x4(view, vi) || !invert_4x4(proj, pi))
    return PT_ERR_INVALID;
  for(int i = 0; i < 16; ++i)
  {
    out->viewInverse[i] = float(vi[i]);
    out->projInverse[i] = float(pi[i]);
  }
  out->focalDist = float(dist);
  out->aperture  = 0.0f;
  out->nbLights  = 0;
  return PT_OK;
}

int pt_sampler_from_gltf(int has_sampler, int gltf_mag, int gltf_min, int gltf_wrapS, int gltf_wrapT, pt_TextureDesc* io)
{
  if(!io)
    return PT_ERR_INVALID;
  if(!has_sampler)
  {  // textures without a sampler: linear / repeat
    io->magFilter = io->minFilter = PT_FILTER_LINEAR;
    io->wrapS = io->wrapT = PT_WRAP_REPEAT;
    return PT_OK;
  }
  // the reference looks the codes up with std::map::operator[]: unknown codes yield enum value 0 (NEAREST / REPEAT)
  auto filter = [](int code) {
    switch(code)
    {
      case 9729: case 9985: case 9987: return PT_FILTER_LINEAR;
      default: return PT_FILTER_NEAREST;
    }
  };
  auto wrap = [](int code) {
    switch(code)
    {
      case 33071: return PT_WRAP_CLAMP_TO_EDGE;
      case 33648: return PT_WRAP_MIRRORED_REPEAT;
      default: return PT_WRAP_REPEAT;
    }
  };
  io->magFilter = filter(gltf_mag);
  io->minFilter = filter(gltf_min);
  io->wrapS     = wrap(gltf_wrapS);
  io->wrapT     = wrap(gltf_wrapT);
  return PT_OK;
}

}  // extern "C"
