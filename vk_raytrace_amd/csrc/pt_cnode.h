// WideNode -> CompactNode (pt_device.h): the per-node body, plain C++ so that the conversion kernel (pt_accel.hip k_compact_nodes) and the CPU test
// harness (tests/cpp/trace_host.cpp) run the same code.  Quantisation in double, so that "the decoded box encloses the fp32 box" is exact
// arithmetic, not an argument about rounding.  Returns false when the node cannot be represented (non-finite or inverted boxes of real children).
#pragma once
#include <cfloat>
#include <cmath>
#include "pt_device.h"

#if defined(__HIPCC__)
#define CN_FN __host__ __device__ inline
#else
#define CN_FN inline
#endif

CN_FN uint32_t cn_half_of_int(uint32_t v)  // fp16 bit pattern of the integer v (0 .. 2047): exact
{
  if(v == 0)
    return 0u;
  int b = 0;
  while((v >> (b + 1)) != 0)
    ++b;
  return ((uint32_t(b) + 15u) << 10) | ((v << (10 - b)) & 0x3ffu);
}

CN_FN bool cn_encode(const WideNode& w, CompactNode& c)
{
  const float    lo[3][4] = {{w.minx[0].x, w.minx[0].y, w.minx[0].z, w.minx[0].w}, {w.miny[0].x, w.miny[0].y, w.miny[0].z, w.miny[0].w}, {w.minz[0].x, w.minz[0].y, w.minz[0].z, w.minz[0].w}};
  const float    hi[3][4] = {{w.maxx[0].x, w.maxx[0].y, w.maxx[0].z, w.maxx[0].w}, {w.maxy[0].x, w.maxy[0].y, w.maxy[0].z, w.maxy[0].w}, {w.maxz[0].x, w.maxz[0].y, w.maxz[0].z, w.maxz[0].w}};
  const uint32_t cc[4]    = {w.child[0].x, w.child[0].y, w.child[0].z, w.child[0].w};
  float          p[3];
  uint32_t       e[3];
  uint32_t       q[3][8];
  bool           ok = true;
  for(int a = 0; a < 3; ++a)
  {
    float mn = FLT_MAX, mx = -FLT_MAX;
    for(int k = 0; k < 4; ++k)
      if(cc[k] != BVH_NONE)
      {
        const bool fin = fabsf(lo[a][k]) <= FLT_MAX && fabsf(hi[a][k]) <= FLT_MAX;  // (false for NaN as well)
        ok = ok && fin && lo[a][k] <= hi[a][k];
        mn = fminf(mn, lo[a][k]);
        mx = fmaxf(mx, hi[a][k]);
      }
    if(!(mn <= mx))
      mn = mx = 0.f;  // no child at all (cannot happen for a built node): a point grid
    p[a]             = mn;
    const double ext = double(mx) - double(mn);
    int          ee  = 27;  // floor 2^-100: ray-side products with the step never underflow
    while(ee < 254 && double(CN_GRID_MAX) * ldexp(1.0, ee - 127) < ext)
      ++ee;
    ok   = ok && double(CN_GRID_MAX) * ldexp(1.0, ee - 127) >= ext;
    e[a] = uint32_t(ee);
    const double step = ldexp(1.0, ee - 127);
    for(int k = 0; k < 4; ++k)
    {
      double ql = 0.0, qh = 0.0;
      if(cc[k] != BVH_NONE && ok)
      {
        ql = floor((double(lo[a][k]) - double(mn)) / step);
        qh = ceil((double(hi[a][k]) - double(mn)) / step);
        ql = fmin(fmax(ql, 0.0), double(CN_GRID_MAX));
        qh = fmin(fmax(qh, 0.0), double(CN_GRID_MAX));
        while(ql > 0.0 && double(mn) + ql * step > double(lo[a][k]))
          ql -= 1.0;
        while(qh < double(CN_GRID_MAX) && double(mn) + qh * step < double(hi[a][k]))
          qh += 1.0;
        ok = ok && double(mn) + qh * step >= double(hi[a][k]);
      }
      q[a][k]     = uint32_t(ql);
      q[a][4 + k] = uint32_t(qh);
    }
  }
  c.px = p[0]; c.py = p[1]; c.pz = p[2];
  c.exps = e[0] | (e[1] << 8) | (e[2] << 16);
  for(int a = 0; a < 3; ++a)
  {
    c.ax[a].x = cn_half_of_int(q[a][0]) | (cn_half_of_int(q[a][1]) << 16);
    c.ax[a].y = cn_half_of_int(q[a][2]) | (cn_half_of_int(q[a][3]) << 16);
    c.ax[a].z = cn_half_of_int(q[a][4]) | (cn_half_of_int(q[a][5]) << 16);
    c.ax[a].w = cn_half_of_int(q[a][6]) | (cn_half_of_int(q[a][7]) << 16);
  }
  c.child = w.child[0];
  return ok;
}

