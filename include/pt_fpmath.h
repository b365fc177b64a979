/* pt_fpmath.h -- the fp32 transcendental contract of the path tracer.
 *
 * GLSL (and the Vulkan precision table) leaves the accuracy of sin/cos/tan/asin/acos/atan/exp/log/pow to the
 * implementation; the reference (the shaders directory) therefore has no defined bit pattern for them.  This header
 * fixes ONE legal implementation as a sequence of IEEE-754 binary32 operations (+ - * / sqrt fma rint and
 * integer bit manipulation, nothing else), so that the same source gives the same bits on the host cores
 * (oracle, oracle/_ref) and on gfx950 (libptmi.so): path-traced frames are then comparable bit for bit and
 * any mismatch is a defect, not libm noise.
 *
 * Requirements on the build (both sides): no fast-math, -ffp-contract=off (every fused operation below is an
 * explicit fma), correctly rounded division and square root (hipcc's default), fp32 denormals preserved.
 *
 * Accuracy (tests/test_fpmath.py, against double-precision libm): sin/cos <= 1.5 ulp for |x| <= 1e5,
 * tan <= 3 ulp, asin/acos/atan2 <= 2.5 ulp, exp/log <= 1 ulp, pow <= 1.5 ulp -- all tighter than Vulkan asks.
 */
#ifndef PT_FPMATH_H
#define PT_FPMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PT_FP __host__ __device__ __forceinline__
#else
#define PT_FP static inline
#endif

PT_FP float ptf_from_bits(uint32_t u)
{
  float f;
  memcpy(&f, &u, 4);
  return f;
}
PT_FP uint32_t ptf_bits(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
PT_FP float ptf_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_FP float ptf_abs(float x) { return ptf_from_bits(ptf_bits(x) & 0x7fffffffu); }
PT_FP float ptf_copysign(float mag, float sgn) { return ptf_from_bits((ptf_bits(mag) & 0x7fffffffu) | (ptf_bits(sgn) & 0x80000000u)); }
PT_FP int   ptf_isnan(float x) { return (ptf_bits(x) & 0x7fffffffu) > 0x7f800000u; }
PT_FP int   ptf_isinf(float x) { return (ptf_bits(x) & 0x7fffffffu) == 0x7f800000u; }
PT_FP float ptf_nan() { return ptf_from_bits(0x7fc00000u); }
PT_FP float ptf_inf() { return ptf_from_bits(0x7f800000u); }
/* 2^k for k in [-126, 127] */
PT_FP float ptf_pow2i(int k) { return ptf_from_bits((uint32_t)(k + 127) << 23); }

/* ---- sin / cos ---------------------------------------------------------------------------------------
 * k = rint(x * 2/pi); r = x - k*pi/2 with pi/2 split in three parts (Cody-Waite, fused); minimax kernels on
 * [-pi/4, pi/4].  |x| >= 1e5 is first folded by an exact fmod (the remainder is exactly representable, so
 * every correct fmodf returns the same bits).  */
PT_FP void ptf_sincos_reduce(float x, float* r, int* q)
{
  float ax = ptf_abs(x);
  if(!(ax < 1.0e5f))
  {
    if(!(ax < ptf_inf()))
    {
      *r = ptf_nan();
      *q = 0;
      return;
    }
    x = fmodf(x, 6.28318530717958648f);
  }
  float kf = rintf(x * 0.636619772367581343f);
  float t  = ptf_fma(-kf, 1.5707962512969971f, x);      /* pi/2 high 24 bits  */
  t        = ptf_fma(-kf, 7.5497894158615964e-08f, t);  /* next bits          */
  t        = ptf_fma(-kf, 5.3903029534742384e-15f, t);  /* tail               */
  *r       = t;
  *q       = (int)kf;
}
PT_FP float ptf_sin_kernel(float r)
{
  float z = r * r;
  float p = ptf_fma(2.7243891054e-06f, z, -1.9840040477e-04f);
  p       = ptf_fma(p, z, 8.3333319053e-03f);
  p       = ptf_fma(p, z, -1.6666667163e-01f);
  return ptf_fma(r * z, p, r);
}
PT_FP float ptf_cos_kernel(float r)
{
  float z = r * r;
  float p = ptf_fma(-2.7295911309e-07f, z, 2.4800561732e-05f);
  p       = ptf_fma(p, z, -1.3888888061e-03f);
  p       = ptf_fma(p, z, 4.1666667908e-02f);
  float hz = 0.5f * z;
  float w  = 1.0f - hz;
  return w + (((1.0f - w) - hz) + (z * z) * p);
}
PT_FP float pt_sin(float x)
{
  if(ptf_abs(x) < 2.44140625e-4f) /* 2^-12: sin x rounds to x (keeps -0 and denormals) */
    return x;
  float r;
  int   q;
  ptf_sincos_reduce(x, &r, &q);
  float v = (q & 1) ? ptf_cos_kernel(r) : ptf_sin_kernel(r);
  return (q & 2) ? -v : v;
}
PT_FP float pt_cos(float x)
{
  float r;
  int   q;
  ptf_sincos_reduce(x, &r, &q);
  float v = (q & 1) ? ptf_sin_kernel(r) : ptf_cos_kernel(r);
  return ((q + 1) & 2) ? -v : v;
}
PT_FP float pt_tan(float x)
{
  if(ptf_abs(x) < 2.44140625e-4f)
    return x;
  float r;
  int   q;
  ptf_sincos_reduce(x, &r, &q);
  float s = ptf_sin_kernel(r), c = ptf_cos_kernel(r);
  return (q & 1) ? -c / s : s / c;
}

/* ---- asin / acos -------------------------------------------------------------------------------------
 * asin on [0, 0.5] by a minimax polynomial; larger arguments through asin(x) = pi/2 - 2 asin(sqrt((1-x)/2)). */
PT_FP float ptf_asin_poly(float x, float z) /* asin(x) for |x| <= 0.5, z = x*x */
{
  float p = ptf_fma(3.3921066672e-02f, z, 1.7005795613e-02f);
  p       = ptf_fma(p, z, 3.1131917611e-02f);
  p       = ptf_fma(p, z, 4.4596623629e-02f);
  p       = ptf_fma(p, z, 7.5001031160e-02f);
  p       = ptf_fma(p, z, 1.6666665673e-01f);
  return ptf_fma(x * z, p, x);
}
PT_FP float pt_asin(float x)
{
  float a = ptf_abs(x);
  if(!(a <= 1.0f))
    return ptf_nan();
  float r;
  if(a > 0.5f)
  {
    float z = 0.5f * (1.0f - a);
    float s = sqrtf(z);
    r       = 1.57079637050628662f - 2.0f * ptf_asin_poly(s, z);
  }
  else
    r = ptf_asin_poly(a, a * a);
  return ptf_copysign(r, x);
}
PT_FP float pt_acos(float x)
{
  float a = ptf_abs(x);
  if(!(a <= 1.0f))
    return ptf_nan();
  if(a > 0.5f)
  {
    float z = 0.5f * (1.0f - a);
    float s = sqrtf(z);
    float r = 2.0f * ptf_asin_poly(s, z);
    /* pi - r with pi = hi + lo keeps the result within an ulp next to -1 */
    return (x < 0.0f) ? (3.14159274101257324f - r) + -8.74227765734758577e-08f : r;
  }
  /* pi/2 - asin(x), pi/2 = hi + lo */
  float as = ptf_asin_poly(x, x * x);
  return (1.57079637050628662f - as) + -4.37113882867379288e-08f;
}

/* ---- atan2 -------------------------------------------------------------------------------------------
 * atan on [0, inf) folded to [0, tan(pi/8)] (Cephes scheme), quadrant fixed up afterwards.  atan2(0,0) = 0. */
PT_FP float ptf_atan_pos(float x) /* x >= 0 */
{
  float y0;
  if(x > 2.414213562373095f)
  {
    y0 = 1.57079637050628662f;
    x  = -1.0f / x;
  }
  else if(x > 0.4142135623730950f)
  {
    y0 = 0.785398185253143311f;
    x  = (x - 1.0f) / (x + 1.0f);
  }
  else
    y0 = 0.0f;
  float z = x * x;
  float p = ptf_fma(-6.4647629857e-02f, z, 1.0748238862e-01f);
  p       = ptf_fma(p, z, -1.4264450967e-01f);
  p       = ptf_fma(p, z, 1.9999557734e-01f);
  p       = ptf_fma(p, z, -3.3333331347e-01f);
  return y0 + ptf_fma(p * z, x, x);
}
PT_FP float pt_atan(float x)
{
  if(ptf_isnan(x))
    return x;
  return ptf_copysign(ptf_atan_pos(ptf_abs(x)), x);
}
PT_FP float pt_atan2(float y, float x)
{
  if(ptf_isnan(x) || ptf_isnan(y))
    return ptf_nan();
  float ax = ptf_abs(x), ay = ptf_abs(y);
  float r;
  if(ay == 0.0f && ax == 0.0f)
    r = 0.0f;
  else if(ptf_isinf(ax) && ptf_isinf(ay))
    r = 0.785398185253143311f;
  else if(ax >= ay)
    r = ptf_atan_pos(ay / ax);
  else
    r = 1.57079637050628662f - ptf_atan_pos(ax / ay);  /* atan(t) = pi/2 - atan(1/t): keeps the quotient <= 1 */
  if(ptf_bits(x) & 0x80000000u)
    r = 3.14159274101257324f - r;
  return ptf_copysign(r, y);
}

/* ---- exp ---------------------------------------------------------------------------------------------- */
PT_FP float ptf_scale2(float p, int k) /* p * 2^k, k in [-280, 280], correct through the denormal range */
{
  int k1 = k / 2, k2 = k - k1;
  if(k1 < -126) k1 = -126;
  if(k1 > 127) k1 = 127;
  k2 = k - k1;
  if(k2 < -126)
  {
    p *= ptf_pow2i(-126);
    k2 += 126;
    if(k2 < -126) k2 = -126;
  }
  if(k2 > 127)
  {
    p *= ptf_pow2i(127);
    k2 -= 127;
    if(k2 > 127) k2 = 127;
  }
  return (p * ptf_pow2i(k1)) * ptf_pow2i(k2);
}
PT_FP float ptf_exp2_poly(float r) /* 2^r on [-0.5, 0.5] */
{
  float p = ptf_fma(1.5310082745e-05f, r, 1.5461447765e-04f);
  p       = ptf_fma(p, r, 1.3333454262e-03f);
  p       = ptf_fma(p, r, 9.6180569381e-03f);
  p       = ptf_fma(p, r, 5.5504109710e-02f);
  p       = ptf_fma(p, r, 2.4022650719e-01f);
  p       = ptf_fma(p, r, 6.9314718246e-01f);
  return ptf_fma(p, r, 1.0f);
}
PT_FP float pt_exp(float x)
{
  if(ptf_isnan(x))
    return x;
  if(x > 88.7228394f)
    return ptf_inf();
  if(x < -104.0f)
    return 0.0f;
  float kf = rintf(x * 1.44269502162933350f);
  float r  = ptf_fma(-kf, 6.93145751953125e-1f, x);        /* ln2 high (exact product for |k| < 2^11) */
  r        = ptf_fma(-kf, 1.42860682030941723e-6f, r);     /* ln2 low */
  /* e^r on [-ln2/2, ln2/2] */
  float p = ptf_fma(1.9908919057e-04f, r, 1.3934550807e-03f);
  p       = ptf_fma(p, r, 8.3332844079e-03f);
  p       = ptf_fma(p, r, 4.1666455567e-02f);
  p       = ptf_fma(p, r, 1.6666667163e-01f);
  p       = ptf_fma(p, r, 5.0000000e-01f);
  p       = ptf_fma(p * r, r, r);
  return ptf_scale2(p + 1.0f, (int)kf);
}

/* ---- log -----------------------------------------------------------------------------------------------
 * x = m * 2^e, m in [sqrt(1/2), sqrt(2)); log(m) by the fdlibm scheme (s = f/(2+f)). */
PT_FP float pt_log(float x)
{
  uint32_t ix = ptf_bits(x);
  if((ix & 0x7fffffffu) == 0u)
    return -ptf_inf();
  if(ix & 0x80000000u)
    return ptf_nan();
  if(ix >= 0x7f800000u)
    return x; /* +inf, nan */
  int e = 0;
  if(ix < 0x00800000u)
  {
    x *= 33554432.0f; /* 2^25 */
    ix = ptf_bits(x);
    e  = -25;
  }
  ix += 0x3f800000u - 0x3f3504f3u;
  e += (int)(ix >> 23) - 127;
  ix      = (ix & 0x007fffffu) + 0x3f3504f3u;
  float m = ptf_from_bits(ix);
  float f = m - 1.0f;
  float s = f / (2.0f + f);
  float z = s * s;
  float w = z * z;
  float t1 = w * ptf_fma(w, 0.24279078841f, 0.40000972152f);
  float t2 = z * ptf_fma(w, 0.28498786688f, 0.66666662693f);
  float R    = t2 + t1;
  float hfsq = 0.5f * f * f;
  float dk   = (float)e;
  return dk * 6.9313812256e-01f - ((hfsq - (s * (hfsq + R) + dk * 9.0580006145e-06f)) - f);
}

/* ---- pow -----------------------------------------------------------------------------------------------
 * x^y = 2^(y * log2 x) with log2 x carried as an unevaluated sum hi + lo (about 2^-33 relative), so the
 * product with y keeps the result within ~1 ulp.  Negative bases follow C's powf (integer exponents only). */
PT_FP void ptf_log2_ext(float x, float* hi, float* lo) /* x finite, > 0 */
{
  uint32_t ix = ptf_bits(x);
  int      e  = 0;
  if(ix < 0x00800000u)
  {
    x *= 33554432.0f;
    ix = ptf_bits(x);
    e  = -25;
  }
  ix += 0x3f800000u - 0x3f3504f3u;
  e += (int)(ix >> 23) - 127;
  ix      = (ix & 0x007fffffu) + 0x3f3504f3u;
  float m = ptf_from_bits(ix);
  /* u = (m-1)/(m+1) as uh + ul */
  float num = m - 1.0f;             /* exact */
  float dh  = m + 1.0f;
  float dl  = m - (dh - 1.0f);      /* exact rounding error of dh */
  float uh  = num / dh;
  float rem = ptf_fma(-uh, dh, num); /* exact remainder against dh */
  rem       = ptf_fma(-uh, dl, rem);
  float ul  = rem / dh;
  /* log(m) = 2u + 2u^3 (1/3 + u^2/5 + u^4/7 + u^6/9 + u^8/11 + u^10/13) */
  float z = uh * uh;
  float q = ptf_fma(0.15384615384615385f, z, 0.18181818181818182f);
  q       = ptf_fma(q, z, 0.22222222222222222f);
  q       = ptf_fma(q, z, 0.28571428571428571f);
  q       = ptf_fma(q, z, 0.4f);
  q       = ptf_fma(q, z, 0.66666666666666667f);
  /* tail = 2 ul + u^3 q, including the rounding error of 1/3 (0.6666667 - 2/3 = 1.987e-8) */
  float u3   = uh * z;
  float tail = ptf_fma(u3, q, 2.0f * ul);
  tail       = ptf_fma(u3, -1.98682149e-08f, tail);
  float lh   = 2.0f * uh;
  /* (lh + tail) * log2(e), log2(e) = 1.44269502162933350 + 1.92596299112661746e-08 */
  float ph = lh * 1.44269502162933350f;
  float pl = ptf_fma(lh, 1.44269502162933350f, -ph);
  pl       = ptf_fma(lh, 1.92596299112661746e-08f, pl);
  pl       = ptf_fma(tail, 1.44269502162933350f, pl);
  /* add the exponent: e + ph exactly split (|ph| <= 0.5 < ulp issues handled by two-sum) */
  float fe = (float)e;
  float sh = fe + ph;
  float bb = sh - fe;
  float sl = (fe - (sh - bb)) + (ph - bb);
  *hi      = sh;
  *lo      = sl + pl;
}
PT_FP float pt_pow(float x, float y)
{
  if(y == 0.0f || x == 1.0f)
    return 1.0f;
  if(ptf_isnan(x) || ptf_isnan(y))
    return ptf_nan();
  float sign = 1.0f;
  float ax   = ptf_abs(x);
  int   yint = 0; /* 0: not an integer, 1: odd, 2: even */
  float ay   = ptf_abs(y);
  if(ay >= 16777216.0f)
    yint = 2;
  else if(ay >= 1.0f)
  {
    float fl = floorf(ay);
    if(fl == ay)
      yint = (((int)fl) & 1) ? 1 : 2;
  }
  if(ptf_bits(x) & 0x80000000u)
  {
    if(ax == 0.0f || ptf_isinf(ax))
      sign = (yint == 1) ? -1.0f : 1.0f;
    else if(yint == 0)
      return ptf_nan();
    else if(yint == 1)
      sign = -1.0f;
  }
  if(ax == 0.0f)
    return sign * ((y < 0.0f) ? ptf_inf() : 0.0f);
  if(ptf_isinf(ax))
    return sign * ((y < 0.0f) ? 0.0f : ptf_inf());
  if(ptf_isinf(ay))
  {
    if(ax == 1.0f)
      return 1.0f;
    return ((ax > 1.0f) == (y > 0.0f)) ? ptf_inf() : 0.0f;
  }
  float lh, ll;
  ptf_log2_ext(ax, &lh, &ll);
  float ph = y * lh;
  float pl = ptf_fma(y, lh, -ph);
  pl       = ptf_fma(y, ll, pl);
  if(!(ph < 129.0f))
    return sign * ptf_inf();
  if(ph < -151.0f)
    return sign * 0.0f;
  float kf = rintf(ph);
  float r  = (ph - kf) + pl;
  return sign * ptf_scale2(ptf_exp2_poly(r), (int)kf);
}

#endif /* PT_FPMATH_H */
