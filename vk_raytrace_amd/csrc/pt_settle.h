// Settling ONE ray per lane: the closest-hit ray (trace contract T5) and the shadow ray (T6) of a path, from the path state in HBM to the hit
// record / the verdict, through the two-pass stochastic alpha of pt_trace.h with its exact key-ordered fallback.  These are the bodies k_tail
// runs per lane (pt_render.hip); the exact-fallback kernels k_closest_x / k_shadow_x spell the same steps out per stage.
// Plain inline functions of (scene, path state, slot): tests/cpp/trace_host.cpp compiles them for the host and holds them, ray by ray, to
// the contract's exact loop -- hits AND the RNG state afterwards (tests/test_trace_host.py).
#pragma once
#include "pt_internal.h"
#include "pt_trace.h"

// The hit record names the triangle by its leaf slot (flat structure: S.tris[slot] carries instance and primitive) or, with the two-level
// structure (`two`: a BLAS leaf is shared by all instances of its mesh), by its world triangle index (k_shade: instance_of_world_tri).
template <class RB>
PT_DEV void store_hit(const RB& rb, uint32_t slot, uint32_t bslot, uint32_t bw, bool two, float t, float u, float v)
{
  if(bslot == BVH_NONE)
    rb.ps.hit[slot] = make_float4(PT_INFINITY, __uint_as_float(BVH_NONE), 0.f, 0.f);
  else
    rb.ps.hit[slot] = make_float4(t, __uint_as_float(two ? (bw & TRI_INDEX_MASK) : bslot), u, v);
}

// The exact key-ordered loops (trace contract T5 / T6; k_closest_x / k_shadow_x spell the same steps out per stage): what a ray falls back to when the
// two-pass scheme cannot settle it -- a candidate of fractional opacity in front of the hit, or a rejected-candidate draw of exactly 0.0.
// `seed`: the path's RNG state before the ray's first draw; closest: hit record and the state afterwards go to the path state.
template <bool TWO, class RB>
PT_DEV void settle_closest_exact(const DeviceScene& S, const RB& rb, uint32_t slot, f3 o, f3 d, uint32_t seed, uint32_t* stack, uint32_t& nAlpha)
{
  RayHit   h;
  bool     dummy;
  float    tPrev = 0.0f;
  uint32_t wPrev = 0xffffffffu;
  for(;;)
  {
    traverse<TM_RAW_ALL, TWO>(S, o, d, PT_INFINITY, tPrev, wPrev, 0u, stack, h, dummy, rb.counters);
    if(h.slot == BVH_NONE || ((h.w >> 29) & TRI_OPAQUE))
      break;
    ++nAlpha;
    if(alpha_test(S, h.slot, h.u, h.v, seed))
      break;
    tPrev = h.t;
    wPrev = h.w & TRI_INDEX_MASK;
  }
  store_hit(rb, slot, h.slot, h.w, TWO, h.t, h.u, h.v);
  rb.ps.rayD[slot].w = __uint_as_float(seed);
}
// shadow: returns inShadow; `seed` comes back as the path's state afterwards (RTX flavour: untouched, the any-hit shader draws from a copy)
template <bool TWO>
PT_DEV bool settle_shadow_exact(const DeviceScene& S, f3 o, f3 d, float maxDist, int variant, uint32_t& seed, uint32_t* stack, uint32_t& nAlpha, Counters* counters)
{
  const uint32_t seed0 = seed;
  RayHit         h;
  bool           dummy, inShadow = false;
  float          tPrev = 0.0f;
  uint32_t       wPrev = 0xffffffffu;
  for(;;)
  {
    traverse<TM_RAW_ALL, TWO>(S, o, d, maxDist, tPrev, wPrev, 0u, stack, h, dummy, counters);
    if(h.slot == BVH_NONE)
      break;
    if((h.w >> 29) & TRI_OPAQUE)
    {
      inShadow = true;
      break;
    }
    ++nAlpha;
    if(alpha_test(S, h.slot, h.u, h.v, seed))
    {
      inShadow = true;
      break;
    }
    tPrev = h.t;
    wPrev = h.w & TRI_INDEX_MASK;
  }
  if(variant == PT_VARIANT_RTX)
    seed = seed0;
  return inShadow;
}

// the closest-hit ray of a path: hit record -> rb.ps.hit[slot], RNG state after the alpha draws -> rb.ps.rayD[slot].w
template <bool TWO>
PT_DEV void tail_closest(const DeviceScene& S, const RenderBuffers& rb, uint32_t slot, uint32_t* stack, uint32_t& nAlpha)
{
  const f3       o    = xyz(rb.ps.rayO[slot]);
  const float4   dw   = rb.ps.rayD[slot];
  const f3       d    = xyz(dw);
  uint32_t       seed = __float_as_uint(dw.w);
  RayHit         h;
  bool           dummy;
  traverse<TM_CLOSEST, TWO>(S, o, d, PT_INFINITY, 0.0f, 0xffffffffu, 0u, stack, h, dummy, rb.counters);
  bool       fallback = (h.flags & TF_SAW_FRAC) != 0;
  const bool passB    = !fallback && (h.flags & TF_SAW_ZERO) && !pass_a_settles(h.slot, h.t, h.zeroMaxT, h.zeroMaxT2, h.zeroMaxT3, h.count);
  uint32_t   nDraw    = h.count;
  if(passB)
  {
    RayHit c;
    traverse<TM_COUNT, TWO>(S, o, d, h.slot == BVH_NONE ? PT_INFINITY : h.t, 0.0f, 0xffffffffu, h.slot == BVH_NONE ? 0u : (h.w & TRI_INDEX_MASK), stack, c, dummy, rb.counters);
    fallback = (c.flags & TF_SAW_FRAC) != 0;
    nDraw    = c.count;
  }
  if(!fallback)
  {
    if(h.slot != BVH_NONE && !((h.w >> 29) & TRI_OPAQUE))
      ++nDraw;
    uint32_t s2 = seed;
    if(consume_rejected_draws(s2, nDraw))
    {
      store_hit(rb, slot, h.slot, h.w, TWO, h.t, h.u, h.v);
      if(nDraw)
        rb.ps.rayD[slot].w = __uint_as_float(s2);
      nAlpha += nDraw;
      return;
    }
  }
  settle_closest_exact<TWO>(S, rb, slot, o, d, seed, stack, nAlpha);
}

// the shadow ray of a path (k_shadow_s / k_shadow_x): returns inShadow, `seed` = the path's seed afterwards
template <bool TWO>
PT_DEV bool tail_shadow(const DeviceScene& S, const RenderBuffers& rb, uint32_t slot, uint32_t* stack, int variant, uint32_t& seed, uint32_t& nAlpha)
{
  seed                   = __float_as_uint(rb.ps.rayD[slot].w);
  const uint32_t seed0   = seed;
  const f3       o       = xyz(rb.ps.rayO[slot]);
  const f3       d       = xyz(rb.ps.neeDir[slot]);
  const float    maxDist = rb.ps.absorb[slot].w;
  bool           dummy;
  RayHit         h;
  traverse<TM_CLOSEST, TWO>(S, o, d, maxDist, 0.0f, 0xffffffffu, 0u, stack, h, dummy, rb.counters);
  bool       fallback = (h.flags & TF_SAW_FRAC) != 0;
  const bool passB    = !fallback && (h.flags & TF_SAW_ZERO) && !pass_a_settles(h.slot, h.t, h.zeroMaxT, h.zeroMaxT2, h.zeroMaxT3, h.count);
  uint32_t   nDraw    = h.count;
  if(passB)
  {
    RayHit c;
    traverse<TM_COUNT, TWO>(S, o, d, h.slot == BVH_NONE ? maxDist : h.t, 0.0f, 0xffffffffu, h.slot == BVH_NONE ? 0u : (h.w & TRI_INDEX_MASK), stack, c, dummy, rb.counters);
    fallback = (c.flags & TF_SAW_FRAC) != 0;
    nDraw    = c.count;
  }
  if(!fallback)
  {
    if(h.slot != BVH_NONE && !((h.w >> 29) & TRI_OPAQUE))
      ++nDraw;
    uint32_t s2 = seed;
    if(consume_rejected_draws(s2, nDraw))
    {
      seed = variant == PT_VARIANT_RTX ? seed : s2;  // RTX: the any-hit shader draws from a copy (traceray_rtx.glsl:54-55)
      nAlpha += nDraw;
      return h.slot != BVH_NONE;
    }
  }
  seed = seed0;  // (the two-pass attempt above left it untouched unless it settled the ray)
  return settle_shadow_exact<TWO>(S, o, d, maxDist, variant, seed, stack, nAlpha, rb.counters);
}

