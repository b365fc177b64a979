// Packet traversal for coherent rays (the primary rays of an 8x8 pixel block = one wavefront).
//
// All 64 lanes walk ONE traversal: the node / triangle records are fetched with scalar loads into SGPRs (one fetch per
// wave instead of one per lane, no per-lane address arithmetic, no vector-memory latency in the dependent chain), the
// stack is a single wave-level array and the child order is decided once per wave on the scalar unit.  What stays per
// lane is the arithmetic that defines the result: the slab tests against the lane's own ray and best-hit bound, the
// ray/triangle test and the candidate rules of traverse<TM_CLOSEST> (pt_trace.h) -- so hits, alpha counts and flags
// are those of the per-lane traversal (the trace contract is independent of the visiting order).
//
// The fused slab test picks near / far planes by the sign of the ray direction; a packet shares that choice only if
// all its lanes agree on the three signs.  The (rare) packets that do not -- the block containing the optical axis --
// fall back to the per-lane traversal.
#pragma once
#include "pt_trace.h"

#define PACKET_STACK 128  // wave-level stack entries (4-wide nodes push at most 3 per visit; overflow is counted like the per-lane one)

typedef float    pt_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t pt_u4v __attribute__((ext_vector_type(4)));
typedef const pt_f4v __attribute__((address_space(4)))* pt_c4ptr;   // constant address space: uniform addresses become s_load
typedef const pt_u4v __attribute__((address_space(4)))* pt_cu4ptr;

PT_DEV float4 sload4(const void* base, uint32_t byteOff)
{
  const pt_f4v v = *((pt_c4ptr)(unsigned long long)(reinterpret_cast<const char*>(base) + byteOff));
  return make_float4(v.x, v.y, v.z, v.w);
}
PT_DEV uint4 sloadu4(const void* base, uint32_t byteOff)
{
  const pt_u4v v = *((pt_cu4ptr)(unsigned long long)(reinterpret_cast<const char*>(base) + byteOff));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// `valid`: the lane carries a ray.  wstack: PACKET_STACK dwords of LDS shared by the wave.  Returns false when the packet
// is not sign-coherent (nothing was traversed; the caller runs the per-lane traversal instead).
// SHADOW: any-hit semantics of traverse<TM_SHADOW> -- an opaque hit inside (0, tmax) ends the lane (`opaqueHit`), the bound stays
// tmax because an opaque occluder may lie behind the nearest non-opaque candidate.
template <bool SHADOW>
PT_DEV bool traverse_packet(const DeviceScene& S, bool valid, f3 o, f3 d, float tmax, uint32_t* wstack, RayHit& best, bool& opaqueHit, Counters* counters)
{
  const RayBox rb = make_raybox(o, d);
  opaqueHit       = false;
  const unsigned long long vm = __ballot(valid);
  const unsigned long long sx = __ballot(valid && rb.idir.x < 0.0f), sy = __ballot(valid && rb.idir.y < 0.0f), sz = __ballot(valid && rb.idir.z < 0.0f);
  if((sx != 0ull && sx != vm) || (sy != 0ull && sy != vm) || (sz != 0ull && sz != vm))
    return false;
  best.slot = BVH_NONE; best.t = tmax; best.w = 0xffffffffu; best.flags = 0; best.count = 0;
  best.zeroMaxT = best.zeroMaxT2 = best.zeroMaxT3 = -1.0f;
  best.u = best.v = 0.0f;
  if(S.numTris == 0 || vm == 0ull)
    return true;
  const uint32_t offX = sx ? 48u : 0u, offY = sy ? 48u : 0u, offZ = sz ? 48u : 0u;  // wave-uniform

  uint32_t cur = 0;
  int      sp  = 0;
#ifdef PT_HIST
  uint32_t hInner = 0, hLeaf = 0;
#endif
  for(;;)
  {
#ifdef PT_HIST
    if(cur & BVH_LEAF) ++hLeaf; else ++hInner;
#endif
    if(!(cur & BVH_LEAF))
    {
      const uint32_t at = (cur & BVH_SLOT_MASK) << 7;
      const float4   px = sload4(S.wide, at + offX), qx = sload4(S.wide, at + 48u - offX);
      const float4   py = sload4(S.wide, at + 16u + offY), qy = sload4(S.wide, at + 64u - offY);
      const float4   pz = sload4(S.wide, at + 32u + offZ), qz = sload4(S.wide, at + 80u - offZ);
      const uint4    ch = sloadu4(S.wide, at + 96u);
      const float    pxs[4] = {px.x, px.y, px.z, px.w}, qxs[4] = {qx.x, qx.y, qx.z, qx.w};
      const float    pys[4] = {py.x, py.y, py.z, py.w}, qys[4] = {qy.x, qy.y, qy.z, qy.w};
      const float    pzs[4] = {pz.x, pz.y, pz.z, pz.w}, qzs[4] = {qz.x, qz.y, qz.z, qz.w};
      const uint32_t cc[4]  = {ch.x, ch.y, ch.z, ch.w};
      float          key[4];   // wave-uniform ordering key: entry distance of the first lane that hits the child
      uint32_t       cid[4];
      int            nh = 0;
#pragma unroll
      for(int k = 0; k < 4; ++k)
      {
        const float nr = fmaxf(fmaxf(__builtin_fmaf(pxs[k], rb.idir.x, rb.nlo.x), __builtin_fmaf(pys[k], rb.idir.y, rb.nlo.y)), fmaxf(__builtin_fmaf(pzs[k], rb.idir.z, rb.nlo.z), 0.0f)) * 0.9999996f;
        const float fr = fminf(fminf(__builtin_fmaf(qxs[k], rb.idir.x, rb.nhi.x), __builtin_fmaf(qys[k], rb.idir.y, rb.nhi.y)), fminf(__builtin_fmaf(qzs[k], rb.idir.z, rb.nhi.z), SHADOW ? tmax : best.t)) * 1.0000004f;
        const unsigned long long hm = (cc[k] != BVH_NONE) ? __ballot(valid && nr <= fr) : 0ull;
        if(hm)
        {
          key[nh] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(nr), __ffsll((long long)hm) - 1));
          cid[nh] = cc[k];
          ++nh;
        }
      }
      if(nh)
      {
        // scalar insertion sort, nearest first (nh <= 4, all operands wave-uniform)
#pragma unroll
        for(int i = 1; i < 4; ++i)
          for(int j = i; j > 0 && j < nh && key[j] < key[j - 1]; --j)
          {
            const float    tk = key[j]; key[j] = key[j - 1]; key[j - 1] = tk;
            const uint32_t tc = cid[j]; cid[j] = cid[j - 1]; cid[j - 1] = tc;
          }
        for(int i = nh - 1; i >= 1; --i)
        {
          if(sp < PACKET_STACK)
            wstack[sp++] = cid[i];
          else if((threadIdx.x & 63) == 0)
            atomicAdd(&counters->stackOverflow, 1u);
        }
        cur = cid[0];
        continue;
      }
    }
    else
    {
      const uint32_t slot = cur & BVH_SLOT_MASK;
      const float4   t0 = sload4(S.tris, slot * 48u), t1 = sload4(S.tris, slot * 48u + 16u), t2 = sload4(S.tris, slot * 48u + 32u);
      TriRec         tr;
      tr.p0w = t0; tr.e1n = t1; tr.e2p = t2;
      const uint32_t wbits = __float_as_uint(t0.w);
      const uint32_t flags = wbits >> 29;
      const bool     opq   = (flags & TRI_OPAQUE) != 0;
      float          t, u, v;
      if(valid && tri_test(tr, flags, o, d, t, u, v) && t > 0.0f && t < tmax)
      {
        const uint32_t w = wbits & TRI_INDEX_MASK;
        if(SHADOW && opq)
        {
          opaqueHit = true;
          valid     = false;  // the lane leaves the packet
        }
        else if(best.slot == BVH_NONE || key_less(t, w, best.t, best.w & TRI_INDEX_MASK))
        {
          bool certain = opq;
          if(!opq)
          {
            const float op = opacity_class(S, S.alphaRecs[slot], u, v);
            certain        = op >= 1.0f;
            if(!certain)
            {
              best.flags |= (op <= 0.0f) ? TF_SAW_ZERO : TF_SAW_FRAC;
              if(op <= 0.0f)
              {
                best.count++;
                note_zero_candidate(t, best.zeroMaxT, best.zeroMaxT2, best.zeroMaxT3);
              }
            }
          }
          if(certain)
          {
            best.t = t; best.u = u; best.v = v; best.slot = slot; best.w = wbits;
          }
        }
      }
    }
    if(sp == 0 || (SHADOW && __ballot(valid) == 0ull))
      break;
    cur = __builtin_amdgcn_readfirstlane(wstack[--sp]);
  }
#ifdef PT_HIST
  if((threadIdx.x & 63) == 0)
  {
    atomicAdd(&g_hist[7][SHADOW ? 4 : 0], (unsigned long long)hInner); atomicAdd(&g_hist[7][SHADOW ? 5 : 1], (unsigned long long)hLeaf); atomicAdd(&g_hist[7][SHADOW ? 6 : 2], 1ull);
    atomicAdd(&g_hist[7][SHADOW ? 7 : 3], (unsigned long long)__popcll(vm));
  }
#endif
  return true;
}

PT_DEV bool traverse_packet_closest(const DeviceScene& S, bool valid, f3 o, f3 d, uint32_t* wstack, RayHit& best, Counters* counters)
{
  bool dummy;
  return traverse_packet<false>(S, valid, o, d, PT_INFINITY, wstack, best, dummy, counters);
}

#if PT_BVH_WIDTH != 2
// ---- the same for the two-level structure (round 4) ---------------------------------------------------------------------------------------------
// The packet walks the TLAS with its world-space ray constants, enters an instance as a whole (TLAS leaf, instance record and BLAS nodes all through
// scalar loads: one instance at a time per wavefront), continues in the instance's object space with per-lane constants from enter_instance(), and
// is back at TLAS level when the wave-level stack has shrunk to the depth it had on entry -- pt_trace.h traverse<.., TWO> with one traversal per
// wavefront.  Leaves are tested in world space on the T1 triangle rebuilt from the instance matrix (world_tri), so hits are bit-identical to every
// other walk.  Inside an instance the lanes' direction signs may disagree although they agreed in world space (a rotated instance seen near one of
// its axes): such an instance visit runs the sign-free form of the slab test (both planes of a slab per lane, min / max) instead of picking near
// and far planes by address; only a disagreement in WORLD space sends the packet to the per-lane machine, as in the flat kernel.
PT_DEV bool traverse_packet_two(const DeviceScene& S, bool valid, f3 o, f3 d, uint32_t* wstack, RayHit& best, Counters* counters)
{
  const RayBox             rbW = make_raybox(o, d);
  const unsigned long long vm  = __ballot(valid);
  {
    const unsigned long long sx = __ballot(valid && rbW.idir.x < 0.0f), sy = __ballot(valid && rbW.idir.y < 0.0f), sz = __ballot(valid && rbW.idir.z < 0.0f);
    if((sx != 0ull && sx != vm) || (sy != 0ull && sy != vm) || (sz != 0ull && sz != vm))
      return false;
  }
  best.slot = BVH_NONE; best.t = PT_INFINITY; best.w = 0xffffffffu; best.flags = 0; best.count = 0;
  best.zeroMaxT = best.zeroMaxT2 = best.zeroMaxT3 = -1.0f;
  best.u = best.v = 0.0f;
  if(S.numTris == 0 || vm == 0ull)
    return true;
  // wave-uniform state of the space the packet is in
  RayBox      rb    = rbW;
  bool        coh   = true;                     // the lanes agree on the direction signs in the current space
  uint32_t    offX  = __ballot(valid && rbW.idir.x < 0.0f) ? 48u : 0u, offY = __ballot(valid && rbW.idir.y < 0.0f) ? 48u : 0u, offZ = __ballot(valid && rbW.idir.z < 0.0f) ? 48u : 0u;
  const uint32_t wOffX = offX, wOffY = offY, wOffZ = offZ;
  const void* nodes = S.tlas;
  InstCtx     ic{BVH_NONE, 0, 0u};
  uint32_t    cur = 0, guard = 0;
  int         sp  = 0;
#ifdef PT_HIST
  uint32_t hTop = 0, hEnter = 0, hInner = 0, hLeaf = 0, hMixed = 0, hMerged = 0;
#endif
  for(;;)
  {
    if(!(cur & BVH_LEAF))
    {
      if(++guard > PT_TWO_GUARD)
      {
        if((threadIdx.x & 63) == 0)
          atomicAdd(&counters->stackOverflow, 1u);
        break;
      }
#ifdef PT_HIST
      if(ic.inst == BVH_NONE) ++hTop; else ++hInner;
#endif
      const uint32_t at = (cur & BVH_SLOT_MASK) << 7;
      // coherent: near planes at +off, far planes at +48-off per axis; else: lower planes (p*) and upper planes (q*)
      const float4   px = sload4(nodes, at + (coh ? offX : 0u)), qx = sload4(nodes, at + 48u - (coh ? offX : 0u));
      const float4   py = sload4(nodes, at + 16u + (coh ? offY : 0u)), qy = sload4(nodes, at + 64u - (coh ? offY : 0u));
      const float4   pz = sload4(nodes, at + 32u + (coh ? offZ : 0u)), qz = sload4(nodes, at + 80u - (coh ? offZ : 0u));
      const uint4    ch = sloadu4(nodes, at + 96u);
      const float    pxs[4] = {px.x, px.y, px.z, px.w}, qxs[4] = {qx.x, qx.y, qx.z, qx.w};
      const float    pys[4] = {py.x, py.y, py.z, py.w}, qys[4] = {qy.x, qy.y, qy.z, qy.w};
      const float    pzs[4] = {pz.x, pz.y, pz.z, pz.w}, qzs[4] = {qz.x, qz.y, qz.z, qz.w};
      const uint32_t cc[4]  = {ch.x, ch.y, ch.z, ch.w};
      float          key[4];
      uint32_t       cid[4];
      int            nh = 0;
#pragma unroll
      for(int k = 0; k < 4; ++k)
      {
        float nr, fr;
        if(coh)
        {
          nr = fmaxf(fmaxf(__builtin_fmaf(pxs[k], rb.idir.x, rb.nlo.x), __builtin_fmaf(pys[k], rb.idir.y, rb.nlo.y)), fmaxf(__builtin_fmaf(pzs[k], rb.idir.z, rb.nlo.z), 0.0f));
          fr = fminf(fminf(__builtin_fmaf(qxs[k], rb.idir.x, rb.nhi.x), __builtin_fmaf(qys[k], rb.idir.y, rb.nhi.y)), fminf(__builtin_fmaf(qzs[k], rb.idir.z, rb.nhi.z), best.t));
        }
        else
        {  // sign-free: the near parameter of a slab is the smaller of its two planes' (biased towards "hit" like the fused form), the far one the larger
          const float nx = fminf(__builtin_fmaf(pxs[k], rb.idir.x, rb.nlo.x), __builtin_fmaf(qxs[k], rb.idir.x, rb.nlo.x)), fx = fmaxf(__builtin_fmaf(pxs[k], rb.idir.x, rb.nhi.x), __builtin_fmaf(qxs[k], rb.idir.x, rb.nhi.x));
          const float ny = fminf(__builtin_fmaf(pys[k], rb.idir.y, rb.nlo.y), __builtin_fmaf(qys[k], rb.idir.y, rb.nlo.y)), fy = fmaxf(__builtin_fmaf(pys[k], rb.idir.y, rb.nhi.y), __builtin_fmaf(qys[k], rb.idir.y, rb.nhi.y));
          const float nz = fminf(__builtin_fmaf(pzs[k], rb.idir.z, rb.nlo.z), __builtin_fmaf(qzs[k], rb.idir.z, rb.nlo.z)), fz = fmaxf(__builtin_fmaf(pzs[k], rb.idir.z, rb.nhi.z), __builtin_fmaf(qzs[k], rb.idir.z, rb.nhi.z));
          nr = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
          fr = fminf(fminf(fx, fy), fminf(fz, best.t));
        }
        nr *= 0.9999996f;
        fr *= 1.0000004f;
        const unsigned long long hm = (cc[k] != BVH_NONE) ? __ballot(valid && nr <= fr) : 0ull;
        if(hm)
        {
          key[nh] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(nr), __ffsll((long long)hm) - 1));
          cid[nh] = cc[k];
          ++nh;
        }
      }
      if(nh)
      {
#pragma unroll
        for(int i = 1; i < 4; ++i)
          for(int j = i; j > 0 && j < nh && key[j] < key[j - 1]; --j)
          {
            const float    tk = key[j]; key[j] = key[j - 1]; key[j - 1] = tk;
            const uint32_t tc = cid[j]; cid[j] = cid[j - 1]; cid[j - 1] = tc;
          }
        for(int i = nh - 1; i >= 1; --i)
        {
          if(sp < PACKET_STACK)
            wstack[sp++] = cid[i];
          else if((threadIdx.x & 63) == 0)
            atomicAdd(&counters->stackOverflow, 1u);
        }
        cur = cid[0];
        continue;
      }
    }
    else if(ic.inst == BVH_NONE)
    {  // TLAS leaf: the whole packet enters the instance
      const uint32_t lslot = cur & BVH_SLOT_MASK;
      const uint4    a = sloadu4(S.tlasLeaves, lslot * 32u), b = sloadu4(S.tlasLeaves, lslot * 32u + 16u);
      TlasLeaf       tl;
      tl.inst = a.x; tl.nodeBase = a.y; tl.wflags = a.z; tl._pad0 = a.w; tl.padC0 = __uint_as_float(b.x); tl.padC1 = __uint_as_float(b.y); tl._pad1[0] = b.z; tl._pad1[1] = b.w;
      ic = InstCtx{tl.inst, sp, tl.wflags};
      if(tl.inst != PT_INST_MERGED)
      {
        rb = enter_instance(S, tl, o, d);
        const unsigned long long sx = __ballot(valid && rb.idir.x < 0.0f), sy = __ballot(valid && rb.idir.y < 0.0f), sz = __ballot(valid && rb.idir.z < 0.0f);
        const unsigned long long am = __ballot(valid);  // (SHADOW lanes never leave here; kept for symmetry with the flat kernel)
        coh  = !((sx != 0ull && sx != am) || (sy != 0ull && sy != am) || (sz != 0ull && sz != am));
        offX = sx ? 48u : 0u; offY = sy ? 48u : 0u; offZ = sz ? 48u : 0u;
      }
#ifdef PT_HIST
      ++hEnter;
      if(tl.inst == PT_INST_MERGED) ++hMerged; else if(!coh) ++hMixed;
#endif
      nodes = S.wide;
      cur   = tl.nodeBase;
      continue;
    }
    else
    {
      const uint32_t slot = cur & BVH_SLOT_MASK;
      const float4   t0 = sload4(S.tris, slot * 48u), t1 = sload4(S.tris, slot * 48u + 16u), t2 = sload4(S.tris, slot * 48u + 32u);
      TriRec         tr;
      tr.p0w = t0; tr.e1n = t1; tr.e2p = t2;
#ifdef PT_HIST
      ++hLeaf;
#endif
      if(ic.inst != PT_INST_MERGED)
        tr = world_tri(S, ic, tr);
      const uint32_t wbits = __float_as_uint(tr.p0w.w);
      const uint32_t flags = wbits >> 29;
      const bool     opq   = (flags & TRI_OPAQUE) != 0;
      float          t, u, v;
      if(valid && tri_test(tr, flags, o, d, t, u, v) && t > 0.0f && t < PT_INFINITY)
      {
        const uint32_t w = wbits & TRI_INDEX_MASK;
        if(best.slot == BVH_NONE || key_less(t, w, best.t, best.w & TRI_INDEX_MASK))
        {
          bool certain = opq;
          if(!opq)
          {
            const float op = opacity_class(S, S.alphaRecs[slot], u, v);
            certain        = op >= 1.0f;
            if(!certain)
            {
              best.flags |= (op <= 0.0f) ? TF_SAW_ZERO : TF_SAW_FRAC;
              if(op <= 0.0f)
              {
                best.count++;
                note_zero_candidate(t, best.zeroMaxT, best.zeroMaxT2, best.zeroMaxT3);
              }
            }
          }
          if(certain)
          {
            best.t = t; best.u = u; best.v = v; best.slot = slot; best.w = wbits;
          }
        }
      }
    }
    // pop; an instance is left when the stack is back at its entry depth
    if(ic.inst != BVH_NONE && sp == ic.spBase)
    {
      ic.inst = BVH_NONE;
      rb      = rbW;
      coh     = true;
      offX = wOffX; offY = wOffY; offZ = wOffZ;
      nodes   = S.tlas;
    }
    if(sp == 0)
      break;
    cur = __builtin_amdgcn_readfirstlane(wstack[--sp]);
  }
#ifdef PT_HIST
  if((threadIdx.x & 63) == 0)
  {
    atomicAdd(&g_hist[7][8], (unsigned long long)hTop); atomicAdd(&g_hist[7][9], (unsigned long long)hEnter); atomicAdd(&g_hist[7][10], (unsigned long long)hInner);
    atomicAdd(&g_hist[7][11], (unsigned long long)hLeaf); atomicAdd(&g_hist[7][12], 1ull); atomicAdd(&g_hist[7][13], (unsigned long long)hMixed);
    atomicAdd(&g_hist[7][14], (unsigned long long)hMerged);
  }
#endif
  return true;
}
#endif
