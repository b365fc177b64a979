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

