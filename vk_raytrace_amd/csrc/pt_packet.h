// Packet traversal for coherent rays (the primary rays of an 8x8 pixel block = one wavefront).
//
// All 64 lanes walk ONE traversal of the 8-wide structure (pt_cwbvh.h): node and triangle records are fetched with scalar loads into SGPRs
// (one fetch per wave instead of one per lane, no per-lane address arithmetic, no vector-memory latency in the dependent chain), the node /
// group and the stack of postponed groups are wave-level (SGPRs + one LDS array), and the visiting order comes from the octant the
// packet shares.  What stays per lane is the arithmetic that defines the result: the box tests against the lane's own ray and best-hit bound,
// the ray/triangle test and the candidate rules of pass A (pt_trace.h lane_triangle<TM_CLOSEST>) -- so hits, alpha counts and flags are
// those of the per-lane traversal (the trace contract is independent of the visiting order).  A child is visited when ANY lane hits its box.
//
// The octant order needs all lanes to agree on the three direction signs.  The (rare) packets that do not -- the block containing the
// optical axis -- fall back to the per-lane traversal.
#pragma once
#include "pt_trace.h"

#define PACKET_STACK 64  // wave-level stack entries (one per node visit at most: as deep as the tree; overflow is counted like the per-lane one)

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

#define PK_CHILD(k)                                                                                                                                            \
  {                                                                                                                                                            \
    const float tn = fmaxf(fmaxf(__builtin_fmaf(cw_plane(cw_word(nx, (k) >> 1), (k) & 1), sx, blx), __builtin_fmaf(cw_plane(cw_word(ny, (k) >> 1), (k) & 1), sy, bly)), \
                           fmaxf(__builtin_fmaf(cw_plane(cw_word(nz, (k) >> 1), (k) & 1), sz, blz), 0.0f));                                                    \
    const float tf = fminf(fminf(__builtin_fmaf(cw_plane(cw_word(fx, (k) >> 1), (k) & 1), sx, bhx), __builtin_fmaf(cw_plane(cw_word(fy, (k) >> 1), (k) & 1), sy, bhy)), \
                           fminf(__builtin_fmaf(cw_plane(cw_word(fz, (k) >> 1), (k) & 1), sz, bhz), lim));                                                     \
    if(__ballot(valid && tn <= tf))                                                                                                                            \
      hits |= 1u << (k);                                                                                                                                       \
  }

// `valid`: the lane carries a ray.  wstack: PACKET_STACK x 3 words of LDS shared by the wave.  Returns false when the packet
// is not sign-coherent (nothing was traversed; the caller runs the per-lane traversal instead).
PT_DEV bool traverse_packet_closest(const DeviceScene& S, bool valid, f3 o, f3 d, uint32_t* wstack, RayHit& best, Counters* counters)
{
  TraceLane L;  // the per-lane part of the state: ray, best hit, alpha bookkeeping (group / stack fields unused: they are wave-level here)
  lane_begin(L, o, d, PT_INFINITY, false);
  const BoxRay&            R  = L.R;
  const unsigned long long vm = __ballot(valid);
  const unsigned long long sx_ = __ballot(valid && R.idir.x < 0.0f), sy_ = __ballot(valid && R.idir.y < 0.0f), sz_ = __ballot(valid && R.idir.z < 0.0f);
  if((sx_ != 0ull && sx_ != vm) || (sy_ != 0ull && sy_ != vm) || (sz_ != 0ull && sz_ != vm))
    return false;
  best.slot = BVH_NONE; best.t = PT_INFINITY; best.w = 0xffffffffu; best.flags = 0; best.count = 0;
  best.zeroMaxT = best.zeroMaxT2 = best.zeroMaxT3 = -1.0f;
  best.u = best.v = 0.0f;
  if(S.numTris == 0 || vm == 0ull)
    return true;
  const bool     negx = sx_ != 0ull, negy = sy_ != 0ull, negz = sz_ != 0ull;  // wave-uniform
  const uint32_t octinv = 7u ^ ((negx ? 1u : 0u) | (negy ? 2u : 0u) | (negz ? 4u : 0u));
  const uint32_t ox = negx ? 48u : 0u, oy = negy ? 48u : 0u, oz = negz ? 48u : 0u;

  uint32_t gx = 0u | (1u << (24u + octinv)), gy = 0u, gz = 1u;  // the wave-uniform group: the root
  int      sp = 0;
  for(;;)
  {
    if(!(gx >> 24))
    {
      if(sp == 0)
        break;
      --sp;
      gx = __builtin_amdgcn_readfirstlane(wstack[3 * sp]);
      gy = __builtin_amdgcn_readfirstlane(wstack[3 * sp + 1]);
      gz = __builtin_amdgcn_readfirstlane(wstack[3 * sp + 2]);
    }
    const uint32_t r    = 31u - uint32_t(__builtin_clz(gx));
    const uint32_t slot = (r - 24u) ^ octinv;
    gx &= ~(1u << r);
    if((gz >> (8u + slot)) & 1u)
    {  // a leaf: one or two triangles, both records in flight together
      const uint32_t below = (1u << slot) - 1u;
      const uint32_t first = gy + uint32_t(__builtin_popcount((gz >> 8) & below & 0xffu)) + uint32_t(__builtin_popcount((gz >> 16) & below & 0xffu));
      const bool     two   = ((gz >> (16u + slot)) & 1u) != 0u;
      const uint32_t s0 = first * 48u, s1 = (first + (two ? 1u : 0u)) * 48u;
      TriRec         a, b;
      a.p0w = sload4(S.tris, s0); a.e1n = sload4(S.tris, s0 + 16u); a.e2p = sload4(S.tris, s0 + 32u);
      b.p0w = sload4(S.tris, s1); b.e1n = sload4(S.tris, s1 + 16u); b.e2p = sload4(S.tris, s1 + 32u);
#pragma unroll 1
      for(uint32_t j = 0;; ++j)
      {  // (one copy of the triangle code: see lane_step)
        if(valid)
          lane_triangle<TM_CLOSEST, false>(S, L, first + j, a);
        if(!two || j == 1u)
          break;
        a = b;
      }
      continue;
    }
    const uint32_t child = (gx & CW_CHILD_MASK) + uint32_t(__builtin_popcount(gz & ((1u << slot) - 1u) & 0xffu));
    if(gx >> 24)
    {
      if(sp < PACKET_STACK)
      {
        if((threadIdx.x & 63) == 0)
        {
          wstack[3 * sp] = gx; wstack[3 * sp + 1] = gy; wstack[3 * sp + 2] = gz;
        }
        ++sp;
      }
      else if((threadIdx.x & 63) == 0)
        atomicAdd(&counters->stackOverflow, 1u);
    }
    // ---- the node through scalar loads
    const uint32_t at = child * uint32_t(CW_NODE_BYTES);
    const uint4    h0 = sloadu4(S.wide, at), h1 = sloadu4(S.wide, at + 16u);
    const uint4    nx = sloadu4(S.wide, at + CW_OFF_QLO + ox), fx = sloadu4(S.wide, at + CW_OFF_QHI - ox);
    const uint4    ny = sloadu4(S.wide, at + CW_OFF_QLO + 16u + oy), fy = sloadu4(S.wide, at + CW_OFF_QHI + 16u - oy);
    const uint4    nz = sloadu4(S.wide, at + CW_OFF_QLO + 32u + oz), fz = sloadu4(S.wide, at + CW_OFF_QHI + 32u - oz);
    const float sx = __uint_as_float((h0.w & 0xffu) << 23) * R.idir.x, sy = __uint_as_float(((h0.w >> 8) & 0xffu) << 23) * R.idir.y, sz = __uint_as_float(((h0.w >> 16) & 0xffu) << 23) * R.idir.z;
    const float bx = (__uint_as_float(h0.x) - R.o.x) * R.idir.x, by = (__uint_as_float(h0.y) - R.o.y) * R.idir.y, bz = (__uint_as_float(h0.z) - R.o.z) * R.idir.z;
    const float gmax = float(CW_GRID_MAX);
    const float ex = (fabsf(bx) + gmax * fabsf(sx)) * 8.0e-7f, ey = (fabsf(by) + gmax * fabsf(sy)) * 8.0e-7f, ez = (fabsf(bz) + gmax * fabsf(sz)) * 8.0e-7f;
    const float blx = bx - ex, bhx = bx + ex, bly = by - ey, bhy = by + ey, blz = bz - ez, bhz = bz + ez;
    const float lim  = L.bt;
    uint32_t    hits = 0;
    PK_CHILD(0) PK_CHILD(1) PK_CHILD(2) PK_CHILD(3) PK_CHILD(4) PK_CHILD(5) PK_CHILD(6) PK_CHILD(7)
    const uint32_t kinds = (h0.w >> 24) | ((h1.z & 0xffffu) << 8);
    hits &= (kinds | (kinds >> 8)) & 0xffu;
    gx = (h1.x & CW_CHILD_MASK) | (cw_visit_order(hits, octinv) << 24);
    gy = h1.y;
    gz = kinds;
  }
  best.slot = L.bslot; best.t = L.bt; best.u = L.bu; best.v = L.bv; best.w = L.bw; best.flags = L.flags; best.count = L.cnt;
  best.zeroMaxT = L.zeroMaxT; best.zeroMaxT2 = L.zeroMaxT2; best.zeroMaxT3 = L.zeroMaxT3;
  return true;
}
#undef PK_CHILD
