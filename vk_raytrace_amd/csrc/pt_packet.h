// Packet traversal for coherent rays (the primary rays of an 8x8 pixel block = one wavefront).
//
// All 64 lanes walk ONE traversal of the 8-wide structure (pt_cwbvh.h): node and triangle records are fetched with scalar loads into SGPRs
// (one fetch per wave instead of one per lane, no per-lane address arithmetic, no vector-memory latency in the dependent chain), the node /
// triangle groups and the stack of postponed groups are wave-level (SGPRs + one LDS array), and the visiting order comes from the octant the
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

#define PK_UB(x, j) float(((x) >> (8 * (j))) & 0xffu)
#define PK_CHILD(w, j)                                                                                                                                                      \
  {                                                                                                                                                                         \
    const float tn = fmaxf(fmaxf(__builtin_fmaf(PK_UB(nx[w], j), sx, blx), __builtin_fmaf(PK_UB(ny[w], j), sy, bly)), fmaxf(__builtin_fmaf(PK_UB(nz[w], j), sz, blz), 0.0f)); \
    const float tf = fminf(fminf(__builtin_fmaf(PK_UB(fx[w], j), sx, bhx), __builtin_fmaf(PK_UB(fy[w], j), sy, bhy)), fminf(__builtin_fmaf(PK_UB(fz[w], j), sz, bhz), lim));  \
    if(__ballot(valid && tn <= tf))                                                                                                                                         \
      hits |= ((bits[w] >> (8 * (j))) & 0xffu) << ((index[w] >> (8 * (j))) & 0xffu);                                                                                        \
  }

// `valid`: the lane carries a ray.  wstack: PACKET_STACK uint2 of LDS shared by the wave.  Returns false when the packet
// is not sign-coherent (nothing was traversed; the caller runs the per-lane traversal instead).
PT_DEV bool traverse_packet_closest(const DeviceScene& S, bool valid, f3 o, f3 d, uint2* wstack, RayHit& best, Counters* counters)
{
  TraceLane L;  // the per-lane part of the state: ray, best hit, alpha bookkeeping (groups / stack fields unused: they are wave-level here)
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
  const uint32_t oct4   = octinv * 0x01010101u;

  uint32_t ngx = 0u, ngy = (1u << (24u + octinv)) | 1u, tgx = 0u, tgy = 0u;  // wave-uniform groups: the root
  int      sp  = 0;
  for(;;)
  {
    if(tgy)
    {  // up to two triangles of the group, both records in flight together
      const uint32_t j0 = uint32_t(__builtin_ctz(tgy));
      tgy &= tgy - 1u;
      const bool     two = tgy != 0u;
      const uint32_t j1  = two ? uint32_t(__builtin_ctz(tgy)) : j0;
      tgy &= tgy - 1u;
      const uint32_t s0 = (tgx + j0) * 48u, s1 = (tgx + j1) * 48u;
      TriRec         a, b;
      a.p0w = sload4(S.tris, s0); a.e1n = sload4(S.tris, s0 + 16u); a.e2p = sload4(S.tris, s0 + 32u);
      b.p0w = sload4(S.tris, s1); b.e1n = sload4(S.tris, s1 + 16u); b.e2p = sload4(S.tris, s1 + 32u);
      if(valid)
        lane_triangle<TM_CLOSEST, false>(S, L, tgx + j0, a);
      if(two && valid)
        lane_triangle<TM_CLOSEST, false>(S, L, tgx + j1, b);
      continue;
    }
    if(!(ngy & 0xff000000u))
    {
      if(sp == 0)
        break;
      const uint2 e = wstack[--sp];
      const uint32_t ex = __builtin_amdgcn_readfirstlane(e.x), ey = __builtin_amdgcn_readfirstlane(e.y);
      if(!(ey & 0xff000000u))
      {
        tgx = ex;
        tgy = ey;
        continue;
      }
      ngx = ex;
      ngy = ey;
    }
    const uint32_t r    = 31u - uint32_t(__builtin_clz(ngy));
    const uint32_t slot = (r - 24u) ^ octinv;
    ngy &= ~(1u << r);
    const uint32_t child = ngx + uint32_t(__builtin_popcount(ngy & ((1u << slot) - 1u) & 0xffu));
    if(ngy & 0xff000000u)
    {
      if(sp < PACKET_STACK)
      {
        if((threadIdx.x & 63) == 0)
          wstack[sp] = make_uint2(ngx, ngy);
        ++sp;
      }
      else if((threadIdx.x & 63) == 0)
        atomicAdd(&counters->stackOverflow, 1u);
    }
    // ---- the node through scalar loads
    const uint32_t at = child * uint32_t(CW_NODE_BYTES);
    const uint4    h0 = sloadu4(S.wide, at), h1 = sloadu4(S.wide, at + 16u), q0 = sloadu4(S.wide, at + 32u), q1 = sloadu4(S.wide, at + 48u), q2 = sloadu4(S.wide, at + 64u);
    const float sx = __uint_as_float((h0.w & 0xffu) << 23) * R.idir.x, sy = __uint_as_float(((h0.w >> 8) & 0xffu) << 23) * R.idir.y, sz = __uint_as_float(((h0.w >> 16) & 0xffu) << 23) * R.idir.z;
    const float bx = (__uint_as_float(h0.x) - R.o.x) * R.idir.x, by = (__uint_as_float(h0.y) - R.o.y) * R.idir.y, bz = (__uint_as_float(h0.z) - R.o.z) * R.idir.z;
    const float ex = (fabsf(bx) + 255.0f * fabsf(sx)) * 8.0e-7f, ey = (fabsf(by) + 255.0f * fabsf(sy)) * 8.0e-7f, ez = (fabsf(bz) + 255.0f * fabsf(sz)) * 8.0e-7f;
    const float blx = bx - ex, bhx = bx + ex, bly = by - ey, bhy = by + ey, blz = bz - ez, bhz = bz + ez;
    const uint32_t nx[2] = {negx ? q1.z : q0.x, negx ? q1.w : q0.y}, fx[2] = {negx ? q0.x : q1.z, negx ? q0.y : q1.w};
    const uint32_t ny[2] = {negy ? q2.x : q0.z, negy ? q2.y : q0.w}, fy[2] = {negy ? q0.z : q2.x, negy ? q0.w : q2.y};
    const uint32_t nz[2] = {negz ? q2.z : q1.x, negz ? q2.w : q1.y}, fz[2] = {negz ? q1.x : q2.z, negz ? q1.y : q2.w};
    uint32_t       bits[2], index[2];
#pragma unroll
    for(int w = 0; w < 2; ++w)
    {
      const uint32_t meta  = w ? h1.w : h1.z;
      const uint32_t inner = ((meta & (meta << 1)) & 0x10101010u) >> 4;
      index[w]             = (meta ^ (oct4 & (inner * 0xffu))) & 0x1f1f1f1fu;
      bits[w]              = (meta >> 5) & 0x07070707u;
    }
    const float lim  = L.bt;
    uint32_t    hits = 0;
    PK_CHILD(0, 0) PK_CHILD(0, 1) PK_CHILD(0, 2) PK_CHILD(0, 3) PK_CHILD(1, 0) PK_CHILD(1, 1) PK_CHILD(1, 2) PK_CHILD(1, 3)
    ngx = h1.x & CW_CHILD_MASK;
    ngy = (hits & 0xff000000u) | (h0.w >> 24);
    tgx = h1.y;
    tgy = hits & 0x00ffffffu;
  }
  best.slot = L.bslot; best.t = L.bt; best.u = L.bu; best.v = L.bv; best.w = L.bw; best.flags = L.flags; best.count = L.cnt;
  best.zeroMaxT = L.zeroMaxT; best.zeroMaxT2 = L.zeroMaxT2; best.zeroMaxT3 = L.zeroMaxT3;
  return true;
}
#undef PK_CHILD
#undef PK_UB
