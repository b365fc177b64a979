// Sub-group trace machine: FOUR lanes per ray ("quad") on the 4-wide nodes -- the round-6 experiment the round-5 review asked for.
//
// The per-lane machine of pt_machine.h issues every instruction of a node visit for the lanes that happen to be at an inner node (hardware
// counter SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU: 21.4 of 64 lanes in k_closest_p, 23.8 in k_shadow_p, profiles/r06a_binders.json) and makes five
// scattered 16-byte requests per lane and visit.  Here a wavefront carries 16 rays; the four lanes of a quad hold the SAME ray state (every
// per-ray computation is executed identically by the four, so the state never has to be communicated) and differ only in the child they test:
//   * the node is stored child by child (QuadNode, pt_device.h): lane k reads the 32 bytes of child k -- two 16-byte requests per lane, the quad's
//     128 bytes contiguous;
//   * one box test per lane (the arithmetic of wide_node_step: same fp32 planes, same fused slab test, same (1 -+ 4e-7) slack), six FMAs;
//   * the far-to-near order comes from DPP quad_perm moves: a lane's rank is the number of siblings with a smaller key (key = entry distance with
//     the lane index in its two low bits, so keys are distinct; misses sort last);
//   * every hit child but the nearest is written to the ray's stack in one ds_write (position sp + hits - 1 - rank), the nearest is OR-reduced
//     over the quad and becomes the next node; the stack is [level][quad] in LDS, all four lanes read the same word on a pop.
// The triangle step is lane_leaf_core of pt_machine.h, executed redundantly by the four lanes (same semantics by construction: trace contract T2-T6).
//
// The exchange is written against three primitives (rotate by 1 / 2 / 3 inside the quad, OR over the quad) so that tests/cpp/trace_host.cpp runs
// the same phase functions with four emulated lanes and holds the walk to brute force ray by ray (tests/test_trace_host.py).
#pragma once
#include "pt_machine.h"

#define QUAD_RAYS 16     // rays per wavefront
#define QUAD_STACK 48    // stack levels per ray, all in LDS (3 KB per wavefront); deepest level used on the C3 / C5 stand-ins: 22
#define QUAD_KEY_MISS 0xfffffffcu

struct QuadChild {
  uint32_t key;    // entry distance bits with the lane index in bits 0-1; >= QUAD_KEY_MISS: the ray misses this child
  uint32_t child;  // the child reference (as WideNode::child)
};

// phase 1, per lane: the box test of child k of node `cur`
PT_DEV QuadChild quad_box(const QuadNode* __restrict__ nodes, uint32_t cur, uint32_t k, const RayBox& rb, float lim, bool alphaOnly)
{
  const char*    nb = reinterpret_cast<const char*>(nodes);
  const uint32_t at = ((cur & BVH_SLOT_MASK) << 7) | (k << 5);  // sizeof(QuadNode) == 128, 32 bytes per child
  const float4   A  = *reinterpret_cast<const float4*>(nb + at);        // lo.x lo.y lo.z hi.x
  const float4   B  = *reinterpret_cast<const float4*>(nb + (at + 16u));  // hi.y hi.z child -
  const bool     ngx = rb.nearOff[0] != 0, ngy = rb.nearOff[1] != 0, ngz = rb.nearOff[2] != 0;  // negative direction: the upper plane is the near one
  const float    nx = __builtin_fmaf(ngx ? A.w : A.x, rb.idir.x, rb.nlo.x), fx = __builtin_fmaf(ngx ? A.x : A.w, rb.idir.x, rb.nhi.x);
  const float    ny = __builtin_fmaf(ngy ? B.x : A.y, rb.idir.y, rb.nlo.y), fy = __builtin_fmaf(ngy ? A.y : B.x, rb.idir.y, rb.nhi.y);
  const float    nz = __builtin_fmaf(ngz ? B.y : A.z, rb.idir.z, rb.nlo.z), fz = __builtin_fmaf(ngz ? A.z : B.y, rb.idir.z, rb.nhi.z);
  const float    nr = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f)) * 0.9999996f;
  const float    fr = fminf(fminf(fx, fy), fminf(fz, lim)) * 1.0000004f;
  const uint32_t c  = __float_as_uint(B.z);
  const bool     h  = (nr <= fr) && (c != BVH_NONE) && (!alphaOnly || (c & BVH_ALPHA));
  QuadChild      q;
  q.key   = h ? ((__float_as_uint(nr) & ~3u) | k) : (QUAD_KEY_MISS | k);  // nr >= 0: its bit pattern orders like the value
  q.child = c;
  return q;
}
// phase 2, per lane, from the own key and the three siblings' keys: rank = siblings that are nearer, hits = children the ray enters
PT_DEV void quad_rank(uint32_t key, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t& rank, uint32_t& hits)
{
  rank = (k1 < key ? 1u : 0u) + (k2 < key ? 1u : 0u) + (k3 < key ? 1u : 0u);
  hits = (key < QUAD_KEY_MISS ? 1u : 0u) + (k1 < QUAD_KEY_MISS ? 1u : 0u) + (k2 < QUAD_KEY_MISS ? 1u : 0u) + (k3 < QUAD_KEY_MISS ? 1u : 0u);
}
// what a lane contributes to the OR over the quad that yields the next node
PT_DEV uint32_t quad_nearest_part(const QuadChild& q, uint32_t rank) { return (q.key < QUAD_KEY_MISS && rank == 0u) ? q.child : 0u; }

PT_DEV void quad_pop(TraceLane& L, const uint32_t* qstack)
{
  if(L.sp == 0)
  {
    L.done = true;
    return;
  }
  --L.sp;
  L.cur = qstack[L.sp * QUAD_RAYS];
}
// phase 3, per lane: the stack write of this lane's child, the ray's next node.  `nearest`: the OR of quad_nearest_part over the quad.
PT_DEV void quad_apply(TraceLane& L, const QuadChild& q, uint32_t rank, uint32_t hits, uint32_t nearest, uint32_t* qstack, Counters* counters)
{
  if(hits == 0u)
  {
    quad_pop(L, qstack);
    return;
  }
  if(q.key < QUAD_KEY_MISS && rank != 0u)
  {  // far to near: the farthest at sp, the second nearest on top
    const uint32_t at = uint32_t(L.sp) + (hits - 1u - rank);
    if(at < QUAD_STACK)
      qstack[at * QUAD_RAYS] = q.child;
    else
      atomicAdd(&counters->stackOverflow, 1u);  // child dropped (flagged; pt_get_stats reports it)
  }
  const uint32_t top = uint32_t(L.sp) + (hits - 1u);
  L.sp  = int(top < QUAD_STACK ? top : QUAD_STACK);
  L.cur = nearest;
}

#if defined(__HIPCC__)
#if !defined(__HIP_DEVICE_COMPILE__)
// (the host pass of hipcc only needs the names: __global__ bodies are parsed, never run)
PT_DEV uint32_t quad_rot1(uint32_t v) { return v; }
PT_DEV uint32_t quad_rot2(uint32_t v) { return v; }
PT_DEV uint32_t quad_rot3(uint32_t v) { return v; }
PT_DEV uint32_t quad_or(uint32_t v) { return v; }
PT_DEV uint32_t quad_first(uint32_t v) { return v; }
#else
// quad_perm rotations: lane k reads lane (k + r) & 3 of its quad
PT_DEV uint32_t quad_rot1(uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x39, 0xf, 0xf, false)); }  // [1,2,3,0]
PT_DEV uint32_t quad_rot2(uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x4e, 0xf, 0xf, false)); }  // [2,3,0,1]
PT_DEV uint32_t quad_rot3(uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x93, 0xf, 0xf, false)); }  // [3,0,1,2]
PT_DEV uint32_t quad_or(uint32_t v)
{
  v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0xb1, 0xf, 0xf, false));  // [1,0,3,2]
  v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x4e, 0xf, 0xf, false));  // [2,3,0,1]
  return v;
}
PT_DEV uint32_t quad_first(uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x00, 0xf, 0xf, false)); }  // [0,0,0,0]: the quad leader's value
#endif

// One inner-node visit of the quad's ray.  SHADOW as lane_inner.  Every lane of the wavefront that calls this must do so with its whole quad
// (the DPP moves read the siblings' registers): the callers branch on per-ray state only, which the four lanes share.
template <bool SHADOW>
PT_DEV void quad_inner(const DeviceScene& S, TraceLane& L, uint32_t k, uint32_t* qstack, Counters* counters)
{
  const float     lim = (SHADOW || L.pass == 1) ? L.tmax : L.bt;
  const QuadChild q   = quad_box(S.qnodes, L.cur, k, L.rbox, lim, L.pass == 1);
  uint32_t        rank, hits;
  quad_rank(q.key, quad_rot1(q.key), quad_rot2(q.key), quad_rot3(q.key), rank, hits);
  const uint32_t nearest = quad_or(quad_nearest_part(q, rank));
  quad_apply(L, q, rank, hits, nearest, qstack, counters);
}
// One leaf visit: the triangle step of the per-lane machine, executed identically by the four lanes
template <bool SHADOW>
PT_DEV void quad_leaf(const DeviceScene& S, TraceLane& L, const uint32_t* qstack)
{
  const uint32_t slot = L.cur & BVH_SLOT_MASK;
  const TriRec   tr   = S.tris[slot];
  AlphaRec       ar;
  if(L.cur & BVH_ALPHA)
    ar = S.alphaRecs[slot];
  lane_leaf_core<SHADOW, false>(S, L, slot, tr, ar, [&]() { quad_pop(L, qstack); });
}
#endif
