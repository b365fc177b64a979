// BVH traversal and ray/triangle test for gfx950 -- the part of the reference that lives in the
// Vulkan driver + RT cores (shaders/traceray_rq.glsl:110-134 rayQueryProceedEXT loops).
//
// Semantics (DESIGN.md "Trace contract"):
//  * candidates are ordered by the key (t, world triangle index); a query returns the smallest key
//    strictly greater than a previous key inside (0, tmax) -- independent of the BVH shape, so the
//    result is bit-identical for any builder;
//  * back faces are culled unless the instance is double sided; facing is an object-space property,
//    so a mirrored instance flips the sign test (TRI_FLIP);
//  * opaque instances commit directly; non-opaque ones go through the stochastic alpha test, one RNG
//    draw per candidate in key order (T5 closest, T6 shadow: the same walk bounded by the light distance).
//
// Stochastic alpha without re-traversal.  Processing candidates strictly in key order would cost one
// full traversal per rejected candidate (a ray through foliage meets dozens of transparent texels).
// The same result is obtained with at most two traversals:
//   pass A (TM_CLOSEST) finds the nearest CERTAIN hit -- opaque, or opacity >= 1 (the draw
//          r < 1 can never exceed it) -- evaluating the opacity of non-opaque candidates on the fly;
//   pass B (TM_COUNT) only counts the zero-opacity candidates in front of it.  Each of them consumes one
//          draw and is rejected unless that draw is exactly 0.0 (probability 2^-23).
// The caller advances the RNG by the count; if a draw is 0.0, or a candidate with fractional opacity
// (ALPHA_BLEND) lies in front of the certain hit, it falls back to the exact key-ordered loop
// (TM_RAW_*).  All three routes produce identical hits and identical RNG states.
//
// The walk (round 3): 8-wide nodes with quantised child boxes (pt_cwbvh.h).  One node visit = five 16-byte loads per lane, all in flight
// together, and decides 8 children; the hit children are NOT sorted: the builder placed them in octant order, so the hit mask's highest bit
// is the next child to visit.  What a lane carries between steps is
//   a node group      (first inner child, hit bits 31..24 | inner-child mask)   -- the still unvisited hit children of the last node visited
//   a triangle group  (first triangle, 24 hit bits)                              -- the triangles of its hit leaf children
// and a stack of postponed groups: a node visit pushes at most ONE entry (the remainder of the group it came from), so the stack is as deep
// as the tree, not 7x that.  The first STACK_LDS entries of every lane live in LDS laid out [level][lane] (8-byte entries: 64 lanes x 8 B =
// two conflict-free bank rows), deeper entries go to a per-wavefront area in global memory (same layout) -- no scratch memory, no register
// array with a dynamic index.  Overflow beyond STACK_LDS + STACK_SPILL is counted in Counters::stackOverflow and reported by every call that
// hands results to the host.
//
// One step of a lane (lane_step) = up to two triangle tests of its triangle group (their six 16-byte loads in flight together), then -- if
// that emptied the group -- one node visit.  The lock-step walks (traverse<>: exact fallback, ray picker, k_tail) and the refilling trace
// machine of the persistent kernels (pt_render.hip) are the same per-lane code; only who calls lane_step differs.
#pragma once
#include "pt_surface.h"
#include "pt_cwbvh.h"

#define TRACE_BLOCK 64
#ifndef STACK_LDS
#define STACK_LDS 12
#endif
#define STACK_TOTAL 64
#define STACK_SPILL (STACK_TOTAL - STACK_LDS)

enum TraceMode {
  TM_MACHINE = -1,       // the persistent kernels: TM_CLOSEST or TM_COUNT per lane at run time (TraceLane::pass)
  TM_RAW_ALL = 0,        // exact: smallest key > (tPrev,wPrev), every triangle is a candidate, no opacity evaluation
  TM_RAW_NONOPAQUE = 1,  // exact: same, non-opaque triangles only
  TM_CLOSEST = 2,        // pass A: nearest certain hit, flags for uncertain candidates in front of it
  TM_COUNT = 4,          // pass B: number of zero-opacity candidates with key < (tmax, wLimit)
  TM_PICK = 5            // the ray picker's query: TM_RAW_ALL without face culling (every triangle counts)
};
#define TF_SAW_ZERO 1u  // a candidate with opacity <= 0 was seen in front of the (then) best hit
#define TF_SAW_FRAC 2u  // a candidate with 0 < opacity < 1 was seen in front of the (then) best hit

struct RayHit {
  float    t, u, v;
  uint32_t slot;   // TriRec slot (leaf order); BVH_NONE: nothing
  uint32_t w;      // world triangle index | flags << 29
  uint32_t flags;  // TF_*
  uint32_t count;  // TM_COUNT: zero-opacity candidates in range.  TM_CLOSEST: zero-opacity candidates SEEN
  float    zeroMaxT;  // TM_CLOSEST: largest t among the zero-opacity candidates seen (-1: none)
  float    zeroMaxT2, zeroMaxT3;  //       second and third largest
};

// After pass A: if every zero-opacity candidate that was evaluated lies strictly in front of the final certain hit
// (or there is no hit and all of them are inside the ray range), `count` is already the number of draws they
// consume and pass B is unnecessary.  Traversal is near-to-far, so this is the common case; the next most common one
// -- one or two candidates behind the hit -- is settled by tracking the three largest t values.
PT_DEV bool pass_a_settles(uint32_t bslot, float bt, float z1, float z2, float z3, uint32_t& count)
{
  if(bslot == BVH_NONE || z1 < bt)
    return true;
  if(z3 < bt && z1 != bt && z2 != bt)
  {  // every evaluated zero-opacity candidate behind the final hit (seen while the bound was still larger) is among
     // the three largest: subtract them
    count -= (z1 > bt ? 1u : 0u) + (z2 > bt ? 1u : 0u);
    return true;
  }
  return false;  // more than two behind, or a tie in t with the hit (the key order decides): count again in pass B
}
// keeps z1 >= z2 >= z3, the three largest t seen
PT_DEV void note_zero_candidate(float t, float& z1, float& z2, float& z3)
{
  const float a = fminf(z1, t);
  z1            = fmaxf(z1, t);
  const float b = fminf(z2, a);
  z2            = fmaxf(z2, a);
  z3            = fmaxf(z3, b);
}

PT_DEV bool key_less(float ta, uint32_t wa, float tb, uint32_t wb) { return ta < tb || (ta == tb && wa < wb); }

// Moeller-Trumbore on (p0, e1, e2) in the exact operation order of the trace contract (T2, T3).
#ifdef PT_TRI_TEST_OVERRIDE
// host experiments only (tests/cpp/trace_host.cpp -DTH_ROBUST_T2): a candidate replacement of T2 under evaluation takes the place of the contract's
PT_DEV bool tri_test(const TriRec& tr, uint32_t flags, f3 o, f3 d, float& t, float& u, float& v) { return PT_TRI_TEST_OVERRIDE(tr, flags, o, d, t, u, v); }
#else
PT_DEV bool tri_test(const TriRec& tr, uint32_t flags, f3 o, f3 d, float& t, float& u, float& v)
{
  f3    e1  = xyz(tr.e1n), e2 = xyz(tr.e2p), p0 = xyz(tr.p0w);
  f3    pv  = cross3(d, e2);
  float det = dot3(e1, pv);
  if(det == 0.0f)
    return false;
  if(!(flags & TRI_NOCULL))
  {
    bool front = (flags & TRI_FLIP) ? (det < 0.0f) : (det > 0.0f);
    if(!front)
      return false;
  }
  float inv = 1.0f / det;
  f3    tv  = o - p0;
  u         = dot3(tv, pv) * inv;
  if(u < 0.0f || u > 1.0f)
    return false;
  f3 qv = cross3(tv, e1);
  v     = dot3(d, qv) * inv;
  if(v < 0.0f || u + v > 1.0f)
    return false;
  t = dot3(e2, qv) * inv;
  return true;
}
#endif

// ---- traversal stack ------------------------------------------------------------------------------------------------------------------
struct TStack {
  uint2* lds;    // this lane's column of the wavefront's LDS stack: entry k at lds[k * TRACE_BLOCK]
  uint2* spill;  // this lane's column of the wavefront's global spill area (entries STACK_LDS ..): entry k at spill[(k - STACK_LDS) * TRACE_BLOCK]; may be null
};
PT_DEV void stack_push(const TStack& s, int& sp, uint32_t x, uint32_t y, Counters* counters)
{
  if(sp < STACK_LDS)
    s.lds[sp * TRACE_BLOCK] = make_uint2(x, y);
  else if(sp < STACK_TOTAL && s.spill)
    s.spill[(sp - STACK_LDS) * TRACE_BLOCK] = make_uint2(x, y);
  else
  {
    atomicAdd(&counters->stackOverflow, 1u);  // entry dropped (flagged; pt_get_stats / pt_synchronize report it)
    return;
  }
  ++sp;
}
PT_DEV uint2 stack_pop(const TStack& s, int& sp)
{
  --sp;
  return sp < STACK_LDS ? s.lds[sp * TRACE_BLOCK] : s.spill[(sp - STACK_LDS) * TRACE_BLOCK];
}
// the wavefront's spill area of a launch: (wave slot, level, lane)
PT_DEV uint2* spill_column(uint2* spillBase, uint32_t waveSlot) { return spillBase ? spillBase + (size_t(waveSlot) * STACK_SPILL) * TRACE_BLOCK + (threadIdx.x & 63u) : nullptr; }

// ---- per-ray constants of the box tests -------------------------------------------------------------------------------------------------
// A child plane sits at p + q 2^e; its ray parameter is t = q s + b with s = 2^e idir and b = (p - o) idir (one FMA per plane once s and b are
// known for the node).  The box test is not part of the bit-exact contract (results are BVH independent), it only has to be conservative:
// fl(b) is off by at most 2^-23 |b| (difference and product rounded), the FMA by 2^-24 |t|, idir itself by 2^-24 relative, and |t| <= |b| +
// 255 |s|, so biasing b by E = (|b| + 255 |s|) * 8e-7 towards "hit" (b - E for near planes, b + E for far planes) covers all of it twice over
// -- that is 1e-6 of the node's own extent, nothing against the 1/255 of the quantisation.  |d| components below 1e-18 are clamped so that
// idir stays finite (a ray moves < 1 ulp along such an axis over any representable distance).
struct BoxRay {
  f3       o;       // ray origin in the space of the nodes being tested (world; object space inside an instance of the two-level structure)
  f3       idir;
  float    eps;     // extra plane padding (two-level structure: rounding of the ray transform, enter_instance); 0 in world space
  uint32_t oct4;    // octinv * 0x01010101, octinv = 7 ^ (direction sign bits): hit inner children go to bit 24 + (slot ^ octinv), highest bit first
};
PT_DEV BoxRay make_boxray(f3 o, f3 d)
{
  BoxRay      rb;
  const float dx = copysignf(fmaxf(fabsf(d.x), 1.0e-18f), d.x), dy = copysignf(fmaxf(fabsf(d.y), 1.0e-18f), d.y), dz = copysignf(fmaxf(fabsf(d.z), 1.0e-18f), d.z);
  rb.o    = o;
  rb.idir = f3{1.0f / dx, 1.0f / dy, 1.0f / dz};
  rb.eps  = 0.0f;
  const uint32_t oct = (rb.idir.x < 0.0f ? 1u : 0u) | (rb.idir.y < 0.0f ? 2u : 0u) | (rb.idir.z < 0.0f ? 4u : 0u);
  rb.oct4 = (7u ^ oct) * 0x01010101u;
  return rb;
}

// One node visit: the 32-bit hit mask of node `idx` against [0, lim] (bits 31..24 inner children in visiting order, bits 23..0 triangles) and
// the header fields the groups need.  alphaOnly: only children tagged in amask (pass B and the non-opaque walks never need an opaque subtree).
struct NodeHit {
  uint32_t hits, childBase, triBase, imask, amask;
};
#define CW_UB(x, j) float(((x) >> (8 * (j))) & 0xffu)
#define CW_CHILD(w, j)                                                                                                                                                      \
  {                                                                                                                                                                         \
    const float tn = fmaxf(fmaxf(__builtin_fmaf(CW_UB(nx[w], j), sx, blx), __builtin_fmaf(CW_UB(ny[w], j), sy, bly)), fmaxf(__builtin_fmaf(CW_UB(nz[w], j), sz, blz), 0.0f)); \
    const float tf = fminf(fminf(__builtin_fmaf(CW_UB(fx[w], j), sx, bhx), __builtin_fmaf(CW_UB(fy[w], j), sy, bhy)), fminf(__builtin_fmaf(CW_UB(fz[w], j), sz, bhz), lim));  \
    hits |= (tn <= tf) ? (((bits[w] >> (8 * (j))) & 0xffu) << ((index[w] >> (8 * (j))) & 0xffu)) : 0u;                                                                     \
  }
PT_DEV NodeHit cw_test_node(const CwNode* __restrict__ nodes, uint32_t idx, const BoxRay& R, float lim, bool alphaOnly)
{
  const char*    nb = reinterpret_cast<const char*>(nodes);
  const uint32_t at = idx * uint32_t(CW_NODE_BYTES);  // 32-bit byte offsets (the node array is < 4 GB)
  const uint4    h0 = *reinterpret_cast<const uint4*>(nb + at), h1 = *reinterpret_cast<const uint4*>(nb + (at + 16u)), q0 = *reinterpret_cast<const uint4*>(nb + (at + 32u)),
              q1 = *reinterpret_cast<const uint4*>(nb + (at + 48u)), q2 = *reinterpret_cast<const uint4*>(nb + (at + 64u));
  NodeHit        nh;
  nh.childBase = h1.x & CW_CHILD_MASK;
  nh.amask     = h1.x >> 24;
  nh.triBase   = h1.y;
  nh.imask     = h0.w >> 24;
  // per-axis grid step and origin in ray parameters
  const float sx = __uint_as_float((h0.w & 0xffu) << 23) * R.idir.x, sy = __uint_as_float(((h0.w >> 8) & 0xffu) << 23) * R.idir.y, sz = __uint_as_float(((h0.w >> 16) & 0xffu) << 23) * R.idir.z;
  const float bx = (__uint_as_float(h0.x) - R.o.x) * R.idir.x, by = (__uint_as_float(h0.y) - R.o.y) * R.idir.y, bz = (__uint_as_float(h0.z) - R.o.z) * R.idir.z;
  const float ex = __builtin_fmaf(fabsf(R.idir.x), R.eps, (fabsf(bx) + 255.0f * fabsf(sx)) * 8.0e-7f), ey = __builtin_fmaf(fabsf(R.idir.y), R.eps, (fabsf(by) + 255.0f * fabsf(sy)) * 8.0e-7f),
              ez = __builtin_fmaf(fabsf(R.idir.z), R.eps, (fabsf(bz) + 255.0f * fabsf(sz)) * 8.0e-7f);
  const float blx = bx - ex, bhx = bx + ex, bly = by - ey, bhy = by + ey, blz = bz - ez, bhz = bz + ez;
  // near / far planes by the direction sign (q0: qlox, qloy; q1: qloz, qhix; q2: qhiy, qhiz)
  const bool     negx = R.idir.x < 0.0f, negy = R.idir.y < 0.0f, negz = R.idir.z < 0.0f;
  const uint32_t nx[2] = {negx ? q1.z : q0.x, negx ? q1.w : q0.y}, fx[2] = {negx ? q0.x : q1.z, negx ? q0.y : q1.w};
  const uint32_t ny[2] = {negy ? q2.x : q0.z, negy ? q2.y : q0.w}, fy[2] = {negy ? q0.z : q2.x, negy ? q0.w : q2.y};
  const uint32_t nz[2] = {negz ? q2.z : q1.x, negz ? q2.w : q1.y}, fz[2] = {negz ? q1.x : q2.z, negz ? q1.y : q2.w};
  // per child: the bits it contributes and where (inner: 1 bit at 24 + (slot ^ octinv); leaf: its triangles' bits at their offset)
  uint32_t bits[2], index[2];
  const uint32_t am = alphaOnly ? nh.amask : 0xffu;
#pragma unroll
  for(int w = 0; w < 2; ++w)
  {
    const uint32_t meta  = w ? h1.w : h1.z;
    const uint32_t inner = ((meta & (meta << 1)) & 0x10101010u) >> 4;                         // 0x01 in the bytes of inner children
    index[w]             = (meta ^ (R.oct4 & (inner * 0xffu))) & 0x1f1f1f1fu;
    const uint32_t keep  = (((am >> (4 * w)) & 0xfu) * 0x00204081u) & 0x01010101u;             // bit k of the nibble -> byte k
    bits[w]              = ((meta >> 5) & 0x07070707u) & (keep * 0xffu);
  }
  uint32_t hits = 0;
  CW_CHILD(0, 0) CW_CHILD(0, 1) CW_CHILD(0, 2) CW_CHILD(0, 3) CW_CHILD(1, 0) CW_CHILD(1, 1) CW_CHILD(1, 2) CW_CHILD(1, 3)
  nh.hits = hits;
  return nh;
}
#undef CW_CHILD
#undef CW_UB

// ---- two-level walk (TLAS over instances, one object-space BLAS per prim-mesh; reference: src/accelstruct.cpp:110-162) -----------
// A lane is either at TLAS level (InstCtx::inst == BVH_NONE: world-space ray constants, groups index DeviceScene::tlas / tlasLeaves) or inside
// one instance (object-space ray constants, groups index DeviceScene::wide / tris).  The ray parameter t is the same in both spaces
// (the direction is transformed, not renormalised), so the current bound prunes in either.  Entering an instance postpones what is left of the
// TLAS-level groups on the stack; the instance is left when the stack has shrunk back to the depth it had then.
struct InstCtx {
  uint32_t inst;    // BVH_NONE: at TLAS level
  int      spBase;  // stack depth at entry
  uint32_t wflags;  // world index of the instance's first triangle | TRI_* flags << 29
};
#define PT_TWO_GUARD (1u << 20)  // loop-iteration bound of the two-level walks (a corrupt structure must not hang the GPU; reported as a stack overflow)

// Object-space ray constants of a ray entering the instance of TLAS leaf `tl`.  Transforming the ray rounds (o' and d' carry an absolute
// error of a few 2^-24 x |worldToObject| x (|o| + |hit point|)); instead of tracking it per plane the BLAS boxes are grown by
// eps = padC1 * max|o| + padC0 (host-computed bound with a 16x margin, pt_capi.hip: two_level_pad): (plane -+ eps) idir = plane idir -+ eps |idir|.
PT_DEV BoxRay enter_instance(const DeviceScene& S, const TlasLeaf& tl, f3 o, f3 d)
{
  const Affine W  = S.instances[tl.inst].worldToObject;
  BoxRay       rb = make_boxray(xform_point(W, o), xform_dir(W, d));
  rb.eps          = tl.padC1 * fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))) + tl.padC0;
  return rb;
}
// Trace contract T1 at the leaf: the instance matrix applied to the three object-space vertices in the operation order of k_world_tris
// (pt_accel.hip), edges taken in world space -- the record the flat structure stores, rebuilt on the fly.
PT_DEV TriRec world_tri(const DeviceScene& S, const InstCtx& ic, const TriRec& obj)
{
  const Affine   M  = S.instances[ic.inst].objectToWorld;
  const f3       p0 = xform_point(M, xyz(obj.p0w)), p1 = xform_point(M, xyz(obj.e1n)), p2 = xform_point(M, xyz(obj.e2p));
  const f3       e1 = p1 - p0, e2 = p2 - p0;
  const uint32_t prim = __float_as_uint(obj.p0w.w);
  TriRec         r;
  r.p0w = make_float4(p0.x, p0.y, p0.z, __uint_as_float(((ic.wflags & TRI_INDEX_MASK) + prim) | (ic.wflags & ~TRI_INDEX_MASK)));
  r.e1n = make_float4(e1.x, e1.y, e1.z, __uint_as_float(ic.inst));
  r.e2p = make_float4(e2.x, e2.y, e2.z, __uint_as_float(prim));
  return r;
}
// world triangle index -> instance: the LAST instance whose triBase is <= w (empty instances share a base and sort before the owner)
PT_DEV uint32_t instance_of_world_tri(const DeviceScene& S, uint32_t w)
{
  uint32_t lo = 0, hi = S.numInstances - 1;
  while(lo < hi)
  {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if(S.instTriBase[mid] <= w)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

// ---- the per-lane walk ----------------------------------------------------------------------------------------------------------------------
struct TraceLane {
  f3       o, d;           // the ray in world space (the triangle test of the contract runs there)
  BoxRay   R;              // box-test constants of the space the lane is in
  float    tmax;           // exclusive upper bound on t (TM_COUNT: the limit key's t, inclusive for ties)
  float    bt, bu, bv;     // best hit so far (pass A: best CERTAIN hit)
  uint32_t bslot, bw;
  uint32_t ngx, ngy;       // node group: first inner child | hit bits 31..24, inner-child mask 7..0 (no hit bits: nothing to visit)
  uint32_t tgx, tgy;       // triangle group: first triangle (two-level, TLAS level: first TlasLeaf), 24 hit bits
  int      sp;
  uint32_t flags, cnt, wLimit;
  float    zeroMaxT, zeroMaxT2, zeroMaxT3;  // pass A: the three largest t among the zero-opacity candidates seen
  float    tPrev;          // TM_RAW_* / TM_PICK: exclusive lower key
  uint32_t wPrev;
  int      pass;           // TM_MACHINE: 0 = pass A (nearest certain hit), 1 = pass B (count zero-opacity candidates in front of it)
  bool     done;
  bool     sawAlpha;       // a visited node had children tagged non-opaque (shadow rays: without one, the first certain hit is final -- see lane_step)
  bool     anyEnds;        // shadow ray: the first certain hit may end the walk when no non-opaque geometry was met (nothing can consume a draw)
  InstCtx  ic;             // two-level instantiations only: the instance the lane is inside of
  uint32_t steps;          //   and the loop-iteration guard
#ifdef PT_STATS
  uint32_t nNodes, nTris;
#endif
};

PT_DEV void lane_root(TraceLane& L)
{
  // a group whose only hit child is "inner child 0 of base 0" = node 0, at the bit the ray's octant gives slot 0
  L.ngx = 0u;
  L.ngy = (1u << (24u + (L.R.oct4 & 7u))) | 1u;
  L.tgx = 0u;
  L.tgy = 0u;
  L.sp  = 0;
}
PT_DEV void lane_begin(TraceLane& L, f3 o, f3 d, float tmax, bool emptyScene)
{
  L.o = o; L.d = d;
  L.R = make_boxray(o, d);
  L.tmax = tmax; L.bt = tmax; L.bu = 0.f; L.bv = 0.f; L.bslot = BVH_NONE; L.bw = 0xffffffffu;
  lane_root(L);
  L.flags = 0; L.cnt = 0; L.wLimit = 0; L.pass = 0; L.done = emptyScene; L.zeroMaxT = -1.0f; L.zeroMaxT2 = -1.0f; L.zeroMaxT3 = -1.0f;
  L.tPrev = 0.0f; L.wPrev = 0xffffffffu; L.sawAlpha = false; L.anyEnds = false;
  L.ic = InstCtx{BVH_NONE, 0, 0u}; L.steps = 0;
#ifdef PT_STATS
  L.nNodes = 0; L.nTris = 0;
#endif
}
// pass B over the candidates with key < (best hit | ray end)
template <bool TWO = false>
PT_DEV void lane_begin_count(TraceLane& L)
{
  const bool found = L.bslot != BVH_NONE;
  L.wLimit = found ? (L.bw & TRI_INDEX_MASK) : 0u;
  L.tmax   = found ? L.bt : L.tmax;
  if(TWO)
  {  // pass A may have ended inside an instance
    L.ic.inst = BVH_NONE;
    L.R       = make_boxray(L.o, L.d);
  }
  lane_root(L);
  L.flags = 0; L.cnt = 0; L.pass = 1; L.done = false;
}

// One triangle of the lane's triangle group under the candidate rules of `mode`.
template <int MODE, bool TWO>
PT_DEV void lane_triangle(const DeviceScene& S, TraceLane& L, uint32_t slot, TriRec tr)
{
  const int mode = MODE == TM_MACHINE ? (L.pass ? TM_COUNT : TM_CLOSEST) : MODE;
  if(TWO)
    tr = world_tri(S, L.ic, tr);
  const uint32_t wbits = __float_as_uint(tr.p0w.w);
  const uint32_t flags = wbits >> 29;
  const bool     opq   = (flags & TRI_OPAQUE) != 0;
  if((mode == TM_RAW_NONOPAQUE || mode == TM_COUNT) && opq)
    return;
#ifdef PT_STATS
  ++L.nTris;
#endif
  float t, u, v;
  // (TM_COUNT's upper key (tmax, wLimit) includes candidates that tie with the hit in t)
  if(!tri_test(tr, mode == TM_PICK ? (flags | TRI_NOCULL) : flags, L.o, L.d, t, u, v) || !(mode == TM_COUNT ? t <= L.tmax : t < L.tmax))
    return;
  const uint32_t w = wbits & TRI_INDEX_MASK;
  if(mode == TM_RAW_ALL || mode == TM_RAW_NONOPAQUE || mode == TM_PICK)
  {
    if(key_less(L.tPrev, L.wPrev, t, w) && (L.bslot == BVH_NONE || key_less(t, w, L.bt, L.bw & TRI_INDEX_MASK)))
    {
      L.bt = t; L.bu = u; L.bv = v; L.bslot = slot; L.bw = wbits;
    }
  }
  else if(mode == TM_COUNT)
  {
    if(t > 0.0f && key_less(t, w, L.tmax, L.wLimit))
    {
      const float op = opacity_class(S, S.alphaRecs[slot], u, v);
      if(op <= 0.0f)
        L.cnt++;
      else if(op < 1.0f)
        L.flags |= TF_SAW_FRAC;
      // (op >= 1 cannot occur in front of the nearest certain hit)
    }
  }
  else  // TM_CLOSEST
  {
    if(t > 0.0f && (L.bslot == BVH_NONE || key_less(t, w, L.bt, L.bw & TRI_INDEX_MASK)))
    {
      bool certain = opq;
      if(!opq)
      {
        const float op = opacity_class(S, S.alphaRecs[slot], u, v);
        certain        = op >= 1.0f;
        if(!certain)
        {
          L.flags |= (op <= 0.0f) ? TF_SAW_ZERO : TF_SAW_FRAC;
          if(op <= 0.0f)
          {
            L.cnt++;
            note_zero_candidate(t, L.zeroMaxT, L.zeroMaxT2, L.zeroMaxT3);
          }
        }
      }
      if(certain)
      {
        L.bt = t; L.bu = u; L.bv = v; L.bslot = slot; L.bw = wbits;
        // Shadow rays (T6: the nearest certain hit ends the ray, zero-opacity candidates in front of it consume draws): as long as no visited
        // node had a child tagged non-opaque, no candidate anywhere along the ray can consume a draw and ANY certain hit gives the same
        // verdict and the same RNG state as the nearest one -- the reference's own gl_RayFlagsTerminateOnFirstHitEXT (traceray_rq.glsl:157).
        // Sound because unvisited nodes are descendants of visited ones: a subtree without the tag holds opaque triangles only.
        if(L.anyEnds && !L.sawAlpha)
          L.done = true;
      }
    }
  }
}

// One step of the walk: up to two triangles of the lane's triangle group, then -- if the group is empty -- one node visit (of the next hit
// child of the lane's node group, or of a group popped from the stack).  Sets L.done when nothing is left.
template <int MODE, bool TWO>
PT_DEV void lane_step(const DeviceScene& S, TraceLane& L, const TStack& st, Counters* counters)
{
  const int  mode      = MODE == TM_MACHINE ? (L.pass ? TM_COUNT : TM_CLOSEST) : MODE;
  const bool alphaOnly = mode == TM_COUNT || mode == TM_RAW_NONOPAQUE;
  if(TWO && ++L.steps > PT_TWO_GUARD)
  {
    atomicAdd(&counters->stackOverflow, 1u);
    L.done = true;
    return;
  }
  // ---- triangles
  if(L.tgy)
  {
    if(TWO && L.ic.inst == BVH_NONE)
    {  // TLAS level: the group's members are instances.  Enter the first; what is left at this level waits on the stack.
      const uint32_t j = uint32_t(__ffs(int(L.tgy))) - 1u;
      const TlasLeaf tl = S.tlasLeaves[L.tgx + j];
      L.tgy &= L.tgy - 1u;
      if(L.tgy)
        stack_push(st, L.sp, L.tgx, L.tgy, counters);
      if(L.ngy & 0xff000000u)
        stack_push(st, L.sp, L.ngx, L.ngy, counters);
      L.ic  = InstCtx{tl.inst, L.sp, tl.wflags};
      L.R   = enter_instance(S, tl, L.o, L.d);
      L.ngx = tl.nodeBase;                                  // "inner child 0 of base nodeBase" = the BLAS root
      L.ngy = (1u << (24u + (L.R.oct4 & 7u))) | 1u;
      L.tgy = 0u;
    }
    else
    {
      const uint32_t j0  = uint32_t(__ffs(int(L.tgy))) - 1u;
      L.tgy &= L.tgy - 1u;
      const bool     two = L.tgy != 0u;
      const uint32_t j1  = two ? uint32_t(__ffs(int(L.tgy))) - 1u : j0;
      L.tgy &= L.tgy - 1u;  // (0 stays 0)
      const TriRec a = S.tris[L.tgx + j0], b = S.tris[L.tgx + j1];  // six 16-byte loads in flight together
      lane_triangle<MODE, TWO>(S, L, L.tgx + j0, a);
      if(two && !L.done)
        lane_triangle<MODE, TWO>(S, L, L.tgx + j1, b);
      if(L.done)
        return;
    }
  }
  if(L.tgy)
    return;
  // ---- node
  if(!(L.ngy & 0xff000000u))
  {
    if(TWO && L.ic.inst != BVH_NONE && L.sp == L.ic.spBase)
    {  // the instance's subtree is exhausted: back to TLAS level.  The world-space constants are recomputed rather than kept
       // (8 VGPRs for the lifetime of the lane against ~40 instructions per instance visit)
      L.ic.inst = BVH_NONE;
      L.R       = make_boxray(L.o, L.d);
    }
    if(L.sp == 0)
    {
      L.done = true;
      return;
    }
    const uint2 e = stack_pop(st, L.sp);
    if(!(e.y & 0xff000000u))
    {  // a postponed triangle group (two-level: the rest of an instance group)
      L.tgx = e.x;
      L.tgy = e.y;
      return;
    }
    L.ngx = e.x;
    L.ngy = e.y;
  }
  const uint32_t r    = 31u - uint32_t(__clz(int(L.ngy)));
  const uint32_t slot = (r - 24u) ^ (L.R.oct4 & 7u);
  L.ngy &= ~(1u << r);
  const uint32_t child = L.ngx + uint32_t(__popc(L.ngy & ((1u << slot) - 1u) & 0xffu));
  if(L.ngy & 0xff000000u)
    stack_push(st, L.sp, L.ngx, L.ngy, counters);
#ifdef PT_STATS
  ++L.nNodes;
#endif
  const float   lim = mode == TM_COUNT ? L.tmax : L.bt;
  const NodeHit nh  = cw_test_node((TWO && L.ic.inst == BVH_NONE) ? S.tlas : S.wide, child, L.R, lim, alphaOnly);
  L.ngx = nh.childBase;
  L.ngy = (nh.hits & 0xff000000u) | nh.imask;
  L.tgx = nh.triBase;
  L.tgy = nh.hits & 0x00ffffffu;
  L.sawAlpha = L.sawAlpha || nh.amask != 0u;
}

// The lock-step form: one ray per lane until it is done.
// tPrev/wPrev: exclusive lower key (TM_RAW_*); wLimit: with tmax the exclusive upper key (TM_COUNT).
// TWO: the two-level structure; RayHit::slot is then the global BLAS leaf slot, RayHit::w the world index | flags as always.
template <int MODE, bool TWO = false>
PT_DEV void traverse(const DeviceScene& S, f3 o, f3 d, float tmax, float tPrev, uint32_t wPrev, uint32_t wLimit, const TStack& st, RayHit& best, Counters* counters)
{
  TraceLane L;
  lane_begin(L, o, d, tmax, S.numTris == 0);
  L.tPrev  = tPrev;
  L.wPrev  = wPrev;
  L.wLimit = wLimit;
  L.pass   = MODE == TM_COUNT ? 1 : 0;
  while(!L.done)
    lane_step<MODE, TWO>(S, L, st, counters);
  best.slot = L.bslot; best.t = L.bt; best.u = L.bu; best.v = L.bv; best.w = L.bw; best.flags = L.flags; best.count = L.cnt;
  best.zeroMaxT = L.zeroMaxT; best.zeroMaxT2 = L.zeroMaxT2; best.zeroMaxT3 = L.zeroMaxT3;
#ifdef PT_STATS
  atomicAdd(&counters->nodesVisited, (unsigned long long)L.nNodes);
  atomicAdd(&counters->trisTested, (unsigned long long)L.nTris);
#endif
}

// Advances `seed` by the `n` draws the rejected zero-opacity candidates consume.  Returns false if one of
// the draws is exactly 0.0 (that candidate would have passed rand > 0): the caller must take the exact path.
PT_DEV bool consume_rejected_draws(uint32_t& seed, uint32_t n)
{
  for(uint32_t i = 0; i < n; ++i)
    if(rng_next(seed) == 0.0f)
      return false;
  return true;
}
