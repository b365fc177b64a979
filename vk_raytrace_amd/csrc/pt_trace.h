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
// The walk (round 3): 8-wide nodes with child boxes on an 11-bit grid (pt_cwbvh.h).  One node visit = eight 16-byte loads per lane, all in flight
// together, one v_fma_mix_f32 per plane, and decides 8 children; the hit children are NOT sorted: the builder placed them in octant order, so
// after an XOR-permutation of the 8 hit bits with the ray's direction signs the highest bit is the next child to visit -- leaf or inner.
// What a lane carries between steps is ONE group of three words
//   first inner child | hit bits 31..24,   first triangle,   inner mask | leaf mask << 8 | two-triangle-leaf mask << 16
// = the still unvisited hit children of the last node visited, and a stack of postponed groups: a node visit pushes at most ONE entry (the
// remainder of the group it came from), so the stack is as deep as the tree, not 7x that.  The first STACK_LDS entries of every lane live in
// LDS laid out [word][level][lane] (conflict free), deeper entries go to a per-wavefront area in global memory (same layout) -- no scratch
// memory, no register array with a dynamic index.  Overflow beyond STACK_TOTAL is counted in Counters::stackOverflow and reported by every
// call that hands results to the host.
//
// One step of a lane (lane_step) = if the next child of its group is a leaf, that leaf's one or two triangles (both records in flight
// together); then, if the next child is an inner node, that node's visit.  The lock-step walks (traverse<>: exact fallback, ray picker,
// k_tail) and the refilling trace machine of the persistent kernels (pt_render.hip) are the same per-lane code; only who calls lane_step differs.
#pragma once
#include "pt_surface.h"
#include "pt_cwbvh.h"

#define TRACE_BLOCK 64
#ifndef STACK_LDS
#define STACK_LDS 8
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
// Entries are three words, stored as three planes of [level][lane] words: every access is 64 consecutive dwords (one conflict-free bank row).
#define STACK_WORDS 3
struct TStack {
  uint32_t* lds;    // this lane's column of the wavefront's LDS stack: word w of entry k at lds[(w * STACK_LDS + k) * TRACE_BLOCK]
  uint32_t* spill;  // this lane's column of the wavefront's global spill area: word w of entry k >= STACK_LDS at spill[(w * STACK_SPILL + k - STACK_LDS) * TRACE_BLOCK]; may be null
};
PT_DEV void stack_push(const TStack& s, int& sp, uint32_t x, uint32_t y, uint32_t z, Counters* counters)
{
  if(sp < STACK_LDS)
  {
    s.lds[(0 * STACK_LDS + sp) * TRACE_BLOCK] = x;
    s.lds[(1 * STACK_LDS + sp) * TRACE_BLOCK] = y;
    s.lds[(2 * STACK_LDS + sp) * TRACE_BLOCK] = z;
  }
  else if(sp < STACK_TOTAL && s.spill)
  {
    const int k = sp - STACK_LDS;
    s.spill[(0 * STACK_SPILL + k) * TRACE_BLOCK] = x;
    s.spill[(1 * STACK_SPILL + k) * TRACE_BLOCK] = y;
    s.spill[(2 * STACK_SPILL + k) * TRACE_BLOCK] = z;
  }
  else
  {
    atomicAdd(&counters->stackOverflow, 1u);  // entry dropped (flagged; pt_get_stats / pt_synchronize report it)
    return;
  }
  ++sp;
}
PT_DEV void stack_pop(const TStack& s, int& sp, uint32_t& x, uint32_t& y, uint32_t& z)
{
  --sp;
  if(sp < STACK_LDS)
  {
    x = s.lds[(0 * STACK_LDS + sp) * TRACE_BLOCK];
    y = s.lds[(1 * STACK_LDS + sp) * TRACE_BLOCK];
    z = s.lds[(2 * STACK_LDS + sp) * TRACE_BLOCK];
  }
  else
  {
    const int k = sp - STACK_LDS;
    x = s.spill[(0 * STACK_SPILL + k) * TRACE_BLOCK];
    y = s.spill[(1 * STACK_SPILL + k) * TRACE_BLOCK];
    z = s.spill[(2 * STACK_SPILL + k) * TRACE_BLOCK];
  }
}
#define STACK_LDS_WORDS (STACK_WORDS * STACK_LDS * TRACE_BLOCK)       // uint32_t per wavefront in LDS
#define STACK_SPILL_WORDS (STACK_WORDS * STACK_SPILL * TRACE_BLOCK)   // uint32_t per wavefront in the global spill area
// the wavefront's spill area of a launch: (wave slot, word, level, lane)
PT_DEV uint32_t* spill_column(uint32_t* spillBase, uint32_t waveSlot) { return spillBase ? spillBase + size_t(waveSlot) * STACK_SPILL_WORDS + (threadIdx.x & 63u) : nullptr; }

// ---- per-ray constants of the box tests -------------------------------------------------------------------------------------------------
// A child plane sits at p + q 2^e; its ray parameter is t = q s + b with s = 2^e idir and b = (p - o) idir: one v_fma_mix_f32 per plane once s
// and b are known for the node (q is read as fp16 out of a register half, s and b are fp32, the result is fp32 with one rounding).  The box test
// is not part of the bit-exact contract (results are BVH independent), it only has to be conservative: fl(b) is off by at most 2^-23 |b|
// (difference and product rounded), the FMA by 2^-24 |t|, idir itself by 2^-24 relative, and |t| <= |b| + 2047 |s|, so biasing b by
// E = (|b| + 2047 |s|) * 8e-7 towards "hit" (b - E for near planes, b + E for far planes) covers all of it twice over -- that is 1e-6 of the
// node's own extent, nothing against the 1/2047 of the quantisation.  The sign of idir says which plane of a slab is the near one: near and
// far planes are fetched from sign-dependent offsets inside the node (no per-axis min / max, no select).  |d| components below 1e-18 are
// clamped so that idir stays finite (a ray moves < 1 ulp along such an axis over any representable distance).
struct BoxRay {
  f3       o;       // ray origin in the space of the nodes being tested (world; object space inside an instance of the two-level structure)
  f3       idir;
  float    eps;     // extra plane padding (two-level structure: rounding of the ray transform, enter_instance); 0 in world space
  uint32_t octinv;  // 7 ^ (direction sign bits): hit child `slot` goes to bit 24 + (slot ^ octinv) of the group, highest bit first
};
PT_DEV BoxRay make_boxray(f3 o, f3 d)
{
  BoxRay      rb;
  const float dx = copysignf(fmaxf(fabsf(d.x), 1.0e-18f), d.x), dy = copysignf(fmaxf(fabsf(d.y), 1.0e-18f), d.y), dz = copysignf(fmaxf(fabsf(d.z), 1.0e-18f), d.z);
  rb.o    = o;
  rb.idir = f3{1.0f / dx, 1.0f / dy, 1.0f / dz};
  rb.eps  = 0.0f;
  rb.octinv = 7u ^ ((rb.idir.x < 0.0f ? 1u : 0u) | (rb.idir.y < 0.0f ? 2u : 0u) | (rb.idir.z < 0.0f ? 4u : 0u));
  return rb;
}

// fp16 grid coordinate of child `k` (0..7) out of the plane set's four words
#if defined(__HIP_DEVICE_COMPILE__)
typedef _Float16 pt_h2v __attribute__((ext_vector_type(2)));
PT_DEV float cw_plane(uint32_t word, int hi)
{
  const pt_h2v h = __builtin_bit_cast(pt_h2v, word);
  return float(hi ? h.y : h.x);  // folds into v_fma_mix_f32's op_sel
}
#else
PT_DEV float cw_plane(uint32_t word, int hi) { return cw_float_of_half(hi ? (word >> 16) : word); }
#endif
PT_DEV uint32_t cw_word(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// One node visit: the hit children of node `idx` against [0, lim] in SLOT order (bit k = slot k; lane_step permutes them into visiting order)
// and the header fields the group needs.  alphaOnly: only children tagged in amask (pass B and the non-opaque walks never need an opaque subtree).
struct NodeHit {
  uint32_t hits, childBase, triBase, kinds, alphaHits;  // kinds: inner mask | leaf mask << 8 | two-triangle-leaf mask << 16
  uint32_t nearest;                                     // slot of the hit child with the smallest entry distance, 8: none
};
#define CW_CHILD(k)                                                                                                                                            \
  {                                                                                                                                                            \
    const float tn = fmaxf(fmaxf(__builtin_fmaf(cw_plane(cw_word(nx, (k) >> 1), (k) & 1), sx, blx), __builtin_fmaf(cw_plane(cw_word(ny, (k) >> 1), (k) & 1), sy, bly)), \
                           fmaxf(__builtin_fmaf(cw_plane(cw_word(nz, (k) >> 1), (k) & 1), sz, blz), 0.0f));                                                    \
    const float tf = fminf(fminf(__builtin_fmaf(cw_plane(cw_word(fx, (k) >> 1), (k) & 1), sx, bhx), __builtin_fmaf(cw_plane(cw_word(fy, (k) >> 1), (k) & 1), sy, bhy)), \
                           fminf(__builtin_fmaf(cw_plane(cw_word(fz, (k) >> 1), (k) & 1), sz, bhz), lim));                                                     \
    /* the sign bit of tf - tn says "missed"; the 8 bits are shifted together with one v_alignbit_b32 each (children 7 .. 0: child k ends at bit k) */           \
    const float gap = tf - tn;                                                                                                                                 \
    miss            = (miss << 1) | (__float_as_uint(gap) >> 31);                                                                                              \
    const bool nearer = gap >= 0.0f && tn < tnear;                                                                                                             \
    tnear   = nearer ? tn : tnear;                                                                                                                             \
    nearest = nearer ? uint32_t(k) : nearest;                                                                                                                  \
  }
PT_DEV NodeHit cw_test_node(const CwNode* __restrict__ nodes, uint32_t idx, const BoxRay& R, float lim, bool alphaOnly)
{
  const char*    nb = reinterpret_cast<const char*>(nodes);
  const uint32_t at = idx * uint32_t(CW_NODE_BYTES);  // 32-bit byte offsets (the node array is < 4 GB)
  // near / far plane sets by the direction sign, chosen by ADDRESS
  const uint32_t ox = R.idir.x < 0.0f ? 48u : 0u, oy = R.idir.y < 0.0f ? 48u : 0u, oz = R.idir.z < 0.0f ? 48u : 0u;
  const uint4    h0 = *reinterpret_cast<const uint4*>(nb + at), h1 = *reinterpret_cast<const uint4*>(nb + (at + 16u));
  const uint4    nx = *reinterpret_cast<const uint4*>(nb + (at + CW_OFF_QLO + ox)), fx = *reinterpret_cast<const uint4*>(nb + (at + CW_OFF_QHI - ox));
  const uint4    ny = *reinterpret_cast<const uint4*>(nb + (at + CW_OFF_QLO + 16u + oy)), fy = *reinterpret_cast<const uint4*>(nb + (at + CW_OFF_QHI + 16u - oy));
  const uint4    nz = *reinterpret_cast<const uint4*>(nb + (at + CW_OFF_QLO + 32u + oz)), fz = *reinterpret_cast<const uint4*>(nb + (at + CW_OFF_QHI + 32u - oz));
  NodeHit        nh;
  nh.childBase = h1.x & CW_CHILD_MASK;
  nh.triBase   = h1.y;
  nh.kinds     = (h0.w >> 24) | ((h1.z & 0xffffu) << 8);
  const uint32_t amask = h1.x >> 24;
  // per-axis grid step and origin in ray parameters
  const float sx = __uint_as_float((h0.w & 0xffu) << 23) * R.idir.x, sy = __uint_as_float(((h0.w >> 8) & 0xffu) << 23) * R.idir.y, sz = __uint_as_float(((h0.w >> 16) & 0xffu) << 23) * R.idir.z;
  const float bx = (__uint_as_float(h0.x) - R.o.x) * R.idir.x, by = (__uint_as_float(h0.y) - R.o.y) * R.idir.y, bz = (__uint_as_float(h0.z) - R.o.z) * R.idir.z;
  const float gmax = float(CW_GRID_MAX);
  const float ex = __builtin_fmaf(fabsf(R.idir.x), R.eps, (fabsf(bx) + gmax * fabsf(sx)) * 8.0e-7f), ey = __builtin_fmaf(fabsf(R.idir.y), R.eps, (fabsf(by) + gmax * fabsf(sy)) * 8.0e-7f),
              ez = __builtin_fmaf(fabsf(R.idir.z), R.eps, (fabsf(bz) + gmax * fabsf(sz)) * 8.0e-7f);
  const float blx = bx - ex, bhx = bx + ex, bly = by - ey, bhy = by + ey, blz = bz - ez, bhz = bz + ez;
  // children that count: inner or leaf (an empty slot's inverted box can never), and for the alpha-only walks only the tagged ones
  const uint32_t valid = (nh.kinds | (nh.kinds >> 8)) & (alphaOnly ? amask : 0xffu) & 0xffu;
  uint32_t    miss = 0, nearest = 0;
  float       tnear = 3.0e38f;
  CW_CHILD(7) CW_CHILD(6) CW_CHILD(5) CW_CHILD(4) CW_CHILD(3) CW_CHILD(2) CW_CHILD(1) CW_CHILD(0)
  const uint32_t hits = ~miss & valid;
  nh.alphaHits = hits & amask;
  nh.hits      = hits;
  nh.nearest   = ((hits >> nearest) & 1u) ? nearest : 8u;  // (8: none -- nothing hit, or the nearest box belongs to a child that does not count)
  return nh;
}
#undef CW_CHILD

// the 8 hit bits from slot order into visiting order: bit k -> bit k ^ octinv (three conditional swaps of neighbours, pairs, nibbles)
PT_DEV uint32_t cw_visit_order(uint32_t m, uint32_t octinv)
{
  m = (octinv & 1u) ? (((m & 0x55u) << 1) | ((m >> 1) & 0x55u)) : m;
  m = (octinv & 2u) ? (((m & 0x33u) << 2) | ((m >> 2) & 0x33u)) : m;
  m = (octinv & 4u) ? (((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu)) : m;
  return m;
}

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
  uint32_t gx, gy, gz;     // the group: first inner child | hit bits 31..24 (visiting order), first triangle (two-level, TLAS level: first TlasLeaf),
                           // inner mask | leaf mask << 8 | two-triangle-leaf mask << 16 (slot order) | (slot + 1) << 24 of the child to take
                           // BEFORE the hit bits: the nearest hit child of the node just visited.  Nothing pending, no hit bits: group exhausted.
  int      sp;
  uint32_t flags, cnt, wLimit;
  float    zeroMaxT, zeroMaxT2, zeroMaxT3;  // pass A: the three largest t among the zero-opacity candidates seen
  float    tPrev;          // TM_RAW_* / TM_PICK: exclusive lower key
  uint32_t wPrev;
  int      pass;           // TM_MACHINE: 0 = pass A (nearest certain hit), 1 = pass B (count zero-opacity candidates in front of it)
  bool     done;
  bool     sawAlpha;       // a box tagged non-opaque was hit (shadow rays: without one, the first certain hit is final -- see lane_triangle)
  bool     anyEnds;        // shadow ray: the first certain hit may end the walk when no non-opaque geometry was met (nothing can consume a draw)
  InstCtx  ic;             // two-level instantiations only: the instance the lane is inside of
  uint32_t steps;          //   and the loop-iteration guard
#ifdef PT_STATS
  uint32_t nNodes, nTris;
#endif
};

// a group whose only child is "inner child 0 of base `node`" = that node, pending as the nearest
PT_DEV void lane_enter_node(TraceLane& L, uint32_t node)
{
  L.gx = node;
  L.gy = 0u;
  L.gz = 1u | (1u << 24);
}
PT_DEV void lane_begin(TraceLane& L, f3 o, f3 d, float tmax, bool emptyScene)
{
  L.o = o; L.d = d;
  L.R = make_boxray(o, d);
  L.tmax = tmax; L.bt = tmax; L.bu = 0.f; L.bv = 0.f; L.bslot = BVH_NONE; L.bw = 0xffffffffu;
  lane_enter_node(L, 0u);
  L.sp = 0;
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
  lane_enter_node(L, 0u);
  L.sp = 0;
  L.flags = 0; L.cnt = 0; L.pass = 1; L.done = false;
}

// One triangle of a leaf under the candidate rules of `mode`.
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
        // Shadow rays (T6: the nearest certain hit ends the ray, zero-opacity candidates in front of it consume draws): as long as the ray has
        // hit no box tagged non-opaque, no candidate anywhere along it can consume a draw and ANY certain hit gives the same verdict and the
        // same RNG state as the nearest one -- the reference's own gl_RayFlagsTerminateOnFirstHitEXT (traceray_rq.glsl:157).  Sound because
        // the boxes a ray has not been tested against lie inside boxes it hit: an untagged box holds opaque triangles only.
        if(L.anyEnds && !L.sawAlpha)
          L.done = true;
      }
    }
  }
}

// Visiting order of a group: first the pending nearest child (exact: smallest entry distance of the node's hit children), then the other hit
// children in octant order (highest hit bit first).  Going to the nearest child first is what makes the bound shrink early: octant order
// alone tested 2x the triangles of a nearest-first walk (tools/steps_experiment.py).
PT_DEV bool group_has_next(const TraceLane& L) { return ((L.gx >> 24) | (L.gz >> 24)) != 0u; }
PT_DEV uint32_t group_next_slot(const TraceLane& L)
{
  const uint32_t pending = L.gz >> 24;
  return pending ? pending - 1u : (((31u - uint32_t(__clz(int(L.gx)))) - 24u) ^ L.R.octinv);
}
PT_DEV void group_drop_next(TraceLane& L)
{
  if(L.gz >> 24)
    L.gz &= 0x00ffffffu;
  else
    L.gx &= ~(1u << (31u - uint32_t(__clz(int(L.gx)))));
}

// One step of the walk: the next child of the lane's group if it is a leaf (its one or two triangles), then the next child if it is an inner
// node (its visit); an exhausted group is replaced from the stack.  Sets L.done when nothing is left.
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
  // ---- a leaf, if it is the group's next child
  if(group_has_next(L))
  {
    const uint32_t slot = group_next_slot(L);
    if((L.gz >> (8u + slot)) & 1u)
    {
      group_drop_next(L);
      const uint32_t below = (1u << slot) - 1u;
      const uint32_t first = L.gy + uint32_t(__popc((L.gz >> 8) & below & 0xffu)) + uint32_t(__popc((L.gz >> 16) & below & 0xffu));
      if(TWO && L.ic.inst == BVH_NONE)
      {  // TLAS level: the leaf is an instance.  What is left of the group waits on the stack.
        const TlasLeaf tl = S.tlasLeaves[first];
        if(L.gx >> 24)
          stack_push(st, L.sp, L.gx, L.gy, L.gz, counters);
        L.ic = InstCtx{tl.inst, L.sp, tl.wflags};
        L.R  = enter_instance(S, tl, L.o, L.d);
        lane_enter_node(L, tl.nodeBase);  // the BLAS root
      }
      else
      {
        const bool   two = ((L.gz >> (16u + slot)) & 1u) != 0u;
        TriRec       a = S.tris[first];
        const TriRec b = S.tris[first + (two ? 1u : 0u)];  // six 16-byte loads in flight together
        // ONE copy of the triangle code (it contains the any-hit evaluation with its texture filtering: twice inlined it doubled the
        // kernels to ~40 KB each, against an instruction cache that the shade kernel's waves on the same CUs compete for)
#pragma unroll 1
        for(uint32_t j = 0;; ++j)
        {
          lane_triangle<MODE, TWO>(S, L, first + j, a);
          if(!two || j == 1u || L.done)
            break;
          a = b;
        }
        if(L.done)
          return;
      }
    }
  }
  // ---- an exhausted group is replaced from the stack
  if(!group_has_next(L))
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
    stack_pop(st, L.sp, L.gx, L.gy, L.gz);
  }
  // ---- an inner node, if it is the group's next child
  const uint32_t slot = group_next_slot(L);
  if(!((L.gz >> slot) & 1u))
    return;  // a leaf: the next step takes it
  group_drop_next(L);
  const uint32_t child = (L.gx & CW_CHILD_MASK) + uint32_t(__popc(L.gz & ((1u << slot) - 1u) & 0xffu));
  if(L.gx >> 24)
    stack_push(st, L.sp, L.gx, L.gy, L.gz, counters);
#ifdef PT_STATS
  ++L.nNodes;
#endif
  const float   lim = mode == TM_COUNT ? L.tmax : L.bt;
  const NodeHit nh  = cw_test_node((TWO && L.ic.inst == BVH_NONE) ? S.tlas : S.wide, child, L.R, lim, alphaOnly);
  L.gx = nh.childBase | (cw_visit_order(nh.hits & ~(1u << nh.nearest), L.R.octinv) << 24);  // (1 << 8 for "none" clears nothing)
  L.gy = nh.triBase;
  L.gz = nh.kinds | (nh.nearest < 8u ? (nh.nearest + 1u) << 24 : 0u);
  L.sawAlpha = L.sawAlpha || nh.alphaHits != 0u;
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
