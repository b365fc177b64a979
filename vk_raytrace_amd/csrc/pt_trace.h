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
//    draw per candidate in key order (T5 closest, T6 shadow).
//
// Stochastic alpha without re-traversal.  Processing candidates strictly in key order would cost one
// full traversal per rejected candidate (a ray through foliage meets dozens of transparent texels).
// The same result is obtained with at most two traversals:
//   pass A (TM_CLOSEST / TM_SHADOW) finds the nearest CERTAIN hit -- opaque, or opacity >= 1 (the draw
//          r < 1 can never exceed it) -- evaluating the opacity of non-opaque candidates on the fly;
//   pass B (TM_COUNT) only counts the zero-opacity candidates in front of it.  Each of them consumes one
//          draw and is rejected unless that draw is exactly 0.0 (probability 2^-23).
// The caller advances the RNG by the count; if a draw is 0.0, or a candidate with fractional opacity
// (ALPHA_BLEND) lies in front of the certain hit, it falls back to the exact key-ordered loop
// (TM_RAW_*).  All three routes produce identical hits and identical RNG states.
//
// Per-lane traversal stack: the first STACK_LDS entries live in LDS laid out [level][lane] (one bank
// per lane, conflict free: 64 lanes x 4 B = one 256-byte bank row per level), deeper entries spill to
// a small private array.  LBVH depth is unbounded in theory; overflow beyond STACK_LDS+STACK_SPILL is
// counted in Counters::stackOverflow and reported as an error by pt_get_stats.
#pragma once
#include "pt_surface.h"

#define TRACE_BLOCK 64
#ifndef STACK_LDS
#define STACK_LDS 32
#endif
#define STACK_SPILL (64 - STACK_LDS)

enum TraceMode {
  TM_RAW_ALL = 0,        // exact: smallest key > (tPrev,wPrev), every triangle is a candidate, no opacity evaluation
  TM_RAW_NONOPAQUE = 1,  // exact: same, non-opaque triangles only
  TM_CLOSEST = 2,        // pass A of ClosestHit: nearest certain hit, flags for uncertain candidates in front of it
  TM_SHADOW = 3,         // pass A of AnyHit: any opaque hit ends the ray; else nearest certain non-opaque hit + flags
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
  uint32_t count;  // TM_COUNT: zero-opacity candidates in range.  TM_CLOSEST / TM_SHADOW: zero-opacity candidates SEEN
  float    zeroMaxT;  // TM_CLOSEST / TM_SHADOW: largest t among the zero-opacity candidates seen (-1: none)
  float    zeroMaxT2, zeroMaxT3;  //                        second and third largest
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
// T2: Moeller-Trumbore in the fixed operation order of the trace contract.
// (A "certified" form -- accepted candidates kept only when a forward error bound of the same fp32 evaluation certifies them -- closes the one known
// flat != two-level pixel-sample on the CPU harness and costs 7 % on the GPU: DESIGN.md section 3, profiles/r04k_t2_certified_gpu.txt; a robust fp32 / fp64
// form costs 33 %, profiles/r03h_*.  The contract stays plain T2.)
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
  const float nt = dot3(e2, qv);
  t              = nt * inv;
  return true;
}
#endif

#if PT_BVH_WIDTH != 2
// Per-ray constants of the slab test.  The test runs in the fused form t = plane * idir + n with n = -(o * idir):
// one FMA per plane instead of a subtract and a multiply, and -- because the sign of idir says which plane of a slab
// is the near one -- no per-axis min / max: the near and far planes are fetched from sign-dependent offsets inside
// the node.  The box test is not part of the bit-exact contract (results are BVH independent), it only has to be
// conservative: the fused form adds an absolute error of |o * idir| * 2^-24 per axis (cancellation), absorbed by
// biasing n by E = |o * idir| * 2^-21 towards "hit" (nlo for near planes, nhi for far planes); the relative part
// stays covered by the (1 -+ 4e-7) factors.  |d| components below 1e-18 are clamped so that idir stays finite
// (a ray moves < 1 ulp along such an axis over any representable distance).
struct RayBox {
  f3       idir, nlo, nhi;
  uint32_t nearOff[3];  // byte offset of the near-plane quadruple of each axis inside a WideNode (far = the other one)
};
PT_DEV RayBox make_raybox(f3 o, f3 d)
{
  RayBox      rb;
  const float dx = copysignf(fmaxf(fabsf(d.x), 1.0e-18f), d.x), dy = copysignf(fmaxf(fabsf(d.y), 1.0e-18f), d.y), dz = copysignf(fmaxf(fabsf(d.z), 1.0e-18f), d.z);
  rb.idir        = f3{1.0f / dx, 1.0f / dy, 1.0f / dz};
  const f3 oi    = f3{o.x * rb.idir.x, o.y * rb.idir.y, o.z * rb.idir.z};
  const f3 e     = f3{fabsf(oi.x) * 4.76837158e-7f, fabsf(oi.y) * 4.76837158e-7f, fabsf(oi.z) * 4.76837158e-7f};
  rb.nlo         = f3{-oi.x - e.x, -oi.y - e.y, -oi.z - e.z};
  rb.nhi         = f3{-oi.x + e.x, -oi.y + e.y, -oi.z + e.z};
  rb.nearOff[0]  = rb.idir.x < 0.0f ? 48u : 0u;
  rb.nearOff[1]  = rb.idir.y < 0.0f ? 48u : 0u;
  rb.nearOff[2]  = rb.idir.z < 0.0f ? 48u : 0u;
  return rb;
}

template <class Push>
PT_DEV uint32_t wide_node_decide(const float* nx, const float* fx, const float* ny, const float* fy, const float* nz, const float* fz, const uint32_t* cc, float lim, bool alphaOnly,
                                 Push&& push);

// One wide-node visit: slab-tests the 4 child boxes against [0, lim], pushes the hit children far-to-near through
// `push` and returns the nearest one (BVH_NONE when nothing is hit).  Empty slots carry inverted infinite boxes.
// alphaOnly: visit only children tagged BVH_ALPHA (pass B and the non-opaque fallback never need an opaque subtree).
template <class Push>
PT_DEV uint32_t wide_node_step(const WideNode* __restrict__ nodes, uint32_t node, const RayBox& rb, float lim, bool alphaOnly, Push&& push)
{
  static_assert(PT_BVH_WIDTH == 4, "the fused slab test is written for 4-wide nodes");
  const char*    nb = reinterpret_cast<const char*>(nodes);
  const uint32_t at = (node & BVH_SLOT_MASK) << 7;  // sizeof(WideNode) == 128; 32-bit byte offsets (the node array is < 4 GB)
  const float4   px = *reinterpret_cast<const float4*>(nb + (at + rb.nearOff[0])), qx = *reinterpret_cast<const float4*>(nb + (at + 48u - rb.nearOff[0]));
  const float4   py = *reinterpret_cast<const float4*>(nb + (at + 16u + rb.nearOff[1])), qy = *reinterpret_cast<const float4*>(nb + (at + 64u - rb.nearOff[1]));
  const float4   pz = *reinterpret_cast<const float4*>(nb + (at + 32u + rb.nearOff[2])), qz = *reinterpret_cast<const float4*>(nb + (at + 80u - rb.nearOff[2]));
  const uint4    ch = *reinterpret_cast<const uint4*>(nb + (at + 96u));
  const float    nx[4] = {__builtin_fmaf(px.x, rb.idir.x, rb.nlo.x), __builtin_fmaf(px.y, rb.idir.x, rb.nlo.x), __builtin_fmaf(px.z, rb.idir.x, rb.nlo.x), __builtin_fmaf(px.w, rb.idir.x, rb.nlo.x)};
  const float    fx[4] = {__builtin_fmaf(qx.x, rb.idir.x, rb.nhi.x), __builtin_fmaf(qx.y, rb.idir.x, rb.nhi.x), __builtin_fmaf(qx.z, rb.idir.x, rb.nhi.x), __builtin_fmaf(qx.w, rb.idir.x, rb.nhi.x)};
  const float    ny[4] = {__builtin_fmaf(py.x, rb.idir.y, rb.nlo.y), __builtin_fmaf(py.y, rb.idir.y, rb.nlo.y), __builtin_fmaf(py.z, rb.idir.y, rb.nlo.y), __builtin_fmaf(py.w, rb.idir.y, rb.nlo.y)};
  const float    fy[4] = {__builtin_fmaf(qy.x, rb.idir.y, rb.nhi.y), __builtin_fmaf(qy.y, rb.idir.y, rb.nhi.y), __builtin_fmaf(qy.z, rb.idir.y, rb.nhi.y), __builtin_fmaf(qy.w, rb.idir.y, rb.nhi.y)};
  const float    nz[4] = {__builtin_fmaf(pz.x, rb.idir.z, rb.nlo.z), __builtin_fmaf(pz.y, rb.idir.z, rb.nlo.z), __builtin_fmaf(pz.z, rb.idir.z, rb.nlo.z), __builtin_fmaf(pz.w, rb.idir.z, rb.nlo.z)};
  const float    fz[4] = {__builtin_fmaf(qz.x, rb.idir.z, rb.nhi.z), __builtin_fmaf(qz.y, rb.idir.z, rb.nhi.z), __builtin_fmaf(qz.z, rb.idir.z, rb.nhi.z), __builtin_fmaf(qz.w, rb.idir.z, rb.nhi.z)};
  const uint32_t cc[4] = {ch.x, ch.y, ch.z, ch.w};
  return wide_node_decide(nx, fx, ny, fy, nz, fz, cc, lim, alphaOnly, push);
}

// second half of a node visit: from the six ray parameters per child to "push the hit children far-to-near, return the nearest"
template <class Push>
PT_DEV uint32_t wide_node_decide(const float* nx, const float* fx, const float* ny, const float* fy, const float* nz, const float* fz, const uint32_t* cc, float lim, bool alphaOnly,
                                 Push&& push)
{
  float          tn[4];
  uint32_t       cid[4];
  int            nh = 0;
#pragma unroll
  for(int k = 0; k < 4; ++k)
  {
    const float nr = fmaxf(fmaxf(nx[k], ny[k]), fmaxf(nz[k], 0.0f)) * 0.9999996f;
    const float fr = fminf(fminf(fx[k], fy[k]), fminf(fz[k], lim)) * 1.0000004f;
    const bool  h  = (nr <= fr) && (cc[k] != BVH_NONE) && (!alphaOnly || (cc[k] & BVH_ALPHA));
    tn[k]  = h ? nr : 3.0e38f;
    cid[k] = cc[k];
    nh += h ? 1 : 0;
  }
  if(nh == 0)
    return BVH_NONE;
  // push all but the nearest, farthest first
  for(int p = 0; p < PT_BVH_WIDTH - 1 && nh > 1; ++p, --nh)
  {
    float    mt  = -1.0f;
    uint32_t mid = 0;
    int      ms  = 0;
#pragma unroll
    for(int i = 0; i < PT_BVH_WIDTH; ++i)
    {
      bool g = tn[i] < 3.0e38f && tn[i] >= mt;
      mt     = g ? tn[i] : mt;
      mid    = g ? cid[i] : mid;
      ms     = g ? i : ms;
    }
    push(mid);
#pragma unroll
    for(int i = 0; i < PT_BVH_WIDTH; ++i)
      tn[i] = (i == ms) ? 3.0e38f : tn[i];
  }
  uint32_t nearest = BVH_NONE;
#pragma unroll
  for(int i = 0; i < PT_BVH_WIDTH; ++i)
    nearest = tn[i] < 3.0e38f ? cid[i] : nearest;
  return nearest;
}

// fp16 grid coordinate out of a register half (device: folds into v_fma_mix_f32's op_sel)
#if defined(__HIP_DEVICE_COMPILE__)
typedef _Float16 pt_h2v __attribute__((ext_vector_type(2)));
PT_DEV float cn_plane(uint32_t word, int hi)
{
  const pt_h2v h = __builtin_bit_cast(pt_h2v, word);
  return float(hi ? h.y : h.x);
}
#else
PT_DEV float cn_plane(uint32_t word, int hi)
{
  const uint32_t h = (hi ? (word >> 16) : word) & 0xffffu;  // non-negative normal halves and zero only: what the nodes hold
  return h == 0 ? 0.0f : __uint_as_float((((h >> 10) + 112u) << 23) | ((h & 0x3ffu) << 13));
}
#endif
// The same visit on the compact form of the node (pt_device.h CompactNode): five 16-byte requests.  t = q * s + b with s = step * idir (exact: the
// step is a power of two) and b = p * idir + n in one FMA; b is biased by (|b| + 2047 |s|) * 8e-7 towards "hit" (its own rounding is 2^-24 |b|, the
// plane FMA's 2^-24 |t| with |t| <= |b| + 2047 |s|; the (1 -+ 4e-7) factors of the decision stay on top).  The planes of a slab share one request, so
// near / far are picked by the direction sign with selects.
template <class Push>
PT_DEV uint32_t cnode_visit(float4 h, uint4 X, uint4 Y, uint4 Z, uint4 ch, const RayBox& rb, float lim, bool alphaOnly, Push&& push);
template <class Push>
PT_DEV uint32_t wide_node_step_c(const CompactNode* __restrict__ nodes, uint32_t node, const RayBox& rb, float lim, bool alphaOnly, Push&& push)
{
  const char*    nb = reinterpret_cast<const char*>(nodes);
  const uint32_t at = (node & BVH_SLOT_MASK) * uint32_t(sizeof(CompactNode));
  const float4   h  = *reinterpret_cast<const float4*>(nb + at);
  const uint4    X = *reinterpret_cast<const uint4*>(nb + (at + 16u)), Y = *reinterpret_cast<const uint4*>(nb + (at + 32u)), Z = *reinterpret_cast<const uint4*>(nb + (at + 48u));
  const uint4    ch = *reinterpret_cast<const uint4*>(nb + (at + 64u));
  return cnode_visit(h, X, Y, Z, ch, rb, lim, alphaOnly, push);
}
// the visit proper, on the five quads of the node already in registers
template <class Push>
PT_DEV uint32_t cnode_visit(float4 h, uint4 X, uint4 Y, uint4 Z, uint4 ch, const RayBox& rb, float lim, bool alphaOnly, Push&& push)
{
  const uint32_t ex = __float_as_uint(h.w);
  const float    sx = __uint_as_float((ex & 0xffu) << 23) * rb.idir.x, sy = __uint_as_float(((ex >> 8) & 0xffu) << 23) * rb.idir.y, sz = __uint_as_float(((ex >> 16) & 0xffu) << 23) * rb.idir.z;
  const float    blx0 = __builtin_fmaf(h.x, rb.idir.x, rb.nlo.x), bhx0 = __builtin_fmaf(h.x, rb.idir.x, rb.nhi.x);
  const float    bly0 = __builtin_fmaf(h.y, rb.idir.y, rb.nlo.y), bhy0 = __builtin_fmaf(h.y, rb.idir.y, rb.nhi.y);
  const float    blz0 = __builtin_fmaf(h.z, rb.idir.z, rb.nlo.z), bhz0 = __builtin_fmaf(h.z, rb.idir.z, rb.nhi.z);
  const float    gm = float(CN_GRID_MAX);
  const float    blx = blx0 - (fabsf(blx0) + gm * fabsf(sx)) * 8.0e-7f, bhx = bhx0 + (fabsf(bhx0) + gm * fabsf(sx)) * 8.0e-7f;
  const float    bly = bly0 - (fabsf(bly0) + gm * fabsf(sy)) * 8.0e-7f, bhy = bhy0 + (fabsf(bhy0) + gm * fabsf(sy)) * 8.0e-7f;
  const float    blz = blz0 - (fabsf(blz0) + gm * fabsf(sz)) * 8.0e-7f, bhz = bhz0 + (fabsf(bhz0) + gm * fabsf(sz)) * 8.0e-7f;
  const bool     ngx = rb.nearOff[0] != 0, ngy = rb.nearOff[1] != 0, ngz = rb.nearOff[2] != 0;  // negative direction: the upper plane is the near one
  const uint32_t nX0 = ngx ? X.z : X.x, nX1 = ngx ? X.w : X.y, fX0 = ngx ? X.x : X.z, fX1 = ngx ? X.y : X.w;
  const uint32_t nY0 = ngy ? Y.z : Y.x, nY1 = ngy ? Y.w : Y.y, fY0 = ngy ? Y.x : Y.z, fY1 = ngy ? Y.y : Y.w;
  const uint32_t nZ0 = ngz ? Z.z : Z.x, nZ1 = ngz ? Z.w : Z.y, fZ0 = ngz ? Z.x : Z.z, fZ1 = ngz ? Z.y : Z.w;
  const float    nx[4] = {__builtin_fmaf(cn_plane(nX0, 0), sx, blx), __builtin_fmaf(cn_plane(nX0, 1), sx, blx), __builtin_fmaf(cn_plane(nX1, 0), sx, blx), __builtin_fmaf(cn_plane(nX1, 1), sx, blx)};
  const float    fx[4] = {__builtin_fmaf(cn_plane(fX0, 0), sx, bhx), __builtin_fmaf(cn_plane(fX0, 1), sx, bhx), __builtin_fmaf(cn_plane(fX1, 0), sx, bhx), __builtin_fmaf(cn_plane(fX1, 1), sx, bhx)};
  const float    ny[4] = {__builtin_fmaf(cn_plane(nY0, 0), sy, bly), __builtin_fmaf(cn_plane(nY0, 1), sy, bly), __builtin_fmaf(cn_plane(nY1, 0), sy, bly), __builtin_fmaf(cn_plane(nY1, 1), sy, bly)};
  const float    fy[4] = {__builtin_fmaf(cn_plane(fY0, 0), sy, bhy), __builtin_fmaf(cn_plane(fY0, 1), sy, bhy), __builtin_fmaf(cn_plane(fY1, 0), sy, bhy), __builtin_fmaf(cn_plane(fY1, 1), sy, bhy)};
  const float    nz[4] = {__builtin_fmaf(cn_plane(nZ0, 0), sz, blz), __builtin_fmaf(cn_plane(nZ0, 1), sz, blz), __builtin_fmaf(cn_plane(nZ1, 0), sz, blz), __builtin_fmaf(cn_plane(nZ1, 1), sz, blz)};
  const float    fz[4] = {__builtin_fmaf(cn_plane(fZ0, 0), sz, bhz), __builtin_fmaf(cn_plane(fZ0, 1), sz, bhz), __builtin_fmaf(cn_plane(fZ1, 0), sz, bhz), __builtin_fmaf(cn_plane(fZ1, 1), sz, bhz)};
  const uint32_t cc[4] = {ch.x, ch.y, ch.z, ch.w};
  return wide_node_decide(nx, fx, ny, fy, nz, fz, cc, lim, alphaOnly, push);
}
#endif

#if PT_BVH_WIDTH != 2
// ---- two-level walk (TLAS over instances, one object-space BLAS per prim-mesh; reference: src/accelstruct.cpp:110-162) -----------
// A lane is either at TLAS level (InstCtx::inst == BVH_NONE: world-space ray constants, node references index DeviceScene::tlas) or inside
// one instance (object-space ray constants, references index DeviceScene::wide / tris).  The ray parameter t is the same in both spaces
// (the direction is transformed, not renormalised), so the current bound prunes in either.  An instance is left when the traversal stack
// has shrunk back to the depth it had when the instance was entered.
struct InstCtx {
  uint32_t inst;    // BVH_NONE: at TLAS level
  int      spBase;  // stack depth at entry
  uint32_t wflags;  // world index of the instance's first triangle | TRI_* flags << 29
};
#define PT_TWO_GUARD (1u << 20)  // loop-iteration bound of the two-level walks (a corrupt structure must not hang the GPU; reported as a stack overflow)

// Object-space ray constants of a ray entering the instance of TLAS leaf `tl`.  Transforming the ray rounds (o' and d' carry an absolute
// error of a few 2^-24 x |worldToObject| x (|o| + |hit point|)); instead of tracking it per plane the BLAS boxes are grown by
// eps = padC1 * max|o| + padC0 (host-computed bound with a 16x margin, pt_capi.hip: two_level_pad), folded into the per-ray constants:
// (plane -+ eps) * idir + n  =  plane * idir + (n -+ eps * |idir|).
PT_DEV RayBox enter_instance(const DeviceScene& S, const TlasLeaf& tl, f3 o, f3 d)
{
  const Affine W  = S.instances[tl.inst].worldToObject;
  RayBox       rb = make_raybox(xform_point(W, o), xform_dir(W, d));
  const float  eps = tl.padC1 * fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))) + tl.padC0;
  const f3     g   = f3{eps * fabsf(rb.idir.x), eps * fabsf(rb.idir.y), eps * fabsf(rb.idir.z)};
  rb.nlo = rb.nlo - g;
  rb.nhi = rb.nhi + g;
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
  if(S.instBlock)
  {  // both ends from the block table (two independent loads): the answer lies between the last instance that starts at or before the block's first
     // triangle and the last one that starts at or before the next block's
    const uint32_t b = w >> PT_INST_BLOCK_SHIFT;
    lo               = S.instBlock[b];
    hi               = S.instBlock[b + 1];
  }
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
#endif

#ifdef PT_HIST
// measurement build only (tools/gpu_hist.py): per traversal mode, the distribution of per-ray loop iterations
// ([0..31]: floor(log2)+1 buckets) and the wave-level lane utilisation ([32] sum of iterations, [33] sum over waves
// of max x 64, [34] rays, [35] waves, [36] sum over waves of max)
__device__ unsigned long long g_hist[8][40];  // rows 5 / 6: persistent closest / shadow kernels (see pt_render.hip)
#endif

// tPrev/wPrev: exclusive lower key (TM_RAW_*); wLimit: with tmax the exclusive upper key (TM_COUNT).
// `opaqueHit` is only meaningful for TM_SHADOW.
// TWO: the two-level structure (see above); RayHit::slot is then the global BLAS leaf slot, RayHit::w the world index | flags as always.
template <int MODE, bool TWO = false>
PT_DEV void traverse(const DeviceScene& S, f3 o, f3 d, float tmax, float tPrev, uint32_t wPrev, uint32_t wLimit, uint32_t* ldsStack, RayHit& best, bool& opaqueHit,
                     Counters* counters)
{
  best.slot  = BVH_NONE;
  best.t     = tmax;
  best.w     = 0xffffffffu;
  best.flags    = 0;
  best.count    = 0;
  best.zeroMaxT = -1.0f;
  best.zeroMaxT2 = -1.0f;
  best.zeroMaxT3 = -1.0f;
  opaqueHit     = false;
  if(S.numTris == 0)
    return;

#if PT_BVH_WIDTH != 2
  RayBox          rbox = make_raybox(o, d);
  const RayBox    rboxW = rbox;                      // (TWO) the world-space constants, restored when an instance is left
  const WideNode* nodes = TWO ? S.tlas : S.wide;
  InstCtx         ic{BVH_NONE, 0, 0u};
  uint32_t        guard = 0;
#else
  static_assert(!TWO, "the two-level walk is written for the 4-wide layout");
  const f3 idir = f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
#endif
  // TM_SHADOW must keep looking for opaque triangles behind the best alpha candidate; TM_COUNT has a fixed range
#define PT_TLIMIT ((MODE == TM_SHADOW || MODE == TM_COUNT) ? tmax : best.t)
  uint32_t spill[STACK_SPILL];
  int      sp  = 0;
  uint32_t cur = 0;  // root (inner node 0; a one-triangle scene has a single node with one leaf child)
#ifdef PT_STATS
  uint32_t nNodes = 0, nTris = 0;
#endif
#ifdef PT_HIST
  uint32_t nIter = 0;
#endif

  for(;;)
  {
#ifdef PT_HIST
    ++nIter;
#endif
    if(!(cur & BVH_LEAF))
    {
#ifdef PT_STATS
      ++nNodes;
#endif
#if PT_BVH_WIDTH != 2
      if(TWO && ++guard > PT_TWO_GUARD)
      {
        atomicAdd(&counters->stackOverflow, 1u);
        break;
      }
      const uint32_t nxt = wide_node_step(nodes, cur, rbox, PT_TLIMIT, MODE == TM_COUNT || MODE == TM_RAW_NONOPAQUE, [&](uint32_t c) {
        if(sp < STACK_LDS)
          ldsStack[sp++ * TRACE_BLOCK] = c;
        else if(sp < STACK_LDS + STACK_SPILL)
          spill[sp++ - STACK_LDS] = c;
        else
          atomicAdd(&counters->stackOverflow, 1u);  // child dropped (flagged; pt_get_stats reports it)
      });
      if(nxt != BVH_NONE)
      {
        cur = nxt;
        continue;
      }
#else
      const BvhNode* np = S.bvh + (cur & BVH_SLOT_MASK);
      const float4   a = np->a, b = np->b, c = np->c;
      const uint4    ch = np->d;
      // slab test of both children; (bound - o) * idir keeps NaN confined to the degenerate 0*inf case,
      // which fminf/fmaxf (IEEE minNum/maxNum) then ignore -> conservative
      float lx0 = (a.x - o.x) * idir.x, lx1 = (a.w - o.x) * idir.x;
      float ly0 = (a.y - o.y) * idir.y, ly1 = (b.x - o.y) * idir.y;
      float lz0 = (a.z - o.z) * idir.z, lz1 = (b.y - o.z) * idir.z;
      float rx0 = (b.z - o.x) * idir.x, rx1 = (c.y - o.x) * idir.x;
      float ry0 = (b.w - o.y) * idir.y, ry1 = (c.z - o.y) * idir.y;
      float rz0 = (c.x - o.z) * idir.z, rz1 = (c.w - o.z) * idir.z;
      float lnear = fmaxf(fmaxf(fminf(lx0, lx1), fminf(ly0, ly1)), fmaxf(fminf(lz0, lz1), 0.0f)) * 0.9999996f;
      float lfar  = fminf(fminf(fmaxf(lx0, lx1), fmaxf(ly0, ly1)), fminf(fmaxf(lz0, lz1), PT_TLIMIT)) * 1.0000004f;
      float rnear = fmaxf(fmaxf(fminf(rx0, rx1), fminf(ry0, ry1)), fmaxf(fminf(rz0, rz1), 0.0f)) * 0.9999996f;
      float rfar  = fminf(fminf(fmaxf(rx0, rx1), fmaxf(ry0, ry1)), fminf(fmaxf(rz0, rz1), PT_TLIMIT)) * 1.0000004f;
      bool  hl = lnear <= lfar, hr = (rnear <= rfar) && (ch.y != BVH_NONE);
      if(hl && hr)
      {
        uint32_t nearC = ch.x, farC = ch.y;
        if(rnear < lnear)
        {
          nearC = ch.y;
          farC  = ch.x;
        }
        if(sp < STACK_LDS)
          ldsStack[sp++ * TRACE_BLOCK] = farC;
        else if(sp < STACK_LDS + STACK_SPILL)
          spill[sp++ - STACK_LDS] = farC;
        else
          atomicAdd(&counters->stackOverflow, 1u);  // far child dropped (flagged; pt_get_stats reports it)
        cur = nearC;
        continue;
      }
      if(hl || hr)
      {
        cur = hl ? ch.x : ch.y;
        continue;
      }
#endif
    }
#if PT_BVH_WIDTH != 2
    else if(TWO && ic.inst == BVH_NONE)
    {  // TLAS leaf: enter the instance (its BLAS root is an inner node)
      const TlasLeaf tl = S.tlasLeaves[cur & BVH_SLOT_MASK];
      ic    = InstCtx{tl.inst, sp, tl.wflags};
      if(tl.inst != PT_INST_MERGED)
        rbox = enter_instance(S, tl, o, d);
      nodes = S.wide;
      cur   = tl.nodeBase;
      continue;
    }
#endif
    else
    {
      const uint32_t slot  = cur & BVH_SLOT_MASK;
      TriRec         tr    = S.tris[slot];
      AlphaRec       ar;
      if(cur & BVH_ALPHA)  // non-opaque triangle: its any-hit inputs travel with the triangle (one round trip)
        ar = S.alphaRecs[slot];
#if PT_BVH_WIDTH != 2
      if(TWO && ic.inst != PT_INST_MERGED)
        tr = world_tri(S, ic, tr);
#endif
      const uint32_t wbits = __float_as_uint(tr.p0w.w);
      const uint32_t flags = wbits >> 29;
      const bool     opq   = (flags & TRI_OPAQUE) != 0;
      const bool     skip  = (MODE == TM_RAW_NONOPAQUE || MODE == TM_COUNT) && opq;
      if(!skip)
      {
#ifdef PT_STATS
        ++nTris;
#endif
        float t, u, v;
        // (TM_COUNT's upper key (tmax, wLimit) includes candidates that tie with the hit in t)
        if(tri_test(tr, MODE == TM_PICK ? (flags | TRI_NOCULL) : flags, o, d, t, u, v) && (MODE == TM_COUNT ? t <= tmax : t < tmax))
        {
          const uint32_t w = wbits & TRI_INDEX_MASK;
          if(MODE == TM_RAW_ALL || MODE == TM_RAW_NONOPAQUE || MODE == TM_PICK)
          {
            if(key_less(tPrev, wPrev, t, w) && (best.slot == BVH_NONE || key_less(t, w, best.t, best.w & TRI_INDEX_MASK)))
            {
              best.t = t; best.u = u; best.v = v; best.slot = slot; best.w = wbits;
            }
          }
          else if(MODE == TM_COUNT)
          {
            if(t > 0.0f && key_less(t, w, tmax, wLimit))
            {
              const float op = opacity_class(S, ar, u, v);
              if(op <= 0.0f)
                best.count++;
              else if(op < 1.0f)
                best.flags |= TF_SAW_FRAC;
              // (op >= 1 cannot occur in front of the nearest certain hit)
            }
          }
          else  // TM_CLOSEST / TM_SHADOW
          {
            if(MODE == TM_SHADOW && opq)
            {
              if(t > 0.0f)
              {
                opaqueHit = true;
                break;
              }
            }
            else if(t > 0.0f && (best.slot == BVH_NONE || key_less(t, w, best.t, best.w & TRI_INDEX_MASK)))
            {
              bool certain = opq;
              if(!opq)
              {
                const float op = opacity_class(S, ar, u, v);
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
      }
    }
    // pop
#if PT_BVH_WIDTH != 2
    if(TWO && ic.inst != BVH_NONE && sp == ic.spBase)
    {  // the instance's subtree is exhausted: back to TLAS level
      ic.inst = BVH_NONE;
      rbox    = rboxW;
      nodes   = S.tlas;
    }
#endif
    if(sp == 0)
      break;
    --sp;
    cur = sp < STACK_LDS ? ldsStack[sp * TRACE_BLOCK] : spill[sp - STACK_LDS];
  }
#undef PT_TLIMIT
#ifdef PT_HIST
  {
    atomicAdd(&g_hist[MODE][nIter ? 32 - __clz(nIter) : 0], 1ull);
    unsigned long long m = __ballot(1);
    const int          first = __ffsll(m) - 1;
    uint32_t           wmax = 0, wsum = 0, wn = 0;
    while(m)
    {
      const int      l = __ffsll(m) - 1;
      const uint32_t x = __builtin_amdgcn_readlane(nIter, l);
      wmax = x > wmax ? x : wmax;
      wsum += x;
      ++wn;
      m &= m - 1;
    }
    if(int(threadIdx.x & 63) == first)
    {
      atomicAdd(&g_hist[MODE][32], (unsigned long long)wsum);
      atomicAdd(&g_hist[MODE][33], (unsigned long long)wmax * 64ull);
      atomicAdd(&g_hist[MODE][34], (unsigned long long)wn);
      atomicAdd(&g_hist[MODE][35], 1ull);
      atomicAdd(&g_hist[MODE][36], (unsigned long long)wmax);
    }
  }
#endif
#ifdef PT_STATS
  if(PT_STATS == 0 || (PT_STATS == 1 && MODE == TM_SHADOW) || (PT_STATS == 2 && MODE == TM_CLOSEST) || (PT_STATS == 3 && MODE == TM_COUNT))
  {
    atomicAdd(&counters->nodesVisited, (unsigned long long)nNodes);
    atomicAdd(&counters->trisTested, (unsigned long long)nTris);
  }
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
