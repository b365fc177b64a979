// CPU harness around the PRODUCT's traversal source (test infrastructure; built and used by tests/test_trace_host.py only).
//
// vk_raytrace_amd/csrc/pt_trace.h -- traverse<MODE, TWO>, wide_node_step, make_raybox, enter_instance, world_tri, tri_test -- is plain
// inline C++ apart from a handful of intrinsics, so it is compiled here for the host (g++, -ffp-contract=off like the device build) and run
// against a brute-force loop over every world triangle with the same tri_test.  What the GPU parity tests can only show through images is
// checked ray by ray without a GPU: the flat walk and the two-level walk (TLAS + object-space BLASes, per-instance box padding from
// pt_capi.hip's two_level_pad) must report exactly the candidates brute force reports -- every candidate along the ray, in key order.
//
// The acceleration structures are assembled on the host in the product's formats (TriRec, WideNode, TlasLeaf).  Topology comes from the
// product's device builder run through its host emulation (pt_debug_sahdev_topology in libptmi.so); boxes, the 4-wide collapse, the vertex
// form of BLAS leaves and the TLAS proxies restate pt_accel.hip (k_gather's tri_box, k_collapse, k_blas_vertex_form, k_instance_proxies)
// -- they only have to be valid structures of that format, the code under test is the walk.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime.h>  // vector types; nothing is launched
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <atomic>

// the device intrinsics pt_trace.h and the headers it includes use
static inline unsigned int __float_as_uint(float f) { unsigned int u; std::memcpy(&u, &f, 4); return u; }
static inline float        __uint_as_float(unsigned int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline int          __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float        __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
template <class T>
static inline T atomicAdd(T* p, T v) { T o = *p; *p += v; return o; }
// a one-lane "wavefront" for the wave-level helpers of pt_machine.h (the ray supply is not used here; the per-lane state machine is)
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
static inline int                __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline unsigned int       __builtin_amdgcn_readfirstlane(unsigned int x) { return x; }
static const struct { unsigned x, y, z; } threadIdx = {0, 0, 0};

#ifdef TH_ROBUST_T2
// EXPERIMENT (tools/t2_robust_experiment.py; not the contract): T2 with a forward error bound.  The fp32 evaluation is kept whenever its verdict
// cannot be an artefact of rounding: |det|, u, v, 1 - u - v and t are further from their decision boundaries than the rounding error of their
// numerators allows.  Otherwise the same formulas are evaluated in double precision (IEEE, so identical on every side) and rounded once.
#include <atomic>
#include "pt_device.h"
static std::atomic<unsigned long long> g_t2Calls{0}, g_t2Double{0};  // per mode of th_candidates: [brute force + walks], filled from thread-local counts
static thread_local unsigned long long tl_t2Calls = 0, tl_t2Double = 0;
extern "C" void th_t2_stats(unsigned long long* out2) { out2[0] = g_t2Calls.exchange(0); out2[1] = g_t2Double.exchange(0); }
static inline bool th_tri_test_robust(const TriRec& tr, uint32_t flags, f3 o, f3 d, float& t, float& u, float& v)
{
  const f3    e1 = xyz(tr.e1n), e2 = xyz(tr.e2p), p0 = xyz(tr.p0w);
  const f3    pv = cross3(d, e2);
  const float det = dot3(e1, pv);
  const f3    tv = o - p0;
  const f3    qv = cross3(tv, e1);
  const float nu = dot3(tv, pv), nv = dot3(d, qv), nt = dot3(e2, qv);
  // forward error bounds of det and of the three numerators, componentwise: a cross product's component a_i b_j - a_j b_i is off by at most
  // 2 ulp of |a_i b_j| + |a_j b_i|, a 3-term dot by 3 ulp of sum |x_i y_i| plus |x| . (error of y); 8 ulp covers every chain below
  const f3    apv = f3{fabsf(d.y) * fabsf(e2.z) + fabsf(d.z) * fabsf(e2.y), fabsf(d.z) * fabsf(e2.x) + fabsf(d.x) * fabsf(e2.z), fabsf(d.x) * fabsf(e2.y) + fabsf(d.y) * fabsf(e2.x)};
  const f3    atv = f3{fabsf(tv.x), fabsf(tv.y), fabsf(tv.z)};
  const f3    aqv = f3{atv.y * fabsf(e1.z) + atv.z * fabsf(e1.y), atv.z * fabsf(e1.x) + atv.x * fabsf(e1.z), atv.x * fabsf(e1.y) + atv.y * fabsf(e1.x)};
  const float k   = 8.0f * 5.9604645e-8f;
  const float edet = k * (fabsf(e1.x) * apv.x + fabsf(e1.y) * apv.y + fabsf(e1.z) * apv.z);
  const float eu   = k * (atv.x * apv.x + atv.y * apv.y + atv.z * apv.z);
  const float ev   = k * (fabsf(d.x) * aqv.x + fabsf(d.y) * aqv.y + fabsf(d.z) * aqv.z);
  const float et   = k * (fabsf(e2.x) * aqv.x + fabsf(e2.y) * aqv.y + fabsf(e2.z) * aqv.z);
  const float adet = fabsf(det);
  ++tl_t2Calls;
  bool        sure = adet > 4.0f * edet;
  if(sure)
  {
    const float s  = det < 0.0f ? -1.0f : 1.0f;
    const float su = nu * s, sv = nv * s;  // compare numerators against 0 and |det| (no division needed for the verdict)
    const bool  inside  = su > eu && sv > ev && (adet - su - sv) > (eu + ev + edet);
    const bool  outside = su < -eu || sv < -ev || (su + sv - adet) > (eu + ev + edet);
    const bool  tOk     = et <= 4.0e-6f * fabsf(nt);  // relative error of t below the slack of the box tests (leaf padding 4e-6 |coordinate|)
    sure = outside || (inside && tOk);
  }
  if(sure)
  {
    if(det == 0.0f)
      return false;
    if(!(flags & TRI_NOCULL))
    {
      const bool front = (flags & TRI_FLIP) ? (det < 0.0f) : (det > 0.0f);
      if(!front)
        return false;
    }
    const float inv = 1.0f / det;
    u = nu * inv;
    if(u < 0.0f || u > 1.0f)
      return false;
    v = nv * inv;
    if(v < 0.0f || u + v > 1.0f)
      return false;
    t = nt * inv;
    return true;
  }
  // ambiguous in fp32: the same test in double
  ++tl_t2Double;
  const double E1[3] = {e1.x, e1.y, e1.z}, E2[3] = {e2.x, e2.y, e2.z}, D[3] = {d.x, d.y, d.z}, TV[3] = {double(o.x) - p0.x, double(o.y) - p0.y, double(o.z) - p0.z};
  const double PV[3] = {D[1] * E2[2] - D[2] * E2[1], D[2] * E2[0] - D[0] * E2[2], D[0] * E2[1] - D[1] * E2[0]};
  const double DET   = E1[0] * PV[0] + E1[1] * PV[1] + E1[2] * PV[2];
  if(DET == 0.0)
    return false;
  if(!(flags & TRI_NOCULL))
  {
    const bool front = (flags & TRI_FLIP) ? (DET < 0.0) : (DET > 0.0);
    if(!front)
      return false;
  }
  const double U = (TV[0] * PV[0] + TV[1] * PV[1] + TV[2] * PV[2]) / DET;
  if(U < 0.0 || U > 1.0)
    return false;
  const double QV[3] = {TV[1] * E1[2] - TV[2] * E1[1], TV[2] * E1[0] - TV[0] * E1[2], TV[0] * E1[1] - TV[1] * E1[0]};
  const double V     = (D[0] * QV[0] + D[1] * QV[1] + D[2] * QV[2]) / DET;
  if(V < 0.0 || U + V > 1.0)
    return false;
  u = float(U);
  v = float(V);
  t = float((E2[0] * QV[0] + E2[1] * QV[1] + E2[2] * QV[2]) / DET);
  return true;
}
#define PT_TRI_TEST_OVERRIDE th_tri_test_robust
#endif
#ifdef TH_CERTIFIED_T2
// EXPERIMENT (tools/t2_robust_experiment.py; not the contract): "certified" T2 -- the contract's fp32 Moeller-Trumbore, whose ACCEPTED candidates are kept
// only when the forward error bound of round 2's experiment certifies their barycentrics to TH_TAU (and their distance to TH_TAU relative): no fp64, no
// second code path for a wavefront to diverge into, ~35 more fp32 operations per test.  A candidate that fp32 cannot certify counts as a miss ON EVERY SIDE
// (brute force, every walk), so what is left of "BVH-dependent" is a hit that lies up to TH_TAU of its triangle's extent outside the triangle's box.
#include <atomic>
#include "pt_device.h"
#ifndef TH_TAU
#define TH_TAU 0.0078125f  // 2^-7
#endif
static std::atomic<unsigned long long> g_t2Calls{0}, g_t2Double{0};  // [tests, fp32 accepts that certification turned into misses]
static thread_local unsigned long long tl_t2Calls = 0, tl_t2Double = 0;
extern "C" void th_t2_stats(unsigned long long* out2) { out2[0] = g_t2Calls.exchange(0); out2[1] = g_t2Double.exchange(0); }
static std::atomic<unsigned long long> g_t2Accepts{0};
static thread_local unsigned long long tl_t2Accepts = 0;
extern "C" unsigned long long th_t2_accepts() { return g_t2Accepts.exchange(0); }
static inline bool th_tri_test_certified(const TriRec& tr, uint32_t flags, f3 o, f3 d, float& t, float& u, float& v)
{
  const f3    e1 = xyz(tr.e1n), e2 = xyz(tr.e2p), p0 = xyz(tr.p0w);
  const f3    pv = cross3(d, e2);
  const float det = dot3(e1, pv);
  ++tl_t2Calls;
  if(det == 0.0f)
    return false;
  if(!(flags & TRI_NOCULL))
  {
    const bool front = (flags & TRI_FLIP) ? (det < 0.0f) : (det > 0.0f);
    if(!front)
      return false;
  }
  const float inv = 1.0f / det;
  const f3    tv  = o - p0;
  const float nu  = dot3(tv, pv);
  u               = nu * inv;
  if(u < 0.0f || u > 1.0f)
    return false;
  const f3    qv = cross3(tv, e1);
  const float nv = dot3(d, qv);
  v              = nv * inv;
  if(v < 0.0f || u + v > 1.0f)
    return false;
  const float nt = dot3(e2, qv);
  t              = nt * inv;
  // certification of the accepted candidate
  const f3    apv = f3{fabsf(d.y) * fabsf(e2.z) + fabsf(d.z) * fabsf(e2.y), fabsf(d.z) * fabsf(e2.x) + fabsf(d.x) * fabsf(e2.z), fabsf(d.x) * fabsf(e2.y) + fabsf(d.y) * fabsf(e2.x)};
  const f3    atv = f3{fabsf(tv.x), fabsf(tv.y), fabsf(tv.z)};
  const f3    aqv = f3{atv.y * fabsf(e1.z) + atv.z * fabsf(e1.y), atv.z * fabsf(e1.x) + atv.x * fabsf(e1.z), atv.x * fabsf(e1.y) + atv.y * fabsf(e1.x)};
  const float k   = 8.0f * 5.9604645e-8f;
  const float edet = k * (fabsf(e1.x) * apv.x + fabsf(e1.y) * apv.y + fabsf(e1.z) * apv.z);
  const float eu   = k * (atv.x * apv.x + atv.y * apv.y + atv.z * apv.z);
  const float ev   = k * (fabsf(d.x) * aqv.x + fabsf(d.y) * aqv.y + fabsf(d.z) * aqv.z);
  const float et   = k * (fabsf(e2.x) * aqv.x + fabsf(e2.y) * aqv.y + fabsf(e2.z) * aqv.z);
  const float adet = fabsf(det);
  ++tl_t2Accepts;
  const bool ok = (eu + ev + 2.0f * edet) <= TH_TAU * adet && (et * adet + fabsf(nt) * edet) <= TH_TAU * fabsf(nt) * adet;
  if(!ok)
    ++tl_t2Double;
  return ok;
}
#define PT_TRI_TEST_OVERRIDE th_tri_test_certified
#endif
#include "pt_shade.h"  // pt_settle.h (pt_trace.h + the per-ray settle functions k_tail runs) + the shading steps of a path (generate_ray, shade_path, ...)
#include "pt_machine.h"  // the resumable per-lane traversal of the persistent kernels (k_closest_p / k_shadow_p)
#include "pt_cnode.h"    // WideNode -> CompactNode (what pt_accel.hip k_compact_nodes runs per node)
#include "../../include/pt_types.h"

extern "C" int pt_debug_sahdev_topology(uint32_t n, const float* tri9, uint32_t* vals, uint32_t* childL, uint32_t* childR, uint32_t* parI, uint32_t* parL);
extern "C" int pt_debug_two_level_pad(const float* worldMatrix16, float Bo, float* out27);
extern "C" int pt_debug_mat_lines(const pt_SceneDesc* d, void* linesOut, char* err, size_t errLen);
extern "C" int pt_debug_scene_records(const pt_SceneDesc* d, unsigned long long* counts5, void* instOut, float* padOut, void* alphaMatsOut, uint32_t* alphaMapsOut, uint32_t* texelsOut,
                                      void* texRecsOut, char* err, size_t errLen);
extern "C" int pt_build_env_accel(const float* rgba32f, int width, int height, pt_EnvAccel* out, float* out_integral, float* out_average);

namespace {

struct Box {
  float lo[3], hi[3];
  bool  alpha;
};
struct Bvh {
  std::vector<TriRec>   tris;   // leaf order
  std::vector<WideNode> wide;   // node 0 = root
  float                 lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  // the binary tree the wide nodes were collapsed from (th_step_model re-collapses it at other widths)
  std::vector<uint32_t> cl, cr;
  std::vector<Box>      leafBox, innerBox;
};

// pt_accel.hip tri_box: padded box of a record (p0, p0 + e1, p0 + e2)
void tri_box_h(const TriRec& r, float lo[3], float hi[3])
{
  const float p0[3] = {r.p0w.x, r.p0w.y, r.p0w.z};
  const float p1[3] = {r.p0w.x + r.e1n.x, r.p0w.y + r.e1n.y, r.p0w.z + r.e1n.z};
  const float p2[3] = {r.p0w.x + r.e2p.x, r.p0w.y + r.e2p.y, r.p0w.z + r.e2p.z};
  for(int a = 0; a < 3; ++a)
  {
    lo[a] = std::fmin(p0[a], std::fmin(p1[a], p2[a]));
    hi[a] = std::fmax(p0[a], std::fmax(p1[a], p2[a]));
    const float m = std::fmax(std::fabs(lo[a]), std::fabs(hi[a])), pad = m * 4e-6f + 1e-30f;
    lo[a] -= pad;
    hi[a] += pad;
  }
}

float half_area_h(const Box& b)
{
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return dx * dy + dy * dz + dz * dx;
}

// records (edge form, flags in p0w.w >> 29) -> leaf order + 4-wide nodes (k_gather, k_refit, k_emit, k_collapse)
Bvh build_bvh(const std::vector<TriRec>& in)
{
  Bvh            out;
  const uint32_t n = uint32_t(in.size());
  if(n == 0)
    return out;
  std::vector<uint32_t> vals(n), cl(n), cr(n), pi(n), pl(n);
  if(n >= 2)
  {
    std::vector<float> tri9(size_t(n) * 9);
    for(uint32_t i = 0; i < n; ++i)
    {
      const float v[9] = {in[i].p0w.x, in[i].p0w.y, in[i].p0w.z, in[i].e1n.x, in[i].e1n.y, in[i].e1n.z, in[i].e2p.x, in[i].e2p.y, in[i].e2p.z};
      std::memcpy(&tri9[size_t(i) * 9], v, sizeof(v));
    }
    if(pt_debug_sahdev_topology(n, tri9.data(), vals.data(), cl.data(), cr.data(), pi.data(), pl.data()) != 0)
      return out;
  }
  else
    vals[0] = 0;
  out.tris.resize(n);
  std::vector<Box> leaf(n), inner(n > 1 ? n - 1 : 1);
  for(uint32_t i = 0; i < n; ++i)
  {
    out.tris[i] = in[vals[i]];
    tri_box_h(out.tris[i], leaf[i].lo, leaf[i].hi);
    leaf[i].alpha = !((__float_as_uint(out.tris[i].p0w.w) >> 29) & TRI_OPAQUE);
  }
  auto ref_box = [&](uint32_t r) -> const Box& { return (r & BVH_LEAF) ? leaf[r & ~BVH_LEAF] : inner[r]; };
  if(n >= 2)
  {  // boxes bottom-up: a subtree over k leaves owns k-1 consecutive ids starting at its root, so children have larger ids than parents
    for(uint32_t k = n - 1; k-- > 0;)
    {
      const Box &a = ref_box(cl[k]), &b = ref_box(cr[k]);
      for(int x = 0; x < 3; ++x)
      {
        inner[k].lo[x] = std::fmin(a.lo[x], b.lo[x]);
        inner[k].hi[x] = std::fmax(a.hi[x], b.hi[x]);
      }
      inner[k].alpha = a.alpha || b.alpha;
    }
  }
  auto child_ref = [&](uint32_t r) -> uint32_t {
    if(r & BVH_LEAF)
      return BVH_LEAF | (r & ~BVH_LEAF) | (leaf[r & ~BVH_LEAF].alpha ? BVH_ALPHA : 0u);
    return r;
  };
  // collapse (k_collapse): per wide node, open the inner child of largest area until 4 children
  struct Item { uint32_t b2, wide; };
  std::vector<Item> queue{{0u, 0u}};
  out.wide.resize(1);
  for(size_t qi = 0; qi < queue.size(); ++qi)
  {
    const Item it = queue[qi];
    uint32_t   id[4];
    int        cnt = 0;
    if(n == 1)
      id[cnt++] = BVH_LEAF | 0u;
    else
    {
      id[cnt++] = cl[it.b2];
      id[cnt++] = cr[it.b2];
      while(cnt < 4)
      {
        int   best = -1;
        float bestA = -1.f;
        for(int k = 0; k < cnt; ++k)
          if(!(id[k] & BVH_LEAF))
          {
            const float a = half_area_h(inner[id[k]]);
            if(a > bestA)
            {
              bestA = a;
              best  = k;
            }
          }
        if(best < 0)
          break;
        const uint32_t node = id[best];
        id[best]            = id[cnt - 1];
        --cnt;
        id[cnt++] = cl[node];
        id[cnt++] = cr[node];
      }
    }
    WideNode w;
    std::memset(&w, 0, sizeof(w));
    float*    mnx = &w.minx[0].x; float* mny = &w.miny[0].x; float* mnz = &w.minz[0].x;
    float*    mxx = &w.maxx[0].x; float* mxy = &w.maxy[0].x; float* mxz = &w.maxz[0].x;
    uint32_t* ch  = &w.child[0].x;
    for(int k = 0; k < 4; ++k)
    {
      if(k < cnt)
      {
        const Box& b = ref_box(id[k]);
        mnx[k] = b.lo[0]; mny[k] = b.lo[1]; mnz[k] = b.lo[2]; mxx[k] = b.hi[0]; mxy[k] = b.hi[1]; mxz[k] = b.hi[2];
        if(id[k] & BVH_LEAF)
          ch[k] = child_ref(id[k]);
        else
        {
          const uint32_t wid = uint32_t(out.wide.size());
          out.wide.emplace_back();
          queue.push_back({id[k], wid});
          ch[k] = wid | (inner[id[k]].alpha ? BVH_ALPHA : 0u);
        }
      }
      else
      {
        mnx[k] = mny[k] = mnz[k] = FLT_MAX;
        mxx[k] = mxy[k] = mxz[k] = -FLT_MAX;
        ch[k]                    = BVH_NONE;
      }
    }
    out.wide[it.wide] = w;
  }
  const Box& root = n >= 2 ? inner[0] : leaf[0];
  for(int a = 0; a < 3; ++a)
  {
    out.lo[a] = root.lo[a];
    out.hi[a] = root.hi[a];
  }
  out.cl = cl; out.cr = cr; out.leafBox = leaf; out.innerBox = inner;
  return out;
}

struct InstIn {
  uint32_t vertexOffset, firstIndex, triCount, flags;  // flags: TRI_OPAQUE / TRI_NOCULL (TRI_FLIP is derived from the matrix)
  int32_t  primMesh;
  float    worldMatrix[16];  // column-major
};

struct Scene {
  std::vector<float4>      vertices;  // 2 x float4 per vertex
  std::vector<uint32_t>    indices;
  std::vector<InstanceRec> inst;
  std::vector<uint32_t>    instTriBase;
  std::vector<TriRec>      world;     // world index order (brute force)
  Bvh                      flat;
  // two-level
  std::vector<TriRec>      blasTris;
  std::vector<AlphaRec>    blasAlpha;
  std::vector<WideNode>    blasWide;
  Bvh                      tlas;
  std::vector<TlasLeaf>    tlasLeaves;
  std::vector<AlphaRec>    flatAlpha;
  std::vector<uint32_t>    instBlock;  // DeviceScene::instBlock (pt_capi.hip build_tlas)
  std::vector<CompactNode> blasCNodes, tlasCNodes;  // ... and the two-level structure's
  std::vector<CompactNode> flatCNodes;  // PT_TUNE cnodes=1: the flat structure's nodes in the compact form (read by lane_inner only)
  AlphaMat                 alphaMat;
  std::vector<AlphaMat>    alphaMats;   // th_create_scene: the product's own records (pt_debug_scene_records)
  std::vector<uint32_t>    alphaMaps, texels;
  std::vector<TexRec>      texRecs;
  std::vector<uint4>       matLines;  // per material its 128-byte line (pt_device.h mat_line_pack)
  std::vector<pt_GltfShadeMaterial> materials;
  std::vector<pt_Light>    lights;
  std::vector<float4>      env;
  std::vector<pt_EnvAccel> envAccel;
  DeviceScene              dsFlat, dsTwo;
  double                   maxPadRatio = 0;
};

f3 vpos(const Scene& s, uint32_t v) { const float4 a = s.vertices[size_t(v) * 2]; return f3{a.x, a.y, a.z}; }

// k_world_tris (trace contract T1)
TriRec world_record(const Scene& s, const InstanceRec& I, uint32_t inst, uint32_t k, uint32_t w)
{
  const uint32_t* t  = &s.indices[I.firstIndex + 3 * size_t(k)];
  const f3        p0 = xform_point(I.objectToWorld, vpos(s, I.vertexOffset + t[0])), p1 = xform_point(I.objectToWorld, vpos(s, I.vertexOffset + t[1])),
           p2 = xform_point(I.objectToWorld, vpos(s, I.vertexOffset + t[2]));
  const f3 e1 = p1 - p0, e2 = p2 - p0;
  TriRec   r;
  r.p0w = make_float4(p0.x, p0.y, p0.z, __uint_as_float(w | (I.flags << 29)));
  r.e1n = make_float4(e1.x, e1.y, e1.z, __uint_as_float(inst));
  r.e2p = make_float4(e2.x, e2.y, e2.z, __uint_as_float(k));
  return r;
}

}  // namespace

extern "C" {

// any-hit inputs of triangle k of instance I (k_world_tris): raw texcoords of the three vertices + material
static AlphaRec alpha_record(const Scene& s, const InstanceRec& I, uint32_t k)
{
  const uint32_t* t = &s.indices[I.firstIndex + 3 * size_t(k)];
  const float4    b0 = s.vertices[size_t(I.vertexOffset + t[0]) * 2 + 1], b1 = s.vertices[size_t(I.vertexOffset + t[1]) * 2 + 1], b2 = s.vertices[size_t(I.vertexOffset + t[2]) * 2 + 1];
  AlphaRec        ar;
  ar.uv0[0] = b0.x; ar.uv0[1] = b0.y; ar.uv1[0] = b1.x; ar.uv1[1] = b1.y; ar.uv2[0] = b2.x; ar.uv2[1] = b2.y;
  ar.material = uint32_t(I.materialIndex < 0 ? 0 : I.materialIndex);
  ar._pad     = 0;
  return ar;
}

static std::atomic<unsigned long long> g_innerSteps{0};  // node steps of the machine walks since the last th_take_inner_steps()
extern "C" unsigned long long th_take_inner_steps() { return g_innerSteps.exchange(0); }
static std::atomic<unsigned long long> g_leafSteps{0};  // triangle steps, same bracket
extern "C" unsigned long long th_take_leaf_steps() { return g_leafSteps.exchange(0); }
static std::atomic<unsigned long long> g_spHist[65];  // machine walks: rays by the deepest traversal-stack level they used (tools/stack_depth_experiment.py)
extern "C" void th_take_sp_hist(unsigned long long* out65) { for(int i = 0; i < 65; ++i) out65[i] = g_spHist[i].exchange(0); }
static int g_compactNodes = 0, g_compactOk = 0;  // PT_TUNE cnodes (pt_internal.h)
extern "C" void th_set_compact_nodes(int on) { g_compactNodes = on; }
extern "C" int  th_compact_ok() { return g_compactOk; }

static int g_mergeSingles = 1;  // PT_TUNE mergeSingles (pt_internal.h): the product's default
extern "C" void th_set_merge_singles(int on) { g_mergeSingles = on; }

// flat + two-level structures over s->inst (already filled), pads per instance
static void build_structures(Scene* s, const std::vector<float>& padC0, const std::vector<float>& padC1, uint32_t numPrimMeshes)
{
  const uint32_t numInst  = uint32_t(s->inst.size());
  uint32_t       triTotal = 0;
  s->instTriBase.assign(numInst ? numInst : 1, 0u);
  for(uint32_t i = 0; i < numInst; ++i)
  {
    s->instTriBase[i] = s->inst[i].triBase;
    triTotal += s->inst[i].triCount;
  }
  // ---- flat: world records of every instance, one hierarchy
  s->world.reserve(triTotal);
  for(uint32_t i = 0; i < numInst; ++i)
    for(uint32_t k = 0; k < s->inst[i].triCount; ++k)
      s->world.push_back(world_record(*s, s->inst[i], i, k, s->inst[i].triBase + k));
  s->flat = build_bvh(s->world);
  s->flatAlpha.assign(std::max<size_t>(1, s->flat.tris.size()), AlphaRec{});
  for(size_t i = 0; i < s->flat.tris.size(); ++i)
    s->flatAlpha[i] = alpha_record(*s, s->inst[__float_as_uint(s->flat.tris[i].e1n.w)], __float_as_uint(s->flat.tris[i].e2p.w));
  // ---- two-level: the prim-meshes instantiated once share one world-space structure at slot 0 / node 0 (pt_capi.hip build_merged /
  // pt_accel.hip pt_merged_build) ...
  std::vector<char> isMerged(numInst, 0);
  bool              haveMerged = false;
  float             mlo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mhi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if(g_mergeSingles)
  {
    std::vector<uint32_t> uses(numPrimMeshes, 0);
    for(uint32_t i = 0; i < numInst; ++i)
      if(s->inst[i].triCount)
        uses[s->inst[i].primMesh]++;
    std::vector<TriRec> mw;
    for(uint32_t i = 0; i < numInst; ++i)
      if(s->inst[i].triCount && uses[s->inst[i].primMesh] == 1)
      {
        isMerged[i] = 1;
        for(uint32_t k = 0; k < s->inst[i].triCount; ++k)
          mw.push_back(world_record(*s, s->inst[i], i, k, s->inst[i].triBase + k));
      }
    if(!mw.empty())
    {
      haveMerged = true;
      Bvh b      = build_bvh(mw);
      for(const TriRec& r : b.tris)
      {
        s->blasTris.push_back(r);
        s->blasAlpha.push_back(alpha_record(*s, s->inst[__float_as_uint(r.e1n.w)], __float_as_uint(r.e2p.w)));
        float lo[3], hi[3];
        tri_box_h(r, lo, hi);  // the root box of the structure is the union of its padded leaf boxes
        for(int a = 0; a < 3; ++a)
        {
          mlo[a] = std::fmin(mlo[a], lo[a]);
          mhi[a] = std::fmax(mhi[a], hi[a]);
        }
      }
      for(const WideNode& w : b.wide)
        s->blasWide.push_back(w);  // slot base and node base are 0: the references are already global
    }
  }
  // ... and one object-space BLAS per other prim-mesh that is instantiated (pt_capi.hip build_two_level / pt_accel.hip pt_blas_build)
  std::vector<int64_t> nodeBaseOf(numPrimMeshes, -1);
  for(uint32_t i = 0; i < numInst; ++i)
  {
    const InstanceRec& I = s->inst[i];
    if(I.triCount == 0 || isMerged[i] || nodeBaseOf[I.primMesh] >= 0)
      continue;
    InstanceRec P = I;  // the pseudo-instance: identity transform, no TRI_FLIP
    P.objectToWorld.r0 = make_float4(1, 0, 0, 0); P.objectToWorld.r1 = make_float4(0, 1, 0, 0); P.objectToWorld.r2 = make_float4(0, 0, 1, 0);
    P.flags &= ~TRI_FLIP;
    std::vector<TriRec> obj(I.triCount);
    for(uint32_t k = 0; k < I.triCount; ++k)
      obj[k] = world_record(*s, P, 0, k, k);
    Bvh            b        = build_bvh(obj);
    const uint32_t nodeBase = uint32_t(s->blasWide.size()), slotBase = uint32_t(s->blasTris.size());
    for(TriRec r : b.tris)
    {  // vertex form (k_blas_vertex_form)
      const uint32_t  k = __float_as_uint(r.e2p.w);
      const uint32_t* t = &s->indices[I.firstIndex + 3 * size_t(k)];
      const f3        v0 = vpos(*s, I.vertexOffset + t[0]), v1 = vpos(*s, I.vertexOffset + t[1]), v2 = vpos(*s, I.vertexOffset + t[2]);
      r.p0w = make_float4(v0.x, v0.y, v0.z, __uint_as_float(k));
      r.e1n = make_float4(v1.x, v1.y, v1.z, 0.f);
      r.e2p = make_float4(v2.x, v2.y, v2.z, 0.f);
      s->blasTris.push_back(r);
      s->blasAlpha.push_back(alpha_record(*s, I, k));
    }
    for(WideNode w : b.wide)
    {  // global references (k_blas_rebase)
      uint32_t* ch = &w.child[0].x;
      for(int k = 0; k < 4; ++k)
        if(ch[k] != BVH_NONE)
          ch[k] = (ch[k] & ~BVH_SLOT_MASK) | ((ch[k] & BVH_SLOT_MASK) + ((ch[k] & BVH_LEAF) ? slotBase : nodeBase));
      s->blasWide.push_back(w);
    }
    nodeBaseOf[I.primMesh] = nodeBase;
  }
  if(s->blasAlpha.empty())
    s->blasAlpha.emplace_back();
  // TLAS over the exact world boxes of the instances (k_instance_proxies), as "diagonal" records
  std::vector<TriRec> prox;
  for(uint32_t i = 0; i < numInst; ++i)
  {
    const InstanceRec& I = s->inst[i];
    if(I.triCount == 0 || isMerged[i])
      continue;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for(uint32_t j = 0; j < 3 * I.triCount; ++j)
    {
      const f3    p    = xform_point(I.objectToWorld, vpos(*s, I.vertexOffset + s->indices[I.firstIndex + j]));
      const float q[3] = {p.x, p.y, p.z};
      for(int a = 0; a < 3; ++a)
      {
        lo[a] = std::fmin(lo[a], q[a]);
        hi[a] = std::fmax(hi[a], q[a]);
      }
    }
    TriRec r;
    r.p0w = make_float4(lo[0], lo[1], lo[2], __uint_as_float(i | (I.flags << 29)));
    r.e1n = make_float4(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 0.f);
    r.e2p = make_float4(0.f, 0.f, 0.f, 0.f);
    prox.push_back(r);
  }
  if(haveMerged)
  {  // the merged structure's proxy (pt_tlas_build): its root box
    TriRec r;
    r.p0w = make_float4(mlo[0], mlo[1], mlo[2], __uint_as_float(TRI_INDEX_MASK));
    r.e1n = make_float4(mhi[0] - mlo[0], mhi[1] - mlo[1], mhi[2] - mlo[2], 0.f);
    r.e2p = make_float4(0.f, 0.f, 0.f, 0.f);
    prox.push_back(r);
  }
  s->tlas = build_bvh(prox);
  for(const TriRec& r : s->tlas.tris)
  {
    const uint32_t id = __float_as_uint(r.p0w.w) & TRI_INDEX_MASK;
    TlasLeaf       l;
    std::memset(&l, 0, sizeof(l));
    if(id == TRI_INDEX_MASK)
    {
      l.inst = PT_INST_MERGED;
      s->tlasLeaves.push_back(l);
      continue;
    }
    l.inst     = id;
    l.nodeBase = uint32_t(nodeBaseOf[s->inst[id].primMesh]);
    l.wflags   = s->inst[id].triBase | (s->inst[id].flags << 29);
    l.padC0    = padC0[id];
    l.padC1    = padC1[id];
    s->tlasLeaves.push_back(l);
  }
  if(s->tlasLeaves.empty())
    s->tlasLeaves.emplace_back();
  // ---- the scene records the walk reads
  if(s->alphaMats.empty())
  {
    std::memset(&s->alphaMat, 0, sizeof(s->alphaMat));
    s->alphaMat.factorA = 1.0f; s->alphaMat.tex = -1; s->alphaMat.mapOffset = ALPHA_NO_MAP;
    s->alphaMats.push_back(s->alphaMat);
  }
  if(s->alphaMaps.empty())
    s->alphaMaps.push_back(0u);
  if(s->texels.empty())
    s->texels.push_back(0xffffffffu);
  DeviceScene d;
  std::memset(&d, 0, sizeof(d));
  d.vertices = s->vertices.data(); d.indices = s->indices.data(); d.instances = s->inst.data();
  d.alphaMats = s->alphaMats.data(); d.alphaMaps = s->alphaMaps.data(); d.texels = s->texels.data();
  d.materials = s->materials.empty() ? nullptr : s->materials.data(); d.lights = s->lights.empty() ? nullptr : s->lights.data();
  d.texRecs = s->texRecs.empty() ? nullptr : s->texRecs.data();
  d.matLines = s->matLines.empty() ? nullptr : s->matLines.data();
  d.numTris = triTotal; d.numInstances = numInst;
  s->dsFlat           = d;
  s->dsFlat.wide      = s->flat.wide.data();
  s->dsFlat.tris      = s->flat.tris.data();
  s->dsFlat.alphaRecs = s->flatAlpha.data();
  if(g_compactNodes)
  {
    s->flatCNodes.resize(s->flat.wide.size());
    bool ok = true;
    for(size_t i = 0; i < s->flat.wide.size(); ++i)
      ok = cn_encode(s->flat.wide[i], s->flatCNodes[i]) && ok;
    s->dsFlat.cnodes = ok ? s->flatCNodes.data() : nullptr;
    g_compactOk      = ok ? 1 : 0;
  }
  s->dsTwo            = d;
  s->dsTwo.wide        = s->blasWide.data();
  s->dsTwo.tris        = s->blasTris.data();
  s->dsTwo.alphaRecs   = s->blasAlpha.data();
  s->dsTwo.tlas        = s->tlas.wide.data();
  s->dsTwo.tlasLeaves  = s->tlasLeaves.data();
  s->dsTwo.instTriBase = s->instTriBase.data();
  s->dsTwo.twoLevel    = 1;
  {
    const size_t entries = (size_t(triTotal) >> PT_INST_BLOCK_SHIFT) + 2;
    s->instBlock.assign(entries, 0u);
    uint32_t at = 0;
    for(size_t e = 0; e < entries; ++e)
    {
      const uint64_t first = uint64_t(e) << PT_INST_BLOCK_SHIFT;
      while(at + 1 < s->instTriBase.size() && uint64_t(s->instTriBase[at + 1]) <= first)
        ++at;
      s->instBlock[e] = at;
    }
    s->dsTwo.instBlock = s->instBlock.data();
  }
  if(g_compactNodes && g_compactOk)
  {
    bool ok = true;
    s->blasCNodes.resize(s->blasWide.size());
    for(size_t i = 0; i < s->blasWide.size(); ++i)
      ok = cn_encode(s->blasWide[i], s->blasCNodes[i]) && ok;
    s->tlasCNodes.resize(s->tlas.wide.size());
    for(size_t i = 0; i < s->tlas.wide.size(); ++i)
      ok = cn_encode(s->tlas.wide[i], s->tlasCNodes[i]) && ok;
    if(ok && !s->blasCNodes.empty() && !s->tlasCNodes.empty())
    {
      s->dsTwo.cnodes = s->blasCNodes.data();
      s->dsTwo.ctlas  = s->tlasCNodes.data();
    }
    g_compactOk = ok ? 1 : 0;
  }
}

// Number of (node, child, axis, side) planes of the compact nodes that lie INSIDE the fp32 box they stand for, evaluated in double (must be 0: the
// decoded box has to enclose the original), plus the nodes whose child references differ; also reports how loose the boxes are.
extern "C" unsigned long long th_cnode_violations(void* p, double* meanExtraExtent)
{
  const Scene*       s     = static_cast<const Scene*>(p);
  unsigned long long bad   = 0;
  double             extra = 0.0;
  unsigned long long n     = 0;
  auto check = [&](const std::vector<WideNode>& wide, const std::vector<CompactNode>& cn) {
    for(size_t i = 0; i < wide.size() && i < cn.size(); ++i)
    {
      const WideNode&    w = wide[i];
      const CompactNode& c = cn[i];
      const float*    lo[3] = {&w.minx[0].x, &w.miny[0].x, &w.minz[0].x};
      const float*    hi[3] = {&w.maxx[0].x, &w.maxy[0].x, &w.maxz[0].x};
      const uint32_t* cc    = &w.child[0].x;
      const uint32_t* cd    = &c.child.x;
      const double    org[3] = {c.px, c.py, c.pz};
      for(int k = 0; k < 4; ++k)
      {
        if(cc[k] != cd[k])
          ++bad;
        if(cc[k] == BVH_NONE)
          continue;
        for(int a = 0; a < 3; ++a)
        {
          const double   step = std::ldexp(1.0, int((c.exps >> (8 * a)) & 0xffu) - 127);
          const uint32_t wl = (&c.ax[a].x)[k >> 1], wh = (&c.ax[a].x)[2 + (k >> 1)];
          const double   ql = cn_plane(wl, k & 1), qh = cn_plane(wh, k & 1);
          const double   dl = org[a] + ql * step, dh = org[a] + qh * step;
          if(dl > double(lo[a][k]) || dh < double(hi[a][k]))
            ++bad;
          const double ext = double(hi[a][k]) - double(lo[a][k]);
          extra += ((dh - dl) - ext) / (step * double(CN_GRID_MAX));  // growth of the child's extent in units of the node's grid extent
          ++n;
        }
      }
    }
  };
  check(s->flat.wide, s->flatCNodes);
  check(s->blasWide, s->blasCNodes);
  check(s->tlas.wide, s->tlasCNodes);
  if(meanExtraExtent)
    *meanExtraExtent = n ? extra / double(n) : 0.0;
  return bad;
}

// the same for the 64-byte form of the flat structure's nodes
extern "C" int th_compact_in_use(void* p, int two)  // 1: the walk of that structure reads compact nodes
{
  const Scene* s = static_cast<const Scene*>(p);
  return two ? (s->dsTwo.cnodes != nullptr && s->dsTwo.ctlas != nullptr) : (s->dsFlat.cnodes != nullptr ? 1 : 0);
}

// plain geometry + per-instance flags (every instance's material is the default: no any-hit evaluation is reachable with TRI_OPAQUE)
void* th_create(const float* vertices8, uint32_t numVerts, const uint32_t* indices, uint32_t numIdx, const InstIn* in, uint32_t numInst, const float* primBound, uint32_t numPrimMeshes)
{
  Scene* s = new Scene();
  s->vertices.resize(size_t(numVerts) * 2);
  std::memcpy(s->vertices.data(), vertices8, sizeof(float) * 8 * size_t(numVerts));
  s->indices.assign(indices, indices + numIdx);
  s->inst.resize(numInst);
  std::vector<float> padC0(numInst), padC1(numInst);
  uint32_t           triTotal = 0;
  for(uint32_t i = 0; i < numInst; ++i)
  {
    InstanceRec& I = s->inst[i];
    std::memset(&I, 0, sizeof(I));
    float rec[27];
    if(in[i].primMesh < 0 || uint32_t(in[i].primMesh) >= numPrimMeshes || pt_debug_two_level_pad(in[i].worldMatrix, primBound[in[i].primMesh], rec) != 0)
    {
      delete s;
      return nullptr;
    }
    I.objectToWorld.r0 = make_float4(rec[0], rec[1], rec[2], rec[3]);
    I.objectToWorld.r1 = make_float4(rec[4], rec[5], rec[6], rec[7]);
    I.objectToWorld.r2 = make_float4(rec[8], rec[9], rec[10], rec[11]);
    I.worldToObject.r0 = make_float4(rec[12], rec[13], rec[14], rec[15]);
    I.worldToObject.r1 = make_float4(rec[16], rec[17], rec[18], rec[19]);
    I.worldToObject.r2 = make_float4(rec[20], rec[21], rec[22], rec[23]);
    padC0[i]           = rec[24];
    padC1[i]           = rec[25];
    I.vertexOffset = in[i].vertexOffset; I.firstIndex = in[i].firstIndex; I.materialIndex = 0; I.primMesh = in[i].primMesh;
    I.triBase = triTotal; I.triCount = in[i].triCount;
    I.flags   = (in[i].flags & (TRI_OPAQUE | TRI_NOCULL)) | (uint32_t(rec[26]) & TRI_FLIP);
    triTotal += I.triCount;
  }
  build_structures(s, padC0, padC1, numPrimMeshes);
  return s;
}

// a full scene description (materials, textures): instance records, flags, alpha view, opacity maps and texel pool are the PRODUCT's
// (pt_capi.hip build_scene_records through pt_debug_scene_records), so the any-hit evaluation reads exactly what the GPU reads
void* th_create_scene(const pt_SceneDesc* d, char* err, size_t errLen)
{
  unsigned long long counts[5] = {0, 0, 0, 0, 0};
  if(pt_debug_scene_records(d, counts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, err, errLen) != 0)
    return nullptr;
  Scene* s = new Scene();
  s->vertices.resize(size_t(d->numVertices) * 2);
  std::memcpy(s->vertices.data(), d->vertices, sizeof(float) * 8 * size_t(d->numVertices));
  s->indices.assign(d->indices, d->indices + d->numIndices);
  s->inst.resize(counts[0]);
  s->alphaMats.resize(counts[1]);
  s->alphaMaps.resize(counts[2]);
  s->texels.resize(counts[3]);
  s->texRecs.resize(d->numTextures ? d->numTextures : 1);
  s->materials.assign(d->materials, d->materials + d->numMaterials);
  s->lights.assign(d->lights, d->lights + d->numLights);
  if(s->lights.empty())
    s->lights.emplace_back();
  std::vector<float> pad(2 * counts[0] + 2);
  if(pt_debug_scene_records(d, counts, s->inst.data(), pad.data(), s->alphaMats.data(), s->alphaMaps.data(), s->texels.data(), s->texRecs.data(), err, errLen) != 0)
  {
    delete s;
    return nullptr;
  }
  s->matLines.assign(size_t(PT_MAT_LINE_QUADS) * std::max<size_t>(1, s->materials.size()), uint4{0u, 0u, 0u, 0u});
  if(pt_debug_mat_lines(d, s->matLines.data(), err, errLen) != 0)  // the product's own lines: their descriptors point into the pool fetched above (plain copies + interleaved groups)
  {
    delete s;
    return nullptr;
  }
  std::vector<float> padC0(counts[0]), padC1(counts[0]);
  for(size_t i = 0; i < counts[0]; ++i)
  {
    padC0[i] = pad[2 * i];
    padC1[i] = pad[2 * i + 1];
  }
  build_structures(s, padC0, padC1, d->numPrimMeshes);
  return s;
}

void th_destroy(void* p) { delete static_cast<Scene*>(p); }

uint32_t th_num_tris(void* p) { return uint32_t(static_cast<Scene*>(p)->world.size()); }
void     th_sizes(void* p, uint32_t* out4)
{
  Scene* s = static_cast<Scene*>(p);
  out4[0] = uint32_t(s->flat.wide.size()); out4[1] = uint32_t(s->blasWide.size()); out4[2] = uint32_t(s->tlas.wide.size()); out4[3] = uint32_t(s->blasTris.size());
}
// world record of triangle w (p0, e1, e2: 9 floats) and its flags -- for the test's double-precision re-evaluation of a disputed candidate
void th_world_tri(void* p, uint32_t w, float* out9, uint32_t* flags)
{
  const TriRec& r = static_cast<Scene*>(p)->world[w];
  const float   v[9] = {r.p0w.x, r.p0w.y, r.p0w.z, r.e1n.x, r.e1n.y, r.e1n.z, r.e2p.x, r.e2p.y, r.e2p.z};
  std::memcpy(out9, v, sizeof(v));
  *flags = __float_as_uint(r.p0w.w) >> 29;
}

// Every candidate of every ray in key order (t, world index), at most maxCand per ray: mode 0 brute force over all world triangles with the
// product's tri_test, 1 the flat walk, 2 the two-level walk -- traverse<TM_RAW_ALL> restarted behind the previous candidate, exactly what the
// exact fallback kernels (k_closest_x / k_shadow_x) do.  outW / outT: nrays x maxCand (0xffffffff: no further candidate).  Returns the
// number of traversal-stack overflows (must be 0).
uint32_t th_candidates(void* p, int mode, uint32_t nrays, const float* org, const float* dir, float tmax, uint32_t maxCand, uint32_t* outW, float* outT)
{
  Scene*   s = static_cast<Scene*>(p);
  Counters total;
  std::memset(&total, 0, sizeof(total));
#pragma omp parallel
  {
    std::vector<uint32_t> stack(size_t(STACK_LDS) * TRACE_BLOCK);
    Counters              cnt;
    std::memset(&cnt, 0, sizeof(cnt));
#pragma omp for schedule(dynamic, 64)
    for(long long r = 0; r < (long long)nrays; ++r)
    {
      const f3 o = f3{org[3 * r], org[3 * r + 1], org[3 * r + 2]}, d = f3{dir[3 * r], dir[3 * r + 1], dir[3 * r + 2]};
      float    tPrev = 0.0f;
      uint32_t wPrev = 0xffffffffu;
      for(uint32_t c = 0; c < maxCand; ++c)
      {
        uint32_t bw = 0xffffffffu;
        float    bt = 0.f;
        if(mode == 0)
        {
          bool found = false;
          for(const TriRec& tr : s->world)
          {
            const uint32_t wbits = __float_as_uint(tr.p0w.w), w = wbits & TRI_INDEX_MASK;
            float          t, u, v;
            if(tri_test(tr, wbits >> 29, o, d, t, u, v) && t < tmax && key_less(tPrev, wPrev, t, w) && (!found || key_less(t, w, bt, bw)))
            {
              found = true;
              bt    = t;
              bw    = w;
            }
          }
        }
        else
        {
          RayHit h;
          bool   dummy;
          if(mode == 1)
            traverse<TM_RAW_ALL, false>(s->dsFlat, o, d, tmax, tPrev, wPrev, 0u, stack.data(), h, dummy, &cnt);
          else
            traverse<TM_RAW_ALL, true>(s->dsTwo, o, d, tmax, tPrev, wPrev, 0u, stack.data(), h, dummy, &cnt);
          if(h.slot != BVH_NONE)
          {
            bt = h.t;
            bw = h.w & TRI_INDEX_MASK;
          }
        }
        outW[size_t(r) * maxCand + c] = bw;
        outT[size_t(r) * maxCand + c] = bt;
        if(bw == 0xffffffffu)
        {
          for(uint32_t k = c + 1; k < maxCand; ++k)
          {
            outW[size_t(r) * maxCand + k] = 0xffffffffu;
            outT[size_t(r) * maxCand + k] = 0.f;
          }
          break;
        }
        tPrev = bt;
        wPrev = bw;
      }
    }
#if defined(TH_ROBUST_T2) || defined(TH_CERTIFIED_T2)
    g_t2Calls += tl_t2Calls; g_t2Double += tl_t2Double;
    tl_t2Calls = tl_t2Double = 0;
#endif
#ifdef TH_CERTIFIED_T2
    g_t2Accepts += tl_t2Accepts;
    tl_t2Accepts = 0;
#endif
#pragma omp critical
    total.stackOverflow += cnt.stackOverflow;
  }
  return total.stackOverflow;
}

// The product's per-ray settle functions (pt_settle.h: what k_tail runs per lane) against the contract's exact loop, ray by ray.
//   kind 0: closest-hit ray (T5), kind 1: shadow ray (T6, bounded by tmax[r]);  two: 0 flat structure, 1 two-level structure;
//   exact 0: tail_closest / tail_shadow (pass A, pass B, consume_rejected_draws, fallback), exact 1: the key-ordered loop with one
//   alpha_test per non-opaque candidate (k_closest_x / k_shadow_x = the definition), exact 2: the TRACE MACHINE of the persistent kernels
//   (pt_machine.h lane_begin / lane_inner / lane_leaf / lane_pop / lane_begin_count driven like k_closest_p / k_shadow_p drive one lane; the
//   few lines of their service round -- pass A -> pass B transition, bulk draws, hand-over to the exact loop -- are restated here).
// out per ray: w (world triangle index of the hit, 0xffffffff none; for shadow rays 1 / 0 = in shadow or not), t, u, v, seed afterwards,
// number of alpha draws counted.  Returns the number of traversal-stack overflows.
uint32_t th_settle(void* p, int kind, int two, int exact, int variant, uint32_t nrays, const float* org, const float* dir, const float* tmax, const uint32_t* seeds, uint32_t* outW,
                   float* outTUV, uint32_t* outSeed, uint32_t* outDraws)
{
  Scene*             s = static_cast<Scene*>(p);
  const DeviceScene& S = two ? s->dsTwo : s->dsFlat;
  std::vector<float4> rayO(nrays), rayD(nrays), absorb(nrays), neeDir(nrays), hit(nrays);
  for(uint32_t r = 0; r < nrays; ++r)
  {
    rayO[r]   = make_float4(org[3 * r], org[3 * r + 1], org[3 * r + 2], 0.f);
    rayD[r]   = make_float4(dir[3 * r], dir[3 * r + 1], dir[3 * r + 2], __uint_as_float(seeds[r]));
    neeDir[r] = make_float4(dir[3 * r], dir[3 * r + 1], dir[3 * r + 2], 1.f);
    absorb[r] = make_float4(0.f, 0.f, 0.f, tmax ? tmax[r] : PT_INFINITY);
  }
  Counters total;
  std::memset(&total, 0, sizeof(total));
#pragma omp parallel
  {
    std::vector<uint32_t> stack(size_t(STACK_LDS) * TRACE_BLOCK);
    Counters              cnt;
    std::memset(&cnt, 0, sizeof(cnt));
    RenderBuffers rb;
    std::memset(&rb, 0, sizeof(rb));
    rb.ps.rayO.p = rayO.data(); rb.ps.rayD.p = rayD.data(); rb.ps.absorb.p = absorb.data(); rb.ps.neeDir.p = neeDir.data(); rb.ps.hit.p = hit.data();
    rb.counters = &cnt;
#pragma omp for schedule(dynamic, 64)
    for(long long r = 0; r < (long long)nrays; ++r)
    {
      const f3    o = xyz(rayO[r]), d = xyz(rayD[r]);
      uint32_t    seed = seeds[r], draws = 0;
      bool machineFallback = false;
      unsigned long long innerSteps = 0, leafSteps = 0;
      if(exact == 2)
      {
        TraceLane             L;
        std::vector<uint32_t> spill(STACK_SPILL);
        int                   maxSp = 0;
        lane_begin(L, o, d, kind == 0 ? PT_INFINITY : absorb[r].w, S.numTris == 0);
        for(;;)
        {
          while(!L.done)
          {
            if(!(L.cur & BVH_LEAF))
            {
              ++innerSteps;
              if(two) lane_inner<false, true>(S, L, stack.data(), spill.data(), &cnt);
              else lane_inner<false, false>(S, L, stack.data(), spill.data(), &cnt);
              maxSp = L.sp > maxSp ? L.sp : maxSp;
            }
            if(!L.done && (L.cur & BVH_LEAF))
            {
              ++leafSteps;
              if(two) lane_leaf<false, true>(S, L, stack.data(), spill.data());
              else lane_leaf<false, false>(S, L, stack.data(), spill.data());
            }
          }
          // service round of k_closest_p / k_shadow_p for this lane
          bool fallback = (L.flags & TF_SAW_FRAC) != 0;
          if(!fallback && L.pass == 0 && (L.flags & TF_SAW_ZERO) && !pass_a_settles(L.bslot, L.bt, L.zeroMaxT, L.zeroMaxT2, L.zeroMaxT3, L.cnt))
          {
            if(two) lane_begin_count<true>(L); else lane_begin_count<false>(L);
            continue;
          }
          if(!fallback)
          {
            uint32_t nDraw = L.cnt;
            if(L.bslot != BVH_NONE && !((L.bw >> 29) & TRI_OPAQUE))
              ++nDraw;
            uint32_t s2 = seed;
            if(consume_rejected_draws(s2, nDraw))
            {
              draws = nDraw;
              if(kind == 0)
              {
                outW[r]       = L.bslot == BVH_NONE ? BVH_NONE : (L.bw & TRI_INDEX_MASK);
                outTUV[3 * r] = L.bslot == BVH_NONE ? PT_INFINITY : L.bt; outTUV[3 * r + 1] = L.bslot == BVH_NONE ? 0.f : L.bu; outTUV[3 * r + 2] = L.bslot == BVH_NONE ? 0.f : L.bv;
                outSeed[r]    = nDraw ? s2 : seed;
              }
              else
              {
                outW[r]       = L.bslot != BVH_NONE ? 1u : 0u;
                outTUV[3 * r] = outTUV[3 * r + 1] = outTUV[3 * r + 2] = 0.f;
                outSeed[r]    = variant == PT_VARIANT_RTX ? seed : s2;
              }
            }
            else
              fallback = true;
          }
          machineFallback = fallback;  // queueX / queueX2: the exact kernels take over (below)
          g_spHist[maxSp < 64 ? maxSp : 64]++;
          g_innerSteps += innerSteps;
          g_leafSteps += leafSteps;
          break;
        }
      }
      if(exact == 2 && !machineFallback)
      {
      }
      else if(!exact)
      {
        if(kind == 0)
        {
          if(two) tail_closest<true>(S, rb, uint32_t(r), stack.data(), draws); else tail_closest<false>(S, rb, uint32_t(r), stack.data(), draws);
          const float4 h = hit[r];
          outW[r]        = __float_as_uint(h.y);   // flat: leaf slot; two-level: world index -- translated below
          if(outW[r] != BVH_NONE && !two)
            outW[r] = __float_as_uint(S.tris[outW[r]].p0w.w) & TRI_INDEX_MASK;
          outTUV[3 * r] = h.x; outTUV[3 * r + 1] = h.z; outTUV[3 * r + 2] = h.w;
          outSeed[r]    = __float_as_uint(rayD[r].w);
        }
        else
        {
          const bool sh = two ? tail_shadow<true>(S, rb, uint32_t(r), stack.data(), variant, seed, draws) : tail_shadow<false>(S, rb, uint32_t(r), stack.data(), variant, seed, draws);
          outW[r]       = sh ? 1u : 0u;
          outTUV[3 * r] = outTUV[3 * r + 1] = outTUV[3 * r + 2] = 0.f;
          outSeed[r]    = seed;
        }
      }
      else
      {  // trace contract T5 / T6: candidates strictly in key order, an opaque one commits, a non-opaque one draws once
        const uint32_t seed0 = seed;
        const float    lim   = kind == 0 ? PT_INFINITY : absorb[r].w;
        float          tPrev = 0.0f;
        uint32_t       wPrev = 0xffffffffu;
        RayHit         h;
        bool           dummy, found = false;
        for(;;)
        {
          if(two) traverse<TM_RAW_ALL, true>(S, o, d, lim, tPrev, wPrev, 0u, stack.data(), h, dummy, &cnt); else traverse<TM_RAW_ALL, false>(S, o, d, lim, tPrev, wPrev, 0u, stack.data(), h, dummy, &cnt);
          if(h.slot == BVH_NONE)
            break;
          if((h.w >> 29) & TRI_OPAQUE)
          {
            found = true;
            break;
          }
          ++draws;
          if(alpha_test(S, h.slot, h.u, h.v, seed))
          {
            found = true;
            break;
          }
          tPrev = h.t;
          wPrev = h.w & TRI_INDEX_MASK;
        }
        if(kind == 0)
        {
          outW[r]       = found ? (h.w & TRI_INDEX_MASK) : BVH_NONE;
          outTUV[3 * r] = found ? h.t : PT_INFINITY; outTUV[3 * r + 1] = found ? h.u : 0.f; outTUV[3 * r + 2] = found ? h.v : 0.f;
          outSeed[r]    = seed;
        }
        else
        {
          outW[r]       = found ? 1u : 0u;
          outTUV[3 * r] = outTUV[3 * r + 1] = outTUV[3 * r + 2] = 0.f;
          outSeed[r]    = variant == PT_VARIANT_RTX ? seed0 : seed;
        }
      }
      outDraws[r] = draws;
    }
#if defined(TH_ROBUST_T2) || defined(TH_CERTIFIED_T2)
    g_t2Calls += tl_t2Calls; g_t2Double += tl_t2Double;
    tl_t2Calls = tl_t2Double = 0;
#endif
#ifdef TH_CERTIFIED_T2
    g_t2Accepts += tl_t2Accepts;
    tl_t2Accepts = 0;
#endif
#pragma omp critical
    total.stackOverflow += cnt.stackOverflow;
  }
  return total.stackOverflow;
}

// DESIGN EXPERIMENT (tools/steps_experiment.py; nothing of the product runs here except tri_test): how many DEPENDENT memory round trips does a
// closest-hit walk of the flat structure need per ray -- the quantity that bounds the trace stages -- for node width 4
// (the product's layout) or 8, and with the triangles of a node's leaf children fetched together in one round trip instead of one per
// triangle?  The binary tree of the flat structure is collapsed again at the requested width (same greedy rule as k_collapse) and walked
// nearest-first with a plain slab test; a step = one node fetch, or one (batch of) triangle fetch(es).
// out per ray: steps, nodes visited, triangles tested.
void th_step_model(void* p, int width, int batchLeaves, uint32_t nrays, const float* org, const float* dir, const float* tmaxIn, uint32_t* out3)
{
  Scene*     s = static_cast<Scene*>(p);
  const Bvh& b = s->flat;
  const uint32_t n = uint32_t(b.tris.size());
  struct WN { int cnt; Box box[8]; uint32_t ref[8]; };
  std::vector<WN> nodes;
  if(n >= 2)
  {
    struct Item { uint32_t b2, wide; };
    std::vector<Item> queue{{0u, 0u}};
    nodes.resize(1);
    for(size_t qi = 0; qi < queue.size(); ++qi)
    {
      const Item it = queue[qi];
      uint32_t   id[8];
      int        cnt = 0;
      id[cnt++] = b.cl[it.b2];
      id[cnt++] = b.cr[it.b2];
      while(cnt < width)
      {
        int   best = -1;
        float bestA = -1.f;
        for(int k = 0; k < cnt; ++k)
          if(!(id[k] & BVH_LEAF) && half_area_h(b.innerBox[id[k]]) > bestA)
          {
            bestA = half_area_h(b.innerBox[id[k]]);
            best  = k;
          }
        if(best < 0)
          break;
        const uint32_t node = id[best];
        id[best]            = id[cnt - 1];
        --cnt;
        id[cnt++] = b.cl[node];
        id[cnt++] = b.cr[node];
      }
      WN w;
      w.cnt = cnt;
      for(int k = 0; k < cnt; ++k)
      {
        if(id[k] & BVH_LEAF)
        {
          w.box[k] = b.leafBox[id[k] & ~BVH_LEAF];
          w.ref[k] = id[k];
        }
        else
        {
          w.box[k] = b.innerBox[id[k]];
          w.ref[k] = uint32_t(nodes.size());
          nodes.emplace_back();
          queue.push_back({id[k], w.ref[k]});
        }
      }
      nodes[it.wide] = w;
    }
  }
#pragma omp parallel for schedule(dynamic, 64)
  for(long long r = 0; r < (long long)nrays; ++r)
  {
    const f3 o = f3{org[3 * r], org[3 * r + 1], org[3 * r + 2]}, d = f3{dir[3 * r], dir[3 * r + 1], dir[3 * r + 2]};
    const float id3[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z}, o3[3] = {o.x, o.y, o.z};
    float       best = tmaxIn ? tmaxIn[r] : 3.0e38f;
    uint32_t    steps = 0, nn = 0, nt = 0;
    auto        test_tri = [&](uint32_t slot) {
      const TriRec& tr = b.tris[slot];
      float         t, u, v;
      ++nt;
      if(tri_test(tr, __float_as_uint(tr.p0w.w) >> 29, o, d, t, u, v) && t > 0.f && t < best)
        best = t;
    };
    if(n == 1)
    {
      test_tri(0);
      steps = 1;
    }
    else if(n >= 2)
    {
      uint32_t stack[256];
      int      sp = 0;
      uint32_t cur = 0;
      for(;;)
      {
        if(cur & BVH_LEAF)
        {
          ++steps;
          test_tri(cur & ~BVH_LEAF);
        }
        else
        {
          ++steps;
          ++nn;
          const WN& w = nodes[cur];
          float     tn[8];
          uint32_t  rf[8];
          int       nh = 0;
          bool      anyLeaf = false;
          for(int k = 0; k < w.cnt; ++k)
          {
            float t0 = 0.f, t1 = best;
            for(int a = 0; a < 3; ++a)
            {
              const float ta = (w.box[k].lo[a] - o3[a]) * id3[a], tb = (w.box[k].hi[a] - o3[a]) * id3[a];
              t0 = std::fmax(t0, std::fmin(ta, tb));
              t1 = std::fmin(t1, std::fmax(ta, tb));
            }
            if(t0 * 0.9999996f <= t1 * 1.0000004f)
            {
              if(batchLeaves && (w.ref[k] & BVH_LEAF))
              {
                anyLeaf = true;
                test_tri(w.ref[k] & ~BVH_LEAF);  // fetched together with the node's other hit leaves: one round trip (counted below)
              }
              else
              {
                tn[nh] = t0;
                rf[nh] = w.ref[k];
                ++nh;
              }
            }
          }
          if(anyLeaf)
            ++steps;
          // far-to-near onto the stack
          for(int i = 0; i < nh; ++i)
            for(int j = i + 1; j < nh; ++j)
              if(tn[j] > tn[i])
              {
                std::swap(tn[i], tn[j]);
                std::swap(rf[i], rf[j]);
              }
          for(int i = 0; i < nh && sp < 256; ++i)
            if(!batchLeaves || tn[i] <= best)
              stack[sp++] = rf[i];
        }
        if(sp == 0)
          break;
        cur = stack[--sp];
      }
    }
    out3[3 * r] = steps; out3[3 * r + 1] = nn; out3[3 * r + 2] = nt;
  }
}

// ---- whole frames with the product's shading source on the host ---------------------------------------------------------------------------
// What k_generate / k_tail / k_accumulate do per lane (pt_shade.h, pt_settle.h), run as a loop over the path slots of one frame at a time:
// camera ray, then per bounce closest hit -> shade_path -> shadow ray -> NEE add + Russian roulette, then the running mean.  Same tile / slot
// layout as the device (32 x 32 pixel tiles of 16 8x8 blocks).  The caller compares the image with the oracle bit for bit.
int th_set_env(void* p, const float* rgba, int w, int h, float* integral)
{
  Scene* s = static_cast<Scene*>(p);
  s->env.resize(size_t(w) * h);
  std::memcpy(s->env.data(), rgba, sizeof(float) * 4 * size_t(w) * h);
  s->envAccel.resize(size_t(w) * h);
  float avg = 0.f;
  const int rc = pt_build_env_accel(rgba, w, h, s->envAccel.data(), integral, &avg);
  for(DeviceScene* d : {&s->dsFlat, &s->dsTwo})
  {
    d->env = s->env.data(); d->envAccel = s->envAccel.data(); d->envW = w; d->envH = h;
  }
  return rc;
}
void th_set_camera(void* p, const pt_SceneCamera* cam, const pt_SunAndSky* ss)
{
  Scene* s = static_cast<Scene*>(p);
  for(DeviceScene* d : {&s->dsFlat, &s->dsTwo})
  {
    d->camera = *cam;
    d->sunsky = *ss;
  }
}
// frames 0 .. frames-1 of `st` (st->frame is ignored) accumulated like the device does; out: row-major width x height x 4.
// rank / nranks: the image-tile shard of pt_set_shard (tiles with (tx + ty) % nranks == rank, in increasing order: pt_resize); only the pixels
// of the rank's own tiles are written.
uint32_t th_render_shard(void* p, int two, const pt_RtxState* stIn, int variant, int frames, int rank, int nranks, float* out);
uint32_t th_render(void* p, int two, const pt_RtxState* stIn, int variant, int frames, float* out) { return th_render_shard(p, two, stIn, variant, frames, 0, 1, out); }
uint32_t th_render_shard(void* p, int two, const pt_RtxState* stIn, int variant, int frames, int rank, int nranks, float* out)
{
  Scene*             s = static_cast<Scene*>(p);
  const DeviceScene& S = two ? s->dsTwo : s->dsFlat;
  const int          W = stIn->size[0], H = stIn->size[1];
  FrameParams        fp;
  std::memset(&fp, 0, sizeof(fp));
  fp.st = *stIn; fp.width = W; fp.height = H; fp.tilesX = (W + PT_TILE - 1) / PT_TILE; fp.tilesY = (H + PT_TILE - 1) / PT_TILE;
  std::vector<uint32_t> slotTile;
  for(int ty = 0; ty < fp.tilesY; ++ty)
    for(int tx = 0; tx < fp.tilesX; ++tx)
      if((tx + ty) % nranks == rank)
        slotTile.push_back(uint32_t(ty * fp.tilesX + tx));
  fp.rank = rank; fp.nranks = nranks; fp.numLocalTiles = uint32_t(slotTile.size()); fp.numSlots = fp.numLocalTiles * 1024u; fp.batch = 1; fp.variant = variant;
  if(slotTile.empty())
    return 0;
  const uint32_t        n = fp.numSlots;
  std::vector<float4>   st9[9];
  for(auto& v : st9)
    v.assign(n, make_float4(0, 0, 0, 0));
  std::vector<float4>   frame(n, make_float4(0, 0, 0, 0));
  Counters total;
  std::memset(&total, 0, sizeof(total));
  RenderBuffers rb;
  std::memset(&rb, 0, sizeof(rb));
  rb.ps.rayO.p = st9[0].data(); rb.ps.rayD.p = st9[1].data(); rb.ps.thr.p = st9[2].data(); rb.ps.rad.p = st9[3].data(); rb.ps.absorb.p = st9[4].data();
  rb.ps.neeDir.p = st9[5].data(); rb.ps.neeRad.p = st9[6].data(); rb.ps.hit.p = st9[7].data(); rb.ps.sum.p = st9[8].data();
  rb.frame = frame.data(); rb.slotTile = slotTile.data();
  for(int f = 0; f < frames; ++f)
  {
    fp.st.frame = f;
    for(int smp = 0; smp < fp.st.maxSamples; ++smp)
    {
      fp.sample = smp;
#pragma omp parallel
      {
        std::vector<uint32_t> stack(size_t(STACK_LDS) * TRACE_BLOCK);
        Counters              cnt;
        std::memset(&cnt, 0, sizeof(cnt));
        RenderBuffers lrb = rb;
        lrb.counters      = &cnt;
#pragma omp for schedule(dynamic, 256)
        for(long long sl = 0; sl < (long long)n; ++sl)
        {
          const uint32_t slot = uint32_t(sl);
          int            px, py;
          if(!slot_pixel(fp, lrb.slotTile, slot, px, py))
            continue;
          generate_ray(S, lrb, fp, slot, 0u, px, py);
          uint32_t nAlpha = 0;
          for(int depth = 0; depth < fp.st.maxDepth; ++depth)
          {
            if(two) tail_closest<true>(S, lrb, slot, stack.data(), nAlpha); else tail_closest<false>(S, lrb, slot, stack.data(), nAlpha);
            uint32_t  events = 0;
            const int to     = shade_path<-1>(S, lrb, fp, slot, depth, events);
            bool      survive = to == SHADE_TO_NEXT;
            if(to == SHADE_TO_SHADOW)
            {
              uint32_t   seed;
              const bool inShadow = two ? tail_shadow<true>(S, lrb, slot, stack.data(), variant, seed, nAlpha) : tail_shadow<false>(S, lrb, slot, stack.data(), variant, seed, nAlpha);
              survive             = finish_bounce_core(lrb, slot, inShadow, seed) && depth != fp.st.maxDepth - 1;
            }
            if(!survive)
              break;
          }
        }
#pragma omp critical
        total.stackOverflow += cnt.stackOverflow;
      }
#pragma omp parallel for schedule(static)
      for(long long ps = 0; ps < (long long)n; ++ps)
      {
        int px, py;
        if(slot_pixel(fp, rb.slotTile, uint32_t(ps), px, py))
          accumulate_pixel(rb, fp, uint32_t(ps));
      }
    }
  }
  for(uint32_t slot = 0; slot < n; ++slot)
  {  // k_untile
    int px, py;
    if(slot_pixel(fp, rb.slotTile, slot, px, py))
      std::memcpy(out + (size_t(py) * W + px) * 4, &frame[slot], 16);
  }
  return total.stackOverflow;
}

}  // extern "C"
