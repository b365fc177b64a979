// Device acceleration-structure build for gfx950.  Replaces AccelStructure::create of the reference
// (src/accelstruct.cpp:55-162: one BLAS per prim-mesh + one TLAS instance per node, built by the Vulkan driver).
//
// pt_accel_build: ONE hierarchy over a set of primitives, as 4-wide nodes (WideNode) over leaf-ordered records (TriRec):
//   k_world_tris   instance transforms applied in fp32 (trace contract T1) -> TriRec + centroid   (or ready-made records: dProxies)
//   topology       device binned SAH (default, pt_sahdev.h) | host SAH (cross-check) | PLOC | Karras radix tree (k_morton, radix sort, k_hierarchy)
//   k_gather       records in leaf order + padded leaf boxes
//   k_refit        bottom-up AABB merge with per-node arrival counters (agent-scope fence per hand-off)
//   k_emit         64-byte binary nodes holding both child boxes
//   k_collapse     4-wide nodes, greedy by surface area, one level per launch
// It is used three ways: over the world-space triangles of every node (the flat structure, default), over the object-space triangles of
// one prim-mesh (a BLAS of the two-level structure: pt_blas_build), and over the world boxes of the instances given as "diagonal" records
// (the TLAS: pt_tlas_build).  Temporaries come out of the caller's arena (PtScratch) or are allocated singly; everything runs on the
// caller's stream.
#include <hip/hip_runtime.h>
#include <vector>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include "pt_device.h"
#include "pt_internal.h"
#include "pt_sahdev.h"
#include "pt_cnode.h"

namespace {

constexpr int SORT_ITEMS = 2048;  // keys per one-wave block

PT_DEV uint32_t order_bits(float f)
{
  uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__host__ __device__ inline float unorder_bits(uint32_t u)
{
  u ^= ((u >> 31) ? 0x80000000u : 0xffffffffu);
  float f;
#ifdef __HIP_DEVICE_COMPILE__
  f = __uint_as_float(u);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}

PT_DEV f3 load_pos(const float4* vertices, uint32_t v)
{
  float4 a = vertices[size_t(v) * 2];
  return f3{a.x, a.y, a.z};
}

// ---- T1: world-space triangles --------------------------------------------------------------------
__global__ void k_world_tris(uint32_t numTris, const InstanceRec* __restrict__ inst, uint32_t numInst, const float4* __restrict__ vertices,
                             const uint32_t* __restrict__ indices, TriRec* __restrict__ out, AlphaRec* __restrict__ alphaOut, float4* __restrict__ cen, uint32_t* __restrict__ bounds)
{
  uint32_t w     = blockIdx.x * blockDim.x + threadIdx.x;
  bool     valid = w < numTris;
  f3       c     = f3{0, 0, 0};
  if(valid)
  {
    // binary search: last instance with triBase <= w
    uint32_t lo = 0, hi = numInst - 1;
    while(lo < hi)
    {
      uint32_t mid = (lo + hi + 1) >> 1;
      if(inst[mid].triBase <= w)
        lo = mid;
      else
        hi = mid - 1;
    }
    // (the LAST instance with triBase <= w is the owner: empty instances sharing a base sort before it)
    const InstanceRec& I = inst[lo];
    uint32_t           k = w - I.triBase;
    const uint32_t*    t = indices + I.firstIndex + 3 * size_t(k);
    Affine             M = I.objectToWorld;
    f3                 p0 = xform_point(M, load_pos(vertices, I.vertexOffset + t[0]));
    f3                 p1 = xform_point(M, load_pos(vertices, I.vertexOffset + t[1]));
    f3                 p2 = xform_point(M, load_pos(vertices, I.vertexOffset + t[2]));
    f3                 e1 = p1 - p0, e2 = p2 - p0;
    TriRec             r;
    r.p0w  = make_float4(p0.x, p0.y, p0.z, __uint_as_float(w | (I.flags << 29)));
    r.e1n  = make_float4(e1.x, e1.y, e1.z, __uint_as_float(lo));
    r.e2p  = make_float4(e2.x, e2.y, e2.z, __uint_as_float(k));
    out[w] = r;
    {  // any-hit inputs (raw texcoords of the three vertices + material)
      const float4 b0 = vertices[size_t(I.vertexOffset + t[0]) * 2 + 1], b1 = vertices[size_t(I.vertexOffset + t[1]) * 2 + 1], b2 = vertices[size_t(I.vertexOffset + t[2]) * 2 + 1];
      AlphaRec     ar;
      ar.uv0[0] = b0.x; ar.uv0[1] = b0.y; ar.uv1[0] = b1.x; ar.uv1[1] = b1.y; ar.uv2[0] = b2.x; ar.uv2[1] = b2.y;
      ar.material = uint32_t(I.materialIndex < 0 ? 0 : I.materialIndex);
      ar._pad     = 0;
      alphaOut[w] = ar;
    }
    f3 mn  = f3{fminf(p0.x, fminf(p1.x, p2.x)), fminf(p0.y, fminf(p1.y, p2.y)), fminf(p0.z, fminf(p1.z, p2.z))};
    f3 mx  = f3{fmaxf(p0.x, fmaxf(p1.x, p2.x)), fmaxf(p0.y, fmaxf(p1.y, p2.y)), fmaxf(p0.z, fmaxf(p1.z, p2.z))};
    c      = (mn + mx) * 0.5f;
    cen[w] = make_float4(c.x, c.y, c.z, 0.f);
  }
  // wave-level min/max, then one atomic per wave and component
  float mnx = valid ? c.x : FLT_MAX, mny = valid ? c.y : FLT_MAX, mnz = valid ? c.z : FLT_MAX;
  float mxx = valid ? c.x : -FLT_MAX, mxy = valid ? c.y : -FLT_MAX, mxz = valid ? c.z : -FLT_MAX;
  for(int off = 32; off > 0; off >>= 1)
  {
    mnx = fminf(mnx, __shfl_xor(mnx, off));
    mny = fminf(mny, __shfl_xor(mny, off));
    mnz = fminf(mnz, __shfl_xor(mnz, off));
    mxx = fmaxf(mxx, __shfl_xor(mxx, off));
    mxy = fmaxf(mxy, __shfl_xor(mxy, off));
    mxz = fmaxf(mxz, __shfl_xor(mxz, off));
  }
  if((threadIdx.x & 63) == 0)
  {
    atomicMin(&bounds[0], order_bits(mnx));
    atomicMin(&bounds[1], order_bits(mny));
    atomicMin(&bounds[2], order_bits(mnz));
    atomicMax(&bounds[3], order_bits(mxx));
    atomicMax(&bounds[4], order_bits(mxy));
    atomicMax(&bounds[5], order_bits(mxz));
  }
}

PT_DEV uint32_t spread10(uint32_t v)
{
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ void k_morton(uint32_t n, const float4* __restrict__ cen, const uint32_t* __restrict__ bounds, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  f3     lo = f3{unorder_bits(bounds[0]), unorder_bits(bounds[1]), unorder_bits(bounds[2])};
  f3     hi = f3{unorder_bits(bounds[3]), unorder_bits(bounds[4]), unorder_bits(bounds[5])};
  float4 c  = cen[i];
  auto   q  = [](float v, float a, float b) {
    float e = b - a;
    float t = e > 0.f ? (v - a) / e : 0.f;
    int   k = int(t * 1024.0f);
    return uint32_t(k < 0 ? 0 : (k > 1023 ? 1023 : k));
  };
  keys[i] = (spread10(q(c.x, lo.x, hi.x)) << 2) | (spread10(q(c.y, lo.y, hi.y)) << 1) | spread10(q(c.z, lo.z, hi.z));
  vals[i] = i;
}

// ---- LSD radix sort, 8 bits per pass; one wave (64 lanes) owns SORT_ITEMS consecutive keys ----------
__global__ void __launch_bounds__(64) k_sort_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ hist, uint32_t numBlocks)
{
  __shared__ uint32_t lh[256];
  for(int d = threadIdx.x; d < 256; d += 64)
    lh[d] = 0;
  __syncthreads();
  uint32_t base = blockIdx.x * SORT_ITEMS;
  for(int it = 0; it < SORT_ITEMS / 64; ++it)
  {
    uint32_t i = base + it * 64 + threadIdx.x;
    if(i < n)
      atomicAdd(&lh[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  for(int d = threadIdx.x; d < 256; d += 64)
    hist[size_t(d) * numBlocks + blockIdx.x] = lh[d];
}

// exclusive scan of `total` counters with a single 256-thread block
__global__ void __launch_bounds__(256) k_sort_scan(uint32_t* __restrict__ hist, uint32_t total)
{
  __shared__ uint32_t part[256];
  uint32_t            chunk = (total + 255) / 256;
  uint32_t            b     = threadIdx.x * chunk;
  uint32_t            e     = b + chunk < total ? b + chunk : total;
  uint32_t            s     = 0;
  for(uint32_t i = b; i < e; ++i)
    s += hist[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for(int off = 1; off < 256; off <<= 1)
  {
    uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;
  for(uint32_t i = b; i < e; ++i)
  {
    uint32_t v = hist[i];
    hist[i]    = run;
    run += v;
  }
}

__global__ void __launch_bounds__(64) k_sort_scatter(const uint32_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn, uint32_t* __restrict__ keysOut,
                                                     uint32_t* __restrict__ valsOut, uint32_t n, int shift, const uint32_t* __restrict__ hist, uint32_t numBlocks)
{
  __shared__ uint32_t offs[256];
  for(int d = threadIdx.x; d < 256; d += 64)
    offs[d] = hist[size_t(d) * numBlocks + blockIdx.x];
  __syncthreads();
  const uint32_t           base = blockIdx.x * SORT_ITEMS;
  const unsigned long long lt   = (1ull << threadIdx.x) - 1ull;
  for(int it = 0; it < SORT_ITEMS / 64; ++it)
  {
    uint32_t           i      = base + it * 64 + threadIdx.x;
    bool               active = i < n;
    uint32_t           key    = active ? keysIn[i] : 0u;
    uint32_t           val    = active ? valsIn[i] : 0u;
    uint32_t           d      = (key >> shift) & 255u;
    unsigned long long same   = __ballot(active);
#pragma unroll
    for(int b = 0; b < 8; ++b)
    {
      bool               bit = (d >> b) & 1u;
      unsigned long long bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    uint32_t rank = __popcll(same & lt);
    uint32_t pos  = 0;
    if(active)
      pos = offs[d] + rank;
    __syncthreads();
    if(active && rank == 0)
      offs[d] += __popcll(same);
    __syncthreads();
    if(active)
    {
      keysOut[pos] = key;
      valsOut[pos] = val;
    }
  }
}

// ---- leaf gather: sorted TriRec + padded leaf boxes ---------------------------------------------------
PT_DEV void tri_box(const TriRec& r, f3& lo, f3& hi)
{
  f3 p0 = xyz(r.p0w);
  f3 p1 = p0 + xyz(r.e1n);
  f3 p2 = p0 + xyz(r.e2p);
  lo    = f3{fminf(p0.x, fminf(p1.x, p2.x)), fminf(p0.y, fminf(p1.y, p2.y)), fminf(p0.z, fminf(p1.z, p2.z))};
  hi    = f3{fmaxf(p0.x, fmaxf(p1.x, p2.x)), fmaxf(p0.y, fmaxf(p1.y, p2.y)), fmaxf(p0.z, fmaxf(p1.z, p2.z))};
  // conservative padding: the slab test must never reject a triangle that the triangle test accepts.
  // p0+e1 is not exactly p1 (1 ulp) and the hit point itself carries a few ulps of error.
  f3 m  = f3{fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y)), fmaxf(fabsf(lo.z), fabsf(hi.z))};
  f3 pad = m * 4e-6f + 1e-30f;
  lo    = lo - pad;
  hi    = hi + pad;
}

__global__ void k_gather(uint32_t n, const uint32_t* __restrict__ vals, const TriRec* __restrict__ in, TriRec* __restrict__ out, const AlphaRec* __restrict__ alphaIn,
                         AlphaRec* __restrict__ alphaOut, float4* __restrict__ leafLo, float4* __restrict__ leafHi)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  TriRec r    = in[vals[i]];
  out[i]      = r;
  alphaOut[i] = alphaIn[vals[i]];
  f3 lo, hi;
  tri_box(r, lo, hi);
  // .w of the lower corner: 1 when the subtree holds a non-opaque triangle (propagated by k_refit, becomes the
  // BVH_ALPHA tag of child references: the alpha-only traversals skip everything else)
  leafLo[i] = make_float4(lo.x, lo.y, lo.z, ((__float_as_uint(r.p0w.w) >> 29) & TRI_OPAQUE) ? 0.f : 1.f);
  leafHi[i] = make_float4(hi.x, hi.y, hi.z, 0.f);
}

// ---- Karras 2012 ------------------------------------------------------------------------------------
PT_DEV int delta(const uint32_t* keys, int n, int i, int j)
{
  if(j < 0 || j >= n)
    return -1;
  uint32_t a = keys[i], b = keys[j];
  if(a == b)
    return 32 + __clz(uint32_t(i) ^ uint32_t(j));
  return __clz(a ^ b);
}

__global__ void k_hierarchy(int n, const uint32_t* __restrict__ keys, uint32_t* __restrict__ childL, uint32_t* __restrict__ childR, uint32_t* __restrict__ parentOfInner,
                            uint32_t* __restrict__ parentOfLeaf)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n - 1)
    return;
  int d    = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while(delta(keys, n, i, i + lmax * d) > dmin)
    lmax <<= 1;
  int l = 0;
  for(int t = lmax >> 1; t >= 1; t >>= 1)
    if(delta(keys, n, i, i + (l + t) * d) > dmin)
      l += t;
  int j     = i + l * d;
  int dnode = delta(keys, n, i, j);
  int s     = 0;
  int t     = l;
  do
  {
    t = (t + 1) >> 1;
    if(delta(keys, n, i, i + (s + t) * d) > dnode)
      s += t;
  } while(t > 1);
  int gamma = i + s * d + (d < 0 ? d : 0);
  int lo = i < j ? i : j, hi = i < j ? j : i;
  if(lo == gamma)
  {
    childL[i]           = uint32_t(gamma) | BVH_LEAF;
    parentOfLeaf[gamma] = i;
  }
  else
  {
    childL[i]            = gamma;
    parentOfInner[gamma] = i;
  }
  if(hi == gamma + 1)
  {
    childR[i]               = uint32_t(gamma + 1) | BVH_LEAF;
    parentOfLeaf[gamma + 1] = i;
  }
  else
  {
    childR[i]                = gamma + 1;
    parentOfInner[gamma + 1] = i;
  }
  if(i == 0)
    parentOfInner[0] = BVH_NONE;
}

// Bottom-up merge.  The second thread to reach a node owns it; the hand-off between the two arrivals
// crosses CUs, so it is an agent-scope release (fence + drained vmcnt) before the counter and an
// agent-scope acquire after it (per-CU L1 is never refreshed by other CUs' stores on gfx950).
__global__ void k_refit(int n, const uint32_t* __restrict__ childL, const uint32_t* __restrict__ childR, const uint32_t* __restrict__ parentOfInner,
                        const uint32_t* __restrict__ parentOfLeaf, const float4* __restrict__ leafLo, const float4* __restrict__ leafHi, float4* nodeLo, float4* nodeHi,
                        unsigned int* arrive)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  uint32_t cur = parentOfLeaf[i];
  while(cur != BVH_NONE)
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int prev = __hip_atomic_fetch_add(&arrive[cur], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(prev == 0)
      return;  // first arrival: the sibling subtree is not finished yet
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    uint32_t l = childL[cur], r = childR[cur];
    float4   llo = (l & BVH_LEAF) ? leafLo[l & ~BVH_LEAF] : nodeLo[l];
    float4   lhi = (l & BVH_LEAF) ? leafHi[l & ~BVH_LEAF] : nodeHi[l];
    float4   rlo = (r & BVH_LEAF) ? leafLo[r & ~BVH_LEAF] : nodeLo[r];
    float4   rhi = (r & BVH_LEAF) ? leafHi[r & ~BVH_LEAF] : nodeHi[r];
    nodeLo[cur]  = make_float4(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z), fmaxf(llo.w, rlo.w));
    nodeHi[cur]  = make_float4(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), 0.f);
    cur          = parentOfInner[cur];
  }
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) -------------------------------------------------
// Bottom-up agglomerative build over the Morton-sorted leaves, all on the device: every cluster looks at the `radius` clusters on either side
// of it in the (Morton-ordered) cluster array and picks the one whose union with it has the smallest surface area; mutual choices merge
// into a new inner node; the array is compacted; repeat until one cluster is left.  Unlike the radix tree of Karras 2012 (k_hierarchy) the
// topology follows the surface-area heuristic locally, which is what makes the host SAH builder's trees fast to trace.
// Ties in area are broken by (i xor j): a symmetric key, so runs of identical boxes still pair up as buddies instead of forming one merge per round.
PT_DEV float half_area(float4 lo, float4 hi);
PT_DEV float union_half_area(float4 alo, float4 ahi, float4 blo, float4 bhi)
{
  float dx = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x), dy = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y), dz = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
  return dx * dy + dy * dz + dz * dx;
}
__global__ void k_ploc_init(uint32_t n, const float4* __restrict__ leafLo, const float4* __restrict__ leafHi, uint32_t* __restrict__ cid, float4* __restrict__ clo, float4* __restrict__ chi)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  cid[i] = i | BVH_LEAF;
  clo[i] = leafLo[i];
  chi[i] = leafHi[i];
}
__global__ void __launch_bounds__(256) k_ploc_nn(uint32_t m, int radius, const float4* __restrict__ clo, const float4* __restrict__ chi, uint32_t* __restrict__ nn)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= m)
    return;
  const float4 lo = clo[i], hi = chi[i];
  const int    j0 = int(i) - radius < 0 ? 0 : int(i) - radius, j1 = int(i) + radius > int(m) - 1 ? int(m) - 1 : int(i) + radius;
  float        bestA = FLT_MAX;
  uint32_t     best = BVH_NONE, bestX = 0xffffffffu;
  for(int j = j0; j <= j1; ++j)
  {
    if(j == int(i))
      continue;
    const float    a = union_half_area(lo, hi, clo[j], chi[j]);
    const uint32_t x = i ^ uint32_t(j);
    if(a < bestA || (a == bestA && x < bestX))
    {
      bestA = a;
      best  = uint32_t(j);
      bestX = x;
    }
  }
  nn[i] = best;
}
// mutual nearest neighbours merge: the lower index keeps the merged cluster, the higher one is dropped.  Inner node ids are handed out
// downwards from n-2, so that the last merge -- the root -- is node 0 like in the other builders.
__global__ void __launch_bounds__(256) k_ploc_merge(uint32_t m, uint32_t numInner, const uint32_t* __restrict__ nn, uint32_t* __restrict__ cid, float4* __restrict__ clo, float4* __restrict__ chi,
                                                    uint32_t* __restrict__ valid, uint32_t* __restrict__ mergeCounter, uint32_t* __restrict__ childL, uint32_t* __restrict__ childR,
                                                    uint32_t* __restrict__ parentOfInner, uint32_t* __restrict__ parentOfLeaf, float4* __restrict__ nodeLo, float4* __restrict__ nodeHi)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= m)
    return;
  const uint32_t j = nn[i];
  if(j == BVH_NONE || nn[j] != i)
  {
    valid[i] = 1u;
    return;
  }
  if(i > j)
  {
    valid[i] = 0u;
    return;
  }
  const uint32_t id = numInner - 1u - atomicAdd(mergeCounter, 1u);
  const uint32_t a = cid[i], b = cid[j];
  childL[id] = a;
  childR[id] = b;
  if(a & BVH_LEAF) parentOfLeaf[a & ~BVH_LEAF] = id; else parentOfInner[a] = id;
  if(b & BVH_LEAF) parentOfLeaf[b & ~BVH_LEAF] = id; else parentOfInner[b] = id;
  const float4 alo = clo[i], ahi = chi[i], blo = clo[j], bhi = chi[j];
  const float4 lo  = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), fmaxf(alo.w, blo.w));  // .w: subtree holds non-opaque triangles
  const float4 hi  = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.f);
  nodeLo[id] = lo;
  nodeHi[id] = hi;
  cid[i]     = id;
  clo[i]     = lo;
  chi[i]     = hi;
  valid[i]   = 1u;
}
// compaction of the surviving clusters: block-local exclusive scan + block totals, scan of the totals by one block, scatter
__global__ void __launch_bounds__(1024) k_ploc_scan_blocks(uint32_t m, const uint32_t* __restrict__ valid, uint32_t* __restrict__ pos, uint32_t* __restrict__ blockSum)
{
  __shared__ uint32_t sh[1024];
  const uint32_t      i = blockIdx.x * 1024u + threadIdx.x;
  const uint32_t      v = i < m ? valid[i] : 0u;
  sh[threadIdx.x]       = v;
  __syncthreads();
  for(uint32_t off = 1; off < 1024u; off <<= 1)
  {
    uint32_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  if(i < m)
    pos[i] = sh[threadIdx.x] - v;
  if(threadIdx.x == 1023u)
    blockSum[blockIdx.x] = sh[1023];
}
__global__ void __launch_bounds__(1024) k_ploc_scan_sums(uint32_t numBlocks, uint32_t* __restrict__ blockSum, uint32_t* __restrict__ total)
{
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  if(threadIdx.x == 0)
    carry = 0;
  __syncthreads();
  for(uint32_t base = 0; base < numBlocks; base += 1024u)
  {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < numBlocks ? blockSum[i] : 0u;
    sh[threadIdx.x]  = v;
    __syncthreads();
    for(uint32_t off = 1; off < 1024u; off <<= 1)
    {
      uint32_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if(i < numBlocks)
      blockSum[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if(threadIdx.x == 1023u)
      carry += sh[1023];
    __syncthreads();
  }
  if(threadIdx.x == 0)
    *total = carry;
}
__global__ void __launch_bounds__(256) k_ploc_compact(uint32_t m, const uint32_t* __restrict__ valid, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ blockSum,
                                                      const uint32_t* __restrict__ cid, const float4* __restrict__ clo, const float4* __restrict__ chi, uint32_t* __restrict__ cid2,
                                                      float4* __restrict__ clo2, float4* __restrict__ chi2)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= m || !valid[i])
    return;
  const uint32_t p = blockSum[i >> 10] + pos[i];
  cid2[p] = cid[i];
  clo2[p] = clo[i];
  chi2[p] = chi[i];
}

// ---- binned SAH on the device: kernels around the per-thread bodies of pt_sahdev.h ----------------------------------------------------
__global__ void k_sd_prims(uint32_t n, const TriRec* __restrict__ tris, float4* __restrict__ plo, float4* __restrict__ phi, uint32_t* __restrict__ idx, uint32_t* __restrict__ primWork,
                           uint32_t rootWork)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  sd_prim(i, tris, plo, phi);
  idx[i]      = i;
  primWork[i] = rootWork;
}
// forest builds: every primitive starts in the work item of ITS root (TriRec::e1n.w = the instance = the root), SD_NONE for the roots that go
// straight to the small-node list; and every root's parent is "none"
__global__ void k_forest_prim_work(uint32_t n, const TriRec* __restrict__ tris, const uint32_t* __restrict__ rootWork, uint32_t* __restrict__ primWork)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n)
    primWork[i] = rootWork[__float_as_uint(tris[i].e1n.w)];
}
__global__ void k_forest_roots(uint32_t numRoots, const uint32_t* __restrict__ rootNode, uint32_t* __restrict__ parI)
{
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if(r < numRoots)
    parI[rootNode[r]] = BVH_NONE;
}
// the leaf records of a forest of BLASes in vertex form (k_blas_vertex_form with the mesh looked up through the record's instance = root)
__global__ void k_forest_vertex_form(uint32_t n, TriRec* __restrict__ tris, const InstanceRec* __restrict__ pseudo, const float4* __restrict__ vertices, const uint32_t* __restrict__ indices)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const InstanceRec& I = pseudo[__float_as_uint(tris[i].e1n.w)];
  const uint32_t     k = __float_as_uint(tris[i].e2p.w);
  const uint32_t*    t = indices + I.firstIndex + 3 * size_t(k);
  const f3           v0 = load_pos(vertices, I.vertexOffset + t[0]), v1 = load_pos(vertices, I.vertexOffset + t[1]), v2 = load_pos(vertices, I.vertexOffset + t[2]);
  TriRec             r;
  r.p0w   = make_float4(v0.x, v0.y, v0.z, __uint_as_float(k));
  r.e1n   = make_float4(v1.x, v1.y, v1.z, 0.f);
  r.e2p   = make_float4(v2.x, v2.y, v2.z, 0.f);
  tris[i] = r;
}
__global__ void k_sd_init_bins(size_t nBins, uint32_t* __restrict__ binCnt, uint32_t* __restrict__ binBox)
{
  size_t b = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if(b >= nBins)
    return;
  binCnt[b] = 0u;
  for(int q = 0; q < 6; ++q)
    binBox[b * 6 + q] = q < 3 ? SD_ORD_PLUS_INF : SD_ORD_MINUS_INF;
}
// The three per-triangle passes below do what sd_cbounds / sd_bin / sd_partition (pt_sahdev.h, the plain bodies the host emulation runs)
// specify, with the atomics aggregated: near the top of the tree every triangle of a level targets the same few addresses, and same-address
// atomics serialise at ~11 ns each on this chip (269 k triangles x 6 min/max = 18 ms for level 0 alone when issued per lane).
// Lanes of a wave are grouped by the node they belong to (positions are contiguous per node, so a wave holds one or a few groups).
template <class F>
PT_DEV void sd_wave_groups(uint32_t key, bool active, F f)
{
  unsigned long long rem = __ballot(active);
  while(rem)
  {
    const int                leader = __ffsll((long long)rem) - 1;
    const uint32_t           k0     = __shfl(key, leader);
    const unsigned long long m      = __ballot(active && key == k0);
    f(k0, m, leader);
    rem &= ~m;
  }
}
__global__ void __launch_bounds__(256) k_sd_cbounds(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ primWork, SdWork* work, const float4* __restrict__ plo,
                                                    const float4* __restrict__ phi)
{
  const uint32_t pos    = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t w      = pos < n ? primWork[pos] : SD_NONE;
  const bool     active = w != SD_NONE;
  float          c[3]   = {0.f, 0.f, 0.f};
  if(active)
  {
    const uint32_t p  = idx[pos];
    const float4   lo = plo[p], hi = phi[p];
    c[0] = 0.5f * (lo.x + hi.x); c[1] = 0.5f * (lo.y + hi.y); c[2] = 0.5f * (lo.z + hi.z);
  }
  const int lane = threadIdx.x & 63;
  sd_wave_groups(w, active, [&](uint32_t w0, unsigned long long m, int leader) {
    const bool in = (m >> lane) & 1ull;
    float      mn[3], mx[3];
    for(int a = 0; a < 3; ++a)
    {
      mn[a] = in ? c[a] : FLT_MAX;
      mx[a] = in ? c[a] : -FLT_MAX;
      for(int off = 32; off > 0; off >>= 1)
      {
        mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
        mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
      }
    }
    if(lane == leader)
      for(int a = 0; a < 3; ++a)
      {
        atomicMin(&work[w0].cbLo[a], sd_order(mn[a]));
        atomicMax(&work[w0].cbHi[a], sd_order(mx[a]));
      }
  });
}
__global__ void __launch_bounds__(256) k_sd_bin(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ primWork, const SdWork* __restrict__ work,
                                                const float4* __restrict__ plo, const float4* __restrict__ phi, uint32_t* binCnt, uint32_t* binBox)
{
  __shared__ uint32_t lCnt[3 * SD_BINS], lBox[3 * SD_BINS * 6];
  __shared__ uint32_t sW;
  const uint32_t      pos = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t      w   = pos < n ? primWork[pos] : SD_NONE;
  if(threadIdx.x == 0)
    sW = w;
  for(int i = threadIdx.x; i < 3 * SD_BINS; i += blockDim.x)
    lCnt[i] = 0u;
  for(int i = threadIdx.x; i < 3 * SD_BINS * 6; i += blockDim.x)
    lBox[i] = (i % 6) < 3 ? SD_ORD_PLUS_INF : SD_ORD_MINUS_INF;
  __syncthreads();
  const uint32_t w0 = sW;
  // a block whose triangles all belong to one node (every block near the top of the tree) bins into LDS and flushes once
  const bool uniform = __syncthreads_and((pos >= n || w == w0) ? 1 : 0) && w0 != SD_NONE;
  if(!uniform)
  {
    if(pos < n)
      sd_bin(pos, idx, primWork, work, plo, phi, binCnt, binBox);
    return;
  }
  if(pos < n)
  {
    const uint32_t p  = idx[pos];
    const float4   lo = plo[p], hi = phi[p];
    const float    l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    for(int a = 0; a < 3; ++a)
    {
      const float cl = sd_unorder(work[w0].cbLo[a]), ch = sd_unorder(work[w0].cbHi[a]);
      const float ext = ch - cl;
      if(!(ext > 0.f))
        continue;
      const int b = a * SD_BINS + sd_bin_of(0.5f * (l[a] + h[a]), cl, float(SD_BINS) / ext);
      atomicAdd(&lCnt[b], 1u);
      for(int q = 0; q < 3; ++q)
      {
        atomicMin(&lBox[b * 6 + q], sd_order(l[q]));
        atomicMax(&lBox[b * 6 + 3 + q], sd_order(h[q]));
      }
    }
  }
  __syncthreads();
  for(int b = threadIdx.x; b < 3 * SD_BINS; b += blockDim.x)
    if(lCnt[b])
    {
      const size_t g = size_t(w0) * 3 * SD_BINS + b;
      atomicAdd(&binCnt[g], lCnt[b]);
      for(int q = 0; q < 3; ++q)
      {
        atomicMin(&binBox[g * 6 + q], lBox[b * 6 + q]);
        atomicMax(&binBox[g * 6 + 3 + q], lBox[b * 6 + 3 + q]);
      }
    }
}
__global__ void __launch_bounds__(256) k_sd_partition(uint32_t n, const uint32_t* __restrict__ idxIn, const uint32_t* __restrict__ primWorkIn, SdWork* work, SdWork* next,
                                                      const float4* __restrict__ plo, const float4* __restrict__ phi, uint32_t* __restrict__ idxOut, uint32_t* __restrict__ primWorkOut)
{
  const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  const bool     inRange = pos < n;
  const uint32_t w = inRange ? primWorkIn[pos] : SD_NONE;
  const uint32_t p = inRange ? idxIn[pos] : 0u;
  if(inRange && w == SD_NONE)
  {  // finished ranges keep their place
    idxOut[pos]      = p;
    primWorkOut[pos] = SD_NONE;
  }
  const bool active = w != SD_NONE;
  bool       left   = false;
  float      c[3]   = {0.f, 0.f, 0.f};
  if(active)
  {
    const SdWork& W  = work[w];
    const float4  lo = plo[p], hi = phi[p];
    c[0] = 0.5f * (lo.x + hi.x); c[1] = 0.5f * (lo.y + hi.y); c[2] = 0.5f * (lo.z + hi.z);
    left = W.axis < 0 ? pos < W.first + W.nl : sd_bin_of(c[W.axis], W.lo, W.scale) < W.kSplit;
  }
  const int                lane = threadIdx.x & 63;
  const unsigned long long lt   = (1ull << lane) - 1ull;
  sd_wave_groups(w, active, [&](uint32_t w0, unsigned long long m, int leader) {
    const bool               in = (m >> lane) & 1ull;
    const unsigned long long mL = __ballot(in && left), mR = __ballot(in && !left);
    uint32_t                 baseL = 0, baseR = 0;
    if(lane == leader)
    {
      if(mL) baseL = atomicAdd(&work[w0].curL, (uint32_t)__popcll(mL));
      if(mR) baseR = atomicAdd(&work[w0].curR, (uint32_t)__popcll(mR));
    }
    baseL = __shfl(baseL, leader);
    baseR = __shfl(baseR, leader);
    const SdWork& W = work[w0];
    if(in)
    {
      const uint32_t dst = left ? W.first + baseL + (uint32_t)__popcll(mL & lt) : W.first + W.nl + baseR + (uint32_t)__popcll(mR & lt);
      idxOut[dst]        = p;
      primWorkOut[dst]   = left ? W.leftW : W.rightW;
    }
    // centroid bounds of the two children (what sd_cbounds would compute for them on the next level), reduced over the wave first
    for(int side = 0; side < 2; ++side)
    {
      const uint32_t           cw = side == 0 ? W.leftW : W.rightW;
      const unsigned long long ms = side == 0 ? mL : mR;
      if(cw == SD_NONE || ms == 0ull)
        continue;
      const bool mine = (ms >> lane) & 1ull;
      float      mn[3], mx[3];
      for(int a = 0; a < 3; ++a)
      {
        mn[a] = mine ? c[a] : FLT_MAX;
        mx[a] = mine ? c[a] : -FLT_MAX;
        for(int off = 32; off > 0; off >>= 1)
        {
          mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
          mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
      }
      if(lane == leader)
        for(int a = 0; a < 3; ++a)
        {
          atomicMin(&next[cw].cbLo[a], sd_order(mn[a]));
          atomicMax(&next[cw].cbHi[a], sd_order(mx[a]));
        }
    }
  });
}
__global__ void k_sd_split(uint32_t nActive, SdWork* work, const uint32_t* __restrict__ binCnt, const uint32_t* __restrict__ binBox, SdLists L, uint32_t* childL, uint32_t* childR,
                           uint32_t* parI, uint32_t* parL)
{
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if(w < nActive)
    sd_split(w, work, binCnt, binBox, L, childL, childR, parI, parL);
}
__global__ void k_sd_small(uint32_t nSmall, const SdWork* __restrict__ small, uint32_t* idx, const float4* __restrict__ plo, const float4* __restrict__ phi, uint32_t* childL, uint32_t* childR,
                           uint32_t* parI, uint32_t* parL)
{
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if(s < nSmall)
    sd_small(s, small, idx, plo, phi, childL, childR, parI, parL);
}

// ---- tree rotations (Kensler 2008), bottom-up like k_refit ---------------------------------------------------------------
// At inner node X with children (L, R): if L is inner with children (La, Lb), exchanging R with La (or Lb) leaves X's box unchanged and
// replaces L's box by union(R, Lb) (or union(R, La)); symmetric for R.  The exchange with the largest reduction of the child's surface area
// is applied.  One thread owns X and its two children when it gets there (second arrival, subtrees below are finished, nothing above has
// started), so the pointer updates need no further synchronisation than k_refit's hand-off.
__global__ void k_rotate(int n, uint32_t* childL, uint32_t* childR, uint32_t* parentOfInner, uint32_t* parentOfLeaf, const float4* __restrict__ leafLo,
                         const float4* __restrict__ leafHi, float4* nodeLo, float4* nodeHi, unsigned int* arrive)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  auto lo_of = [&](uint32_t c) { return (c & BVH_LEAF) ? leafLo[c & ~BVH_LEAF] : nodeLo[c]; };
  auto hi_of = [&](uint32_t c) { return (c & BVH_LEAF) ? leafHi[c & ~BVH_LEAF] : nodeHi[c]; };
  auto set_parent = [&](uint32_t c, uint32_t p) {
    if(c & BVH_LEAF) parentOfLeaf[c & ~BVH_LEAF] = p; else parentOfInner[c] = p;
  };
  uint32_t cur = parentOfLeaf[i];
  while(cur != BVH_NONE)
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int prev = __hip_atomic_fetch_add(&arrive[cur], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(prev == 0)
      return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t L = childL[cur], R = childR[cur];
    float          bestGain = 0.0f;
    int            best     = -1;  // 0: R <-> La, 1: R <-> Lb, 2: L <-> Ra, 3: L <-> Rb
    if(!(L & BVH_LEAF))
    {
      const uint32_t La = childL[L], Lb = childR[L];
      const float    aL = half_area(nodeLo[L], nodeHi[L]);
      const float    g0 = aL - union_half_area(lo_of(R), hi_of(R), lo_of(Lb), hi_of(Lb));
      const float    g1 = aL - union_half_area(lo_of(R), hi_of(R), lo_of(La), hi_of(La));
      if(g0 > bestGain) { bestGain = g0; best = 0; }
      if(g1 > bestGain) { bestGain = g1; best = 1; }
    }
    if(!(R & BVH_LEAF))
    {
      const uint32_t Ra = childL[R], Rb = childR[R];
      const float    aR = half_area(nodeLo[R], nodeHi[R]);
      const float    g2 = aR - union_half_area(lo_of(L), hi_of(L), lo_of(Rb), hi_of(Rb));
      const float    g3 = aR - union_half_area(lo_of(L), hi_of(L), lo_of(Ra), hi_of(Ra));
      if(g2 > bestGain) { bestGain = g2; best = 2; }
      if(g3 > bestGain) { bestGain = g3; best = 3; }
    }
    if(best >= 0)
    {
      const bool     left  = best < 2;           // the inner child that is rebuilt
      const uint32_t C     = left ? L : R;       // ... it
      const uint32_t S     = left ? R : L;       // the sibling that moves down
      const bool     first = (best & 1) == 0;    // the grandchild that moves up: C's first (a) or second (b) child
      const uint32_t up    = first ? childL[C] : childR[C];
      const uint32_t stay  = first ? childR[C] : childL[C];
      childL[C] = S;
      childR[C] = stay;
      const float4 slo = lo_of(S), shi = hi_of(S), tlo = lo_of(stay), thi = hi_of(stay);
      nodeLo[C] = make_float4(fminf(slo.x, tlo.x), fminf(slo.y, tlo.y), fminf(slo.z, tlo.z), fmaxf(slo.w, tlo.w));
      nodeHi[C] = make_float4(fmaxf(shi.x, thi.x), fmaxf(shi.y, thi.y), fmaxf(shi.z, thi.z), 0.f);
      set_parent(S, C);
      set_parent(up, uint32_t(cur));
      if(left) { childL[cur] = C; childR[cur] = up; } else { childL[cur] = up; childR[cur] = C; }
    }
    cur = parentOfInner[cur];
  }
}

// leaf reference of sorted slot `i`: non-opaque triangles are tagged so that traversal fetches their AlphaRec up front
PT_DEV uint32_t leaf_ref(const TriRec* __restrict__ tris, uint32_t leaf, uint32_t leafOffset = 0u)
{
  const uint32_t slot  = leaf & ~BVH_LEAF;
  const uint32_t flags = __float_as_uint(tris[slot].p0w.w) >> 29;
  return BVH_LEAF | (slot + leafOffset) | ((flags & TRI_OPAQUE) ? 0u : BVH_ALPHA);
}

__global__ void k_emit(int numInner, const uint32_t* __restrict__ childL, const uint32_t* __restrict__ childR, const float4* __restrict__ leafLo,
                       const float4* __restrict__ leafHi, const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi, const TriRec* __restrict__ tris, BvhNode* __restrict__ out,
                       uint32_t leafOffset = 0u)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= numInner)
    return;
  uint32_t l = childL[i], r = childR[i];
  float4   llo = (l & BVH_LEAF) ? leafLo[l & ~BVH_LEAF] : nodeLo[l];
  float4   lhi = (l & BVH_LEAF) ? leafHi[l & ~BVH_LEAF] : nodeHi[l];
  float4   rlo = (r & BVH_LEAF) ? leafLo[r & ~BVH_LEAF] : nodeLo[r];
  float4   rhi = (r & BVH_LEAF) ? leafHi[r & ~BVH_LEAF] : nodeHi[r];
  BvhNode  nd;
  nd.a   = make_float4(llo.x, llo.y, llo.z, lhi.x);
  nd.b   = make_float4(lhi.y, lhi.z, rlo.x, rlo.y);
  nd.c   = make_float4(rlo.z, rhi.x, rhi.y, rhi.z);
  nd.d   = make_uint4((l & BVH_LEAF) ? leaf_ref(tris, l, leafOffset) : (l | (llo.w > 0.f ? BVH_ALPHA : 0u)), (r & BVH_LEAF) ? leaf_ref(tris, r, leafOffset) : (r | (rlo.w > 0.f ? BVH_ALPHA : 0u)), 0u, 0u);
  out[i] = nd;
}

__global__ void k_single_leaf(const float4* leafLo, const float4* leafHi, const TriRec* tris, BvhNode* out)
{
  BvhNode nd;
  float4  lo = leafLo[0], hi = leafHi[0];
  nd.a   = make_float4(lo.x, lo.y, lo.z, hi.x);
  nd.b   = make_float4(hi.y, hi.z, FLT_MAX, FLT_MAX);
  nd.c   = make_float4(FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
  nd.d   = make_uint4(leaf_ref(tris, BVH_LEAF), BVH_NONE, 0u, 0u);
  out[0] = nd;
}


// ---- two-level structure: kernels around pt_accel_build (reference: src/accelstruct.cpp:110-162) ---------------------------------------
// centroids + their bounds of ready-made primitive records (the TLAS's instance boxes)
__global__ void k_proxy_centroids(uint32_t n, const TriRec* __restrict__ prox, float4* __restrict__ cen, uint32_t* __restrict__ bounds)
{
  const uint32_t i     = blockIdx.x * blockDim.x + threadIdx.x;
  const bool     valid = i < n;
  f3             c     = f3{0, 0, 0};
  if(valid)
  {
    const TriRec r = prox[i];
    c      = xyz(r.p0w) + xyz(r.e1n) * 0.5f;
    cen[i] = make_float4(c.x, c.y, c.z, 0.f);
  }
  float mn[3] = {valid ? c.x : FLT_MAX, valid ? c.y : FLT_MAX, valid ? c.z : FLT_MAX};
  float mx[3] = {valid ? c.x : -FLT_MAX, valid ? c.y : -FLT_MAX, valid ? c.z : -FLT_MAX};
  for(int off = 32; off > 0; off >>= 1)
    for(int a = 0; a < 3; ++a)
    {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
    }
  if((threadIdx.x & 63) == 0)
    for(int a = 0; a < 3; ++a)
    {
      atomicMin(&bounds[a], order_bits(mn[a]));
      atomicMax(&bounds[3 + a], order_bits(mx[a]));
    }
}

// BLAS leaf records from the builder's edge form (p0, e1, e2 under the identity transform) to the vertex form the two-level walk transforms
// per instance: the three OBJECT-space positions exactly as the vertex buffer holds them, p0w.w = primitive index
__global__ void k_blas_vertex_form(uint32_t n, TriRec* __restrict__ tris, const float4* __restrict__ vertices, const uint32_t* __restrict__ indices, uint32_t vertexOffset,
                                   uint32_t firstIndex)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const uint32_t  k = __float_as_uint(tris[i].e2p.w);
  const uint32_t* t = indices + firstIndex + 3 * size_t(k);
  const f3        v0 = load_pos(vertices, vertexOffset + t[0]), v1 = load_pos(vertices, vertexOffset + t[1]), v2 = load_pos(vertices, vertexOffset + t[2]);
  TriRec          r;
  r.p0w   = make_float4(v0.x, v0.y, v0.z, __uint_as_float(k));
  r.e1n   = make_float4(v1.x, v1.y, v1.z, 0.f);
  r.e2p   = make_float4(v2.x, v2.y, v2.z, 0.f);
  tris[i] = r;
}
// child references of a BLAS from local to global indices (nodes: + nodeBase, leaf slots: + slotBase), tags kept
__global__ void k_blas_rebase(uint32_t numWide, WideNode* __restrict__ wide, uint32_t nodeBase, uint32_t slotBase)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= numWide)
    return;
  for(int q = 0; q < PT_WIDE_Q; ++q)
  {
    uint32_t* ch = &wide[i].child[q].x;
    for(int k = 0; k < 4; ++k)
      if(ch[k] != BVH_NONE)
        ch[k] = (ch[k] & ~BVH_SLOT_MASK) | ((ch[k] & BVH_SLOT_MASK) + ((ch[k] & BVH_LEAF) ? slotBase : nodeBase));
  }
}

// World box of an instance = exact bounds of its T1 world triangles (the same xform_point the leaf test applies, so the box encloses every
// triangle the walk can test inside it), written as the "diagonal" primitive record the builder takes.  One block per active instance.
__global__ void __launch_bounds__(256) k_instance_proxies(const uint32_t* __restrict__ active, const InstanceRec* __restrict__ inst, const float4* __restrict__ vertices,
                                                          const uint32_t* __restrict__ indices, TriRec* __restrict__ out)
{
  __shared__ float red[6][4];
  const uint32_t     id = active[blockIdx.x];
  const InstanceRec& I  = inst[id];
  const Affine       M  = I.objectToWorld;
  float              mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for(uint32_t j = threadIdx.x; j < 3u * I.triCount; j += blockDim.x)
  {
    const f3 p = xform_point(M, load_pos(vertices, I.vertexOffset + indices[I.firstIndex + j]));
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
  for(int off = 32; off > 0; off >>= 1)
    for(int a = 0; a < 3; ++a)
    {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
    }
  const int wave = threadIdx.x >> 6;
  if((threadIdx.x & 63) == 0)
    for(int a = 0; a < 3; ++a)
    {
      red[a][wave]     = mn[a];
      red[3 + a][wave] = mx[a];
    }
  __syncthreads();
  if(threadIdx.x == 0)
  {
    for(int a = 0; a < 3; ++a)
      for(int w = 1; w < 4; ++w)
      {
        red[a][0]     = fminf(red[a][0], red[a][w]);
        red[3 + a][0] = fmaxf(red[3 + a][0], red[3 + a][w]);
      }
    TriRec r;  // box = bounds of {p0, p0 + e1, p0 + e2}; p0 + e1 may miss the upper corner by an ulp, which the leaf padding of tri_box covers
    r.p0w = make_float4(red[0][0], red[1][0], red[2][0], __uint_as_float(id | (I.flags << 29)));
    r.e1n = make_float4(red[3][0] - red[0][0], red[4][0] - red[1][0], red[5][0] - red[2][0], 0.f);
    r.e2p = make_float4(0.f, 0.f, 0.f, 0.f);
    out[blockIdx.x] = r;
  }
}
__global__ void k_tlas_leaves(uint32_t n, const TriRec* __restrict__ leafOrder, const InstanceRec* __restrict__ inst, const uint32_t* __restrict__ instNodeBase,
                              const float* __restrict__ instPad, TlasLeaf* __restrict__ out, uint32_t mergedNodeBase)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const uint32_t id = __float_as_uint(leafOrder[i].p0w.w) & TRI_INDEX_MASK;
  TlasLeaf       l;
  if(id == TRI_INDEX_MASK)
  {  // the proxy of the merged world-space structure (pt_tlas_build): no transform, no padding
    l.inst     = PT_INST_MERGED;
    l.nodeBase = mergedNodeBase;
    l.wflags   = 0;
    l._pad0    = 0;
    l.padC0 = l.padC1 = 0.f;
    l._pad1[0] = l._pad1[1] = 0;
    out[i]     = l;
    return;
  }
  l.inst     = id;
  l.nodeBase = instNodeBase[id];
  l.wflags   = inst[id].triBase | (inst[id].flags << 29);
  l._pad0    = 0;
  l.padC0    = instPad[2 * size_t(id)];
  l.padC1    = instPad[2 * size_t(id) + 1];
  l._pad1[0] = l._pad1[1] = 0;
  out[i]     = l;
}

// ---- collapse BVH2 -> wide BVH (level-synchronous; one thread per wide node) -------------------------------------
struct CollapseItem {
  uint32_t b2;    // binary node to expand
  uint32_t wide;  // wide node it becomes
  uint32_t root;  // forest builds: the root it belongs to (its wide nodes come out of that root's range)
};
PT_DEV float half_area(float4 lo, float4 hi)
{
  float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  return dx * dy + dy * dz + dz * dx;
}
__global__ void k_collapse(const BvhNode* __restrict__ b2, const CollapseItem* __restrict__ qin, uint32_t nIn, CollapseItem* __restrict__ qout, uint32_t* counters /* [0]=nOut [1]=wideCount */,
                           WideNode* __restrict__ out, uint32_t* __restrict__ rootCount = nullptr, const uint32_t* __restrict__ rootBase = nullptr)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= nIn)
    return;
  const CollapseItem it = qin[i];
  uint32_t           id[PT_BVH_WIDTH];
  float4             lo[PT_BVH_WIDTH], hi[PT_BVH_WIDTH];
  int                n = 0;
  auto               push_children = [&](uint32_t node) {
    const BvhNode nd = b2[node & BVH_SLOT_MASK];
    id[n] = nd.d.x; lo[n] = make_float4(nd.a.x, nd.a.y, nd.a.z, 0.f); hi[n] = make_float4(nd.a.w, nd.b.x, nd.b.y, 0.f); ++n;
    if(nd.d.y != BVH_NONE)
    {
      id[n] = nd.d.y; lo[n] = make_float4(nd.b.z, nd.b.w, nd.c.x, 0.f); hi[n] = make_float4(nd.c.y, nd.c.z, nd.c.w, 0.f); ++n;
    }
  };
  push_children(it.b2);
  while(n < PT_BVH_WIDTH)
  {
    int   best = -1;
    float bestA = -1.f;
    for(int k = 0; k < n; ++k)
      if(!(id[k] & BVH_LEAF))
      {
        float a = half_area(lo[k], hi[k]);
        if(a > bestA)
        {
          bestA = a;
          best  = k;
        }
      }
    if(best < 0)
      break;
    const uint32_t node = id[best];
    // replace slot `best` by the first child, append the second
    id[best] = id[n - 1]; lo[best] = lo[n - 1]; hi[best] = hi[n - 1];
    --n;
    push_children(node);
  }
  WideNode w;
  for(int q = 0; q < PT_WIDE_Q; ++q)
  {
    float* mnx = &w.minx[q].x; float* mny = &w.miny[q].x; float* mnz = &w.minz[q].x;
    float* mxx = &w.maxx[q].x; float* mxy = &w.maxy[q].x; float* mxz = &w.maxz[q].x;
    uint32_t* ch = &w.child[q].x;
    for(int k = 0; k < 4; ++k)
    {
      int c = q * 4 + k;
      if(c < n)
      {
        mnx[k] = lo[c].x; mny[k] = lo[c].y; mnz[k] = lo[c].z; mxx[k] = hi[c].x; mxy[k] = hi[c].y; mxz[k] = hi[c].z;
        if(id[c] & BVH_LEAF)
          ch[k] = id[c];
        else
        {
          uint32_t wid = rootCount ? rootBase[it.root] + atomicAdd(&rootCount[it.root], 1u) : atomicAdd(&counters[1], 1u);
          uint32_t qi  = atomicAdd(&counters[0], 1u);
          qout[qi]     = CollapseItem{id[c] & BVH_SLOT_MASK, wid, it.root};
          ch[k]        = wid | (id[c] & BVH_ALPHA);  // inner reference: wide node index + "subtree holds non-opaque triangles"
        }
      }
      else
      {
        mnx[k] = mny[k] = mnz[k] = FLT_MAX;
        mxx[k] = mxy[k] = mxz[k] = -FLT_MAX;
        ch[k]                    = BVH_NONE;
      }
    }
    w.pad[q] = make_uint4(0, 0, 0, 0);
  }
  out[it.wide] = w;
}

}  // namespace

#define HIPCHK(x)                                                                                              \
  do                                                                                                           \
  {                                                                                                            \
    hipError_t e_ = (x);                                                                                       \
    if(e_ != hipSuccess)                                                                                       \
    {                                                                                                          \
      snprintf(err, errLen, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__);           \
      goto fail;                                                                                               \
    }                                                                                                          \
  } while(0)

// Builds TriRec[numTris] (leaf order) and BvhNode[max(1,numTris-1)] into caller-allocated device memory.
// dProxies (may be null): the primitives are given as ready-made records instead of (instance, triangle) pairs -- the TLAS of the two-level
// structure is built over one "diagonal" record per instance (p0 = box min, e1 = box extent, e2 = 0: its bounding box is the instance's box).
// scratch (may be null): temporaries come out of the caller's arena instead of one device allocation each (a scene of hundreds of BLASes).
int pt_accel_build(hipStream_t stream, const PtTuning& tune, const InstanceRec* dInst, uint32_t numInst, const float4* dVertices, const uint32_t* dIndices, uint32_t numTris,
                   TriRec* dTrisOut, AlphaRec* dAlphaOut, BvhNode* dNodesOut, WideNode* dWideOut, uint32_t* numWideOut, char* err, size_t errLen, const TriRec* dProxies,
                   PtScratch* scratch, const PtForest* forest)
{
  *numWideOut = 0;
  if(numTris == 0)
    return 0;
  PtScratch  localScratch;
  PtScratch& sc = scratch ? *scratch : localScratch;
  const uint32_t n          = numTris;
  const uint32_t sortBlocks = (n + SORT_ITEMS - 1) / SORT_ITEMS;
  const int      B          = 256;
  const uint32_t G          = (n + B - 1) / B;

  TriRec*   dUnsorted = nullptr;
  AlphaRec* dAlphaUnsorted = nullptr;
  float4 *  dCen = nullptr, *dLeafLo = nullptr, *dLeafHi = nullptr, *dNodeLo = nullptr, *dNodeHi = nullptr;
  uint32_t *dKeysA = nullptr, *dKeysB = nullptr, *dValsA = nullptr, *dValsB = nullptr, *dHist = nullptr, *dBounds = nullptr;
  uint32_t *dChildL = nullptr, *dChildR = nullptr, *dParI = nullptr, *dParL = nullptr;
  unsigned* dArrive = nullptr;
  bool      sah     = false, ploc = false, sahDev = false;
  uint32_t  initBounds[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};

  HIPCHK(sc.get((void**)&dUnsorted, sizeof(TriRec) * size_t(n)));
  HIPCHK(sc.get((void**)&dAlphaUnsorted, sizeof(AlphaRec) * size_t(n)));
  HIPCHK(sc.get((void**)&dCen, sizeof(float4) * size_t(n)));
  HIPCHK(sc.get((void**)&dLeafLo, sizeof(float4) * size_t(n)));
  HIPCHK(sc.get((void**)&dLeafHi, sizeof(float4) * size_t(n)));
  HIPCHK(sc.get((void**)&dNodeLo, sizeof(float4) * size_t(n)));
  HIPCHK(sc.get((void**)&dNodeHi, sizeof(float4) * size_t(n)));
  HIPCHK(sc.get((void**)&dKeysA, 4 * size_t(n)));
  HIPCHK(sc.get((void**)&dKeysB, 4 * size_t(n)));
  HIPCHK(sc.get((void**)&dValsA, 4 * size_t(n)));
  HIPCHK(sc.get((void**)&dValsB, 4 * size_t(n)));
  HIPCHK(sc.get((void**)&dHist, 4 * size_t(256) * sortBlocks));
  HIPCHK(sc.get((void**)&dBounds, 4 * 6));
  HIPCHK(sc.get((void**)&dChildL, 4 * size_t(n)));
  HIPCHK(sc.get((void**)&dChildR, 4 * size_t(n)));
  HIPCHK(sc.get((void**)&dParI, 4 * size_t(n)));
  HIPCHK(sc.get((void**)&dParL, 4 * size_t(n)));
  HIPCHK(sc.get((void**)&dArrive, 4 * size_t(n)));
  HIPCHK(hipMemcpyAsync(dBounds, initBounds, sizeof(initBounds), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemsetAsync(dArrive, 0, 4 * size_t(n), stream));

  if(dProxies)
  {
    HIPCHK(hipMemcpyAsync(dUnsorted, dProxies, sizeof(TriRec) * size_t(n), hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemsetAsync(dAlphaUnsorted, 0, sizeof(AlphaRec) * size_t(n), stream));
    k_proxy_centroids<<<G, B, 0, stream>>>(n, dProxies, dCen, dBounds);
  }
  else
    k_world_tris<<<G, B, 0, stream>>>(n, dInst, numInst, dVertices, dIndices, dUnsorted, dAlphaUnsorted, dCen, dBounds);
  sah = tune.sahBuild == 1 && n >= 2;  // sahBuild: 0 device LBVH (Karras), 1 host SAH topology, 2 device PLOC, 3 device binned SAH (default)
  ploc = tune.sahBuild == 2 && n >= 2;
  sahDev = tune.sahBuild == 3 && n >= 2;
  if(sahDev)
  {
    // ---- device binned SAH (pt_sahdev.h): level-synchronous; the number of open nodes comes back to the host between levels
    float4 *  plo = nullptr, *phi = nullptr;
    uint32_t *idxA = nullptr, *idxB = nullptr, *pwA = nullptr, *pwB = nullptr, *binCnt = nullptr, *binBox = nullptr, *dCounts = nullptr;
    SdWork *  workA = nullptr, *workB = nullptr, *small = nullptr;
    const size_t maxWork = size_t(n) / (SD_SMALL + 1) + 2, maxSmall = size_t(n) / 2 + 2;
    bool      okAlloc = true;
    auto      grab = [&](void** p, size_t bytes) { okAlloc = okAlloc && sc.get(p, bytes) == hipSuccess; };
    grab((void**)&plo, 16 * size_t(n)); grab((void**)&phi, 16 * size_t(n)); grab((void**)&idxA, 4 * size_t(n)); grab((void**)&idxB, 4 * size_t(n)); grab((void**)&pwA, 4 * size_t(n));
    grab((void**)&pwB, 4 * size_t(n)); grab((void**)&workA, sizeof(SdWork) * maxWork); grab((void**)&workB, sizeof(SdWork) * maxWork); grab((void**)&small, sizeof(SdWork) * maxSmall);
    grab((void**)&binCnt, 4 * maxWork * 3 * SD_BINS); grab((void**)&binBox, 4 * maxWork * 3 * SD_BINS * 6); grab((void**)&dCounts, 8);
    bool  done  = false;
    if(okAlloc)
    {
      uint32_t nActive = 0, nSmall = 0;
      SdWork   root;
      sd_init_work(root, 0, n, 0);
      uint32_t zero2[2] = {0u, 0u};
      const uint32_t noneParent = BVH_NONE;
      bool     ok = true;
      if(forest)
      {  // one root per hierarchy: the big ones open the first level's work list, the small ones go to the small-node list
        std::vector<SdWork>   big, little;
        std::vector<uint32_t> rootWork(forest->numRoots), rootNode(forest->numRoots);
        for(uint32_t r = 0; r < forest->numRoots; ++r)
        {
          SdWork w;
          sd_init_work(w, forest->first[r], forest->count[r], forest->first[r]);
          rootNode[r] = forest->first[r];
          if(forest->count[r] <= SD_SMALL)
          {
            rootWork[r] = SD_NONE;
            little.push_back(w);
          }
          else
          {
            rootWork[r] = uint32_t(big.size());
            big.push_back(w);
          }
        }
        nActive  = uint32_t(big.size());
        nSmall   = uint32_t(little.size());
        zero2[1] = nSmall;
        uint32_t *dRootWork = nullptr, *dRootNode = nullptr;
        ok = nActive <= maxWork && nSmall <= maxSmall && sc.get((void**)&dRootWork, 4 * size_t(forest->numRoots)) == hipSuccess && sc.get((void**)&dRootNode, 4 * size_t(forest->numRoots)) == hipSuccess;
        ok = ok && hipMemcpyAsync(dRootWork, rootWork.data(), 4 * rootWork.size(), hipMemcpyHostToDevice, stream) == hipSuccess &&
             hipMemcpyAsync(dRootNode, rootNode.data(), 4 * rootNode.size(), hipMemcpyHostToDevice, stream) == hipSuccess;
        if(ok && nActive)
          ok = hipMemcpyAsync(workA, big.data(), sizeof(SdWork) * big.size(), hipMemcpyHostToDevice, stream) == hipSuccess;
        if(ok && nSmall)
          ok = hipMemcpyAsync(small, little.data(), sizeof(SdWork) * little.size(), hipMemcpyHostToDevice, stream) == hipSuccess;
        // ids nobody owns (one per root: a root of k leaves uses k - 1 of its k ids) must still be readable by k_emit
        ok = ok && hipMemsetAsync(dChildL, 0, 4 * size_t(n), stream) == hipSuccess && hipMemsetAsync(dChildR, 0, 4 * size_t(n), stream) == hipSuccess &&
             hipMemsetAsync(dNodeLo, 0, 16 * size_t(n), stream) == hipSuccess && hipMemsetAsync(dNodeHi, 0, 16 * size_t(n), stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(dCounts, zero2, 8, hipMemcpyHostToDevice, stream) == hipSuccess;
        if(ok)
        {
          k_sd_prims<<<G, B, 0, stream>>>(n, dUnsorted, plo, phi, idxA, pwA, SD_NONE);
          k_forest_prim_work<<<G, B, 0, stream>>>(n, dUnsorted, dRootWork, pwA);
          k_forest_roots<<<(forest->numRoots + 255) / 256, 256, 0, stream>>>(forest->numRoots, dRootNode, dParI);
          ok = hipStreamSynchronize(stream) == hipSuccess;  // the host vectors above die with this scope
        }
      }
      else
      {
      ok = hipMemcpyAsync(dParI, &noneParent, 4, hipMemcpyHostToDevice, stream) == hipSuccess;
      if(n <= SD_SMALL)
      {
        nSmall   = 1;
        zero2[1] = 1;
        ok       = ok && hipMemcpyAsync(small, &root, sizeof(root), hipMemcpyHostToDevice, stream) == hipSuccess;
      }
      else
      {
        nActive = 1;
        ok      = ok && hipMemcpyAsync(workA, &root, sizeof(root), hipMemcpyHostToDevice, stream) == hipSuccess;
      }
      ok = ok && hipMemcpyAsync(dCounts, zero2, 8, hipMemcpyHostToDevice, stream) == hipSuccess;
      k_sd_prims<<<G, B, 0, stream>>>(n, dUnsorted, plo, phi, idxA, pwA, nActive ? 0u : SD_NONE);
      }
      int levels = 0;
      while(nActive && ok && levels < 4096)
      {
        const size_t nBins = size_t(nActive) * 3 * SD_BINS;
        k_sd_init_bins<<<unsigned((nBins + 255) / 256), 256, 0, stream>>>(nBins, binCnt, binBox);
        if(levels == 0)  // deeper levels get their centroid bounds from their parents' partition pass
          k_sd_cbounds<<<G, B, 0, stream>>>(n, idxA, pwA, workA, plo, phi);
        k_sd_bin<<<G, B, 0, stream>>>(n, idxA, pwA, workA, plo, phi, binCnt, binBox);
        (void)hipMemsetAsync(dCounts, 0, 4, stream);  // next-level counter; the small-node counter keeps running
        SdLists L{workB, dCounts, small, dCounts + 1};
        k_sd_split<<<(nActive + 63) / 64, 64, 0, stream>>>(nActive, workA, binCnt, binBox, L, dChildL, dChildR, dParI, dParL);
        k_sd_partition<<<G, B, 0, stream>>>(n, idxA, pwA, workA, workB, plo, phi, idxB, pwB);
        uint32_t counts[2] = {0u, 0u};
        ok = hipMemcpyAsync(counts, dCounts, 8, hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess && counts[0] <= maxWork && counts[1] <= maxSmall;
        nActive = counts[0];
        nSmall  = counts[1];
        std::swap(idxA, idxB); std::swap(pwA, pwB); std::swap(workA, workB);
        ++levels;
      }
      if(ok && nActive == 0)
      {
        if(nSmall)
          k_sd_small<<<(nSmall + 63) / 64, 64, 0, stream>>>(nSmall, small, idxA, plo, phi, dChildL, dChildR, dParI, dParL);
        ok   = hipMemcpyAsync(dValsA, idxA, 4 * size_t(n), hipMemcpyDeviceToDevice, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess && hipGetLastError() == hipSuccess;
        done = ok;
      }
    }
    (void)hipGetLastError();
    if(!done)
    {
      snprintf(err, errLen, "device SAH build failed (out of memory or a kernel error)");
      goto fail;
    }
    sah = true;  // from here on like the host SAH builder: topology + leaf order are given
  }
  else
  if(sah)
  {
    // cross-check build: SAH topology on the host from the world-space triangles (pt_sah.hip); boxes stay on the device
    std::vector<TriRec>   hTris(n);
    std::vector<uint32_t> hVals(n), hL(n), hR(n), hPI(n), hPL(n);
    HIPCHK(hipMemcpyAsync(hTris.data(), dUnsorted, sizeof(TriRec) * size_t(n), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    pt_sah_topology(n, hTris.data(), hVals.data(), hL.data(), hR.data(), hPI.data(), hPL.data());
    HIPCHK(hipMemcpyAsync(dValsA, hVals.data(), 4 * size_t(n), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dChildL, hL.data(), 4 * size_t(n), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dChildR, hR.data(), 4 * size_t(n), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dParI, hPI.data(), 4 * size_t(n), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dParL, hPL.data(), 4 * size_t(n), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));  // the host vectors die at the end of this scope
  }
  else
  {
    k_morton<<<G, B, 0, stream>>>(n, dCen, dBounds, dKeysA, dValsA);
    uint32_t *kin = dKeysA, *kout = dKeysB, *vin = dValsA, *vout = dValsB;
    for(int pass = 0; pass < 4; ++pass)
    {
      int shift = pass * 8;
      k_sort_hist<<<sortBlocks, 64, 0, stream>>>(kin, n, shift, dHist, sortBlocks);
      k_sort_scan<<<1, 256, 0, stream>>>(dHist, 256u * sortBlocks);
      k_sort_scatter<<<sortBlocks, 64, 0, stream>>>(kin, vin, kout, vout, n, shift, dHist, sortBlocks);
      std::swap(kin, kout);
      std::swap(vin, vout);
    }
    // after 4 passes the sorted data is back in A
  }
  k_gather<<<G, B, 0, stream>>>(n, dValsA, dUnsorted, dTrisOut, dAlphaUnsorted, dAlphaOut, dLeafLo, dLeafHi);
  if(ploc)
  {
    // PLOC rounds; the cluster count comes back to the host between rounds (the build is not on the timed path).  A round merges at least the
    // globally best pair; if an adversarial input makes the rounds crawl, the radix tree takes over.
    uint32_t *cidA = nullptr, *cidB = nullptr, *dNn = nullptr, *dValid = nullptr, *dPos = nullptr, *dBlockSum = nullptr, *dCnt = nullptr;
    float4 *  cloA = nullptr, *cloB = nullptr, *chiA = nullptr, *chiB = nullptr;
    bool      okAlloc = true;
    auto      grab = [&](void** p, size_t bytes) { okAlloc = okAlloc && sc.get(p, bytes) == hipSuccess; };
    grab((void**)&cidA, 4 * size_t(n)); grab((void**)&cidB, 4 * size_t(n)); grab((void**)&dNn, 4 * size_t(n)); grab((void**)&dValid, 4 * size_t(n)); grab((void**)&dPos, 4 * size_t(n));
    grab((void**)&dBlockSum, 4 * size_t((n + 1023) / 1024 + 1)); grab((void**)&dCnt, 8);
    grab((void**)&cloA, 16 * size_t(n)); grab((void**)&cloB, 16 * size_t(n)); grab((void**)&chiA, 16 * size_t(n)); grab((void**)&chiB, 16 * size_t(n));
    bool done = false;
    if(okAlloc)
    {
      const int radius = PT_PLOC_RADIUS;
      (void)hipMemsetAsync(dCnt, 0, 8, stream);
      k_ploc_init<<<G, B, 0, stream>>>(n, dLeafLo, dLeafHi, cidA, cloA, chiA);
      uint32_t m = n;
      int      rounds = 0, maxRounds = 64;
      for(uint32_t t = n; t > 1; t >>= 1)
        maxRounds += 6;
      bool ok = true;
      while(m > 1 && rounds < maxRounds && ok)
      {
        const uint32_t g = (m + 255) / 256, nb = (m + 1023) / 1024;
        // the top of the tree decides how many subtrees a ray enters: once few clusters are left every cluster considers ALL others
        // (exact agglomerative clustering), not just its Morton neighbourhood
        const int rad = radius;
        k_ploc_nn<<<g, 256, 0, stream>>>(m, rad, cloA, chiA, dNn);
        k_ploc_merge<<<g, 256, 0, stream>>>(m, n - 1, dNn, cidA, cloA, chiA, dValid, dCnt, dChildL, dChildR, dParI, dParL, dNodeLo, dNodeHi);
        k_ploc_scan_blocks<<<nb, 1024, 0, stream>>>(m, dValid, dPos, dBlockSum);
        k_ploc_scan_sums<<<1, 1024, 0, stream>>>(nb, dBlockSum, dCnt + 1);
        k_ploc_compact<<<g, 256, 0, stream>>>(m, dValid, dPos, dBlockSum, cidA, cloA, chiA, cidB, cloB, chiB);
        uint32_t m2 = 0;
        ok = hipMemcpyAsync(&m2, dCnt + 1, 4, hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess && m2 >= 1 && m2 < m;
        m = m2;
        std::swap(cidA, cidB); std::swap(cloA, cloB); std::swap(chiA, chiB);
        ++rounds;
      }
      done = ok && m == 1;
      if(done)
      {
        const uint32_t none = BVH_NONE;
        (void)hipMemcpyAsync(dParI, &none, 4, hipMemcpyHostToDevice, stream);  // the root (node 0) has no parent
      }
    }
    (void)hipGetLastError();
    if(!done)
      ploc = false;  // fall back to the radix tree below
  }
  if(n == 1)
  {
    k_single_leaf<<<1, 1, 0, stream>>>(dLeafLo, dLeafHi, dTrisOut, dNodesOut);
  }
  else
  {
    if(!sah && !ploc)
      k_hierarchy<<<G, B, 0, stream>>>(int(n), dKeysA, dChildL, dChildR, dParI, dParL);
    if(!ploc)  // PLOC wrote the inner boxes while merging
      k_refit<<<G, B, 0, stream>>>(int(n), dChildL, dChildR, dParI, dParL, dLeafLo, dLeafHi, dNodeLo, dNodeHi, dArrive);
    for(int pass = 0; pass < PT_ROTATE_PASSES && !sah; ++pass)
    {
      (void)hipMemsetAsync(dArrive, 0, 4 * size_t(n), stream);
      k_rotate<<<G, B, 0, stream>>>(int(n), dChildL, dChildR, dParI, dParL, dLeafLo, dLeafHi, dNodeLo, dNodeHi, dArrive);
    }
    k_emit<<<(n - 1 + B - 1) / B, B, 0, stream>>>(int(n - 1), dChildL, dChildR, dLeafLo, dLeafHi, dNodeLo, dNodeHi, dTrisOut, dNodesOut, forest ? forest->leafOffset : 0u);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(stream));
  if(forest && !sahDev)
  {
    snprintf(err, errLen, "a forest is built by the device SAH builder only");
    goto fail;
  }

  // ---- collapse to the wide layout, one BVH level per launch (the queue sizes come back to the host between levels;
  // the build is not on the timed path)
  {
    CollapseItem* dQ[2] = {nullptr, nullptr};
    uint32_t*     dCnt  = nullptr;
    HIPCHK(sc.get((void**)&dQ[0], sizeof(CollapseItem) * size_t(n)));
    HIPCHK(sc.get((void**)&dQ[1], sizeof(CollapseItem) * size_t(n)));
    HIPCHK(sc.get((void**)&dCnt, 8));
    CollapseItem first{0u, 0u, 0u};
    uint32_t     cnt[2] = {0u, 1u};  // next-queue size, wide nodes allocated (root = 0)
    uint32_t     nIn = 1;
    uint32_t *   dRootCount = nullptr, *dRootBase = nullptr;
    if(forest)
    {  // every root starts a queue entry of its own: binary root first[r] becomes wide node wideBase[r], its descendants are allocated behind it
      std::vector<CollapseItem> roots(forest->numRoots);
      std::vector<uint32_t>     ones(forest->numRoots, 1u);
      for(uint32_t r = 0; r < forest->numRoots; ++r)
        roots[r] = CollapseItem{forest->first[r], forest->wideBase[r], r};
      HIPCHK(sc.get((void**)&dRootCount, 4 * size_t(forest->numRoots)));
      HIPCHK(sc.get((void**)&dRootBase, 4 * size_t(forest->numRoots)));
      HIPCHK(hipMemcpyAsync(dQ[0], roots.data(), sizeof(CollapseItem) * roots.size(), hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(dRootCount, ones.data(), 4 * ones.size(), hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(dRootBase, forest->wideBase, 4 * size_t(forest->numRoots), hipMemcpyHostToDevice, stream));
      HIPCHK(hipStreamSynchronize(stream));
      nIn = forest->numRoots;
    }
    else
      HIPCHK(hipMemcpyAsync(dQ[0], &first, sizeof(first), hipMemcpyHostToDevice, stream));
    int      cur = 0;
    bool     ok  = true;
    while(nIn && ok)
    {
      HIPCHK(hipMemcpyAsync(dCnt, cnt, 8, hipMemcpyHostToDevice, stream));
      k_collapse<<<(nIn + 63) / 64, 64, 0, stream>>>(dNodesOut, dQ[cur], nIn, dQ[cur ^ 1], dCnt, dWideOut, dRootCount, dRootBase);
      HIPCHK(hipMemcpyAsync(cnt, dCnt, 8, hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      nIn    = cnt[0];
      cnt[0] = 0;
      cur ^= 1;
    }
    *numWideOut = cnt[1];
    if(forest)
    {
      HIPCHK(hipMemcpyAsync(forest->numWide, dRootCount, 4 * size_t(forest->numRoots), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      uint32_t total = 0;
      for(uint32_t r = 0; r < forest->numRoots; ++r)
        total += forest->numWide[r];
      *numWideOut = total;
    }
  }

  sc.release();
  return 0;
fail:
  (void)hipStreamSynchronize(stream);  // nothing of this build may still be running when the arena is reused
  sc.release();
  return -1;
}

// ---- two-level structure: the builds -------------------------------------------------------------------------------------------------
// Every BLAS is an ordinary pt_accel_build over ONE pseudo-instance with the identity transform (object space), written at its bases in the
// shared arrays; then its leaf records are turned into vertex form and its references made global.
int pt_blas_build(hipStream_t stream, const PtTuning& tune, PtBlasDesc* blas, uint32_t numBlas, const float4* dVertices, const uint32_t* dIndices, TriRec* dTris, AlphaRec* dAlpha, WideNode* dWide,
                  char* err, size_t errLen)
{
  if(numBlas == 0)
    return 0;
  uint32_t maxTris = 1;
  for(uint32_t b = 0; b < numBlas; ++b)
    maxTris = std::max(maxTris, blas[b].triCount);
  std::vector<InstanceRec> pseudo(numBlas);
  for(uint32_t b = 0; b < numBlas; ++b)
  {
    InstanceRec& I = pseudo[b];
    std::memset(&I, 0, sizeof(I));
    I.objectToWorld.r0 = I.worldToObject.r0 = make_float4(1.f, 0.f, 0.f, 0.f);
    I.objectToWorld.r1 = I.worldToObject.r1 = make_float4(0.f, 1.f, 0.f, 0.f);
    I.objectToWorld.r2 = I.worldToObject.r2 = make_float4(0.f, 0.f, 1.f, 0.f);
    I.vertexOffset  = blas[b].vertexOffset;
    I.firstIndex    = blas[b].firstIndex;
    I.materialIndex = blas[b].materialIndex;
    I.primMesh      = int32_t(blas[b].primMesh);
    I.triBase       = 0;
    I.triCount      = blas[b].triCount;
    I.flags         = blas[b].flags & ~TRI_FLIP;
  }
  InstanceRec* dPseudo = nullptr;
  int          device  = 0;
  (void)hipGetDevice(&device);
  if(hipMalloc(&dPseudo, sizeof(InstanceRec) * size_t(numBlas)) != hipSuccess)
  {
    snprintf(err, errLen, "BLAS build: out of device memory");
    return -1;
  }
  if(hipMemcpyAsync(dPseudo, pseudo.data(), sizeof(InstanceRec) * size_t(numBlas), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
  {
    (void)hipFree(dPseudo);
    snprintf(err, errLen, "BLAS build: upload failed");
    return -1;
  }
  // Round 6: with the device SAH builder ALL meshes of two or more triangles are built as one forest -- one level-synchronous pass over the concatenated
  // triangles with a root per mesh (pt_internal.h PtForest) instead of one build and ~20 host round trips per mesh (C5 stand-in, 201 meshes: the two-level build 172 -> 70 ms, profiles/r06_forest_build.txt).
  // The arrays' layout is unchanged: mesh b's leaf records at slotBase, its wide nodes from nodeBase on, references global.  What is left for the per-mesh
  // path below: one-triangle meshes, and every mesh under the other builders (build=sah|ploc|lbvh).
  std::vector<char> done(numBlas, 0);
  if(tune.sahBuild == 3)
  {
    // a forest wants its meshes contiguous in the slot range (build_two_level hands out slotBase in order): every maximal run of two or more meshes with
    // at least two triangles each is one forest; what lies between the runs (one-triangle meshes) takes the per-mesh path
    for(uint32_t runStart = 0; runStart < numBlas;)
    {
      std::vector<uint32_t> ids;
      uint32_t              b = runStart;
      while(b < numBlas && blas[b].triCount >= 2 && (ids.empty() || blas[b].slotBase == blas[b - 1].slotBase + blas[b - 1].triCount))
        ids.push_back(b++);
      runStart = ids.empty() ? b + 1 : b;
      if(ids.size() >= 2)
      {
        const uint32_t b0 = ids.front(), slot0 = blas[b0].slotBase;
        uint32_t       nF = 0;
        std::vector<uint32_t> first(ids.size()), count(ids.size()), wideBase(ids.size()), numWide(ids.size(), 0u);
        std::vector<InstanceRec> pf(ids.size());
        for(size_t q = 0; q < ids.size(); ++q)
        {
          const PtBlasDesc& d = blas[ids[q]];
          first[q] = d.slotBase - slot0; count[q] = d.triCount; wideBase[q] = d.nodeBase;
          pf[q]         = pseudo[ids[q]];
          pf[q].triBase = first[q];
          nF += d.triCount;
        }
        InstanceRec* dPf    = nullptr;
        BvhNode*     dNodes = nullptr;
        PtScratch    arena;
        uint32_t     total = 0;
        char         msg[256] = "";
        bool ok = hipMalloc(&dPf, sizeof(InstanceRec) * pf.size()) == hipSuccess && hipMalloc(&dNodes, sizeof(BvhNode) * size_t(nF)) == hipSuccess &&
                  hipMemcpyAsync(dPf, pf.data(), sizeof(InstanceRec) * pf.size(), hipMemcpyHostToDevice, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
        const size_t arenaBytes = size_t(nF) * 704 + (size_t(4) << 20);
        if(ok && hipMalloc((void**)&arena.base, arenaBytes) == hipSuccess)
          arena.cap = arenaBytes;
        else
        {
          arena.base = nullptr;
          (void)hipGetLastError();
        }
        if(ok)
        {
          PtForest F{uint32_t(ids.size()), first.data(), count.data(), wideBase.data(), slot0, numWide.data()};
          ok = pt_accel_build(stream, tune, dPf, uint32_t(pf.size()), dVertices, dIndices, nF, dTris + slot0, dAlpha + slot0, dNodes, dWide, &total, msg, sizeof(msg), nullptr, &arena, &F) == 0;
        }
        if(ok)
        {
          k_forest_vertex_form<<<(nF + 255) / 256, 256, 0, stream>>>(nF, dTris + slot0, dPf, dVertices, dIndices);
          ok = hipStreamSynchronize(stream) == hipSuccess && hipGetLastError() == hipSuccess;
        }
        for(size_t q = 0; q < ids.size() && ok; ++q)
        {
          PtBlasDesc& d = blas[ids[q]];
          d.numWide     = numWide[q];
          done[ids[q]]  = 1;
          if(d.numWide == 0 || d.numWide > std::max(1u, d.triCount - 1))
          {
            snprintf(msg, sizeof(msg), "BLAS %u: %u wide nodes for %u triangles", ids[q], d.numWide, d.triCount);
            ok = false;
          }
        }
        arena.release();
        if(arena.base)
          (void)hipFree(arena.base);
        (void)hipFree(dPf);
        (void)hipFree(dNodes);
        if(!ok)
        {  // the forest is an optimisation: whatever stopped it (memory for its arena, a failed launch), these meshes take the per-mesh path below, which
           // rewrites everything the attempt may have left in their ranges
          (void)hipStreamSynchronize(stream);
          (void)hipGetLastError();
          for(uint32_t q : ids)
            done[q] = 0;
        }
      }
    }
  }
  // A build is a chain of small level-synchronous launches with a host round trip per level: one mesh alone leaves the GPU and the host idle
  // most of the time.  A few host threads, each with its own stream, arena and binary-node scratch, take the meshes from a shared counter
  // (largest first would balance better; the meshes of a scene are usually of similar size).
  const unsigned       numWorkers = std::max(1u, std::min(std::min(numBlas, 16u), uint32_t(tune.blasWorkers > 0 ? tune.blasWorkers : 1)));
  std::atomic<uint32_t> next{0};
  std::atomic<int>      failed{0};
  std::mutex            errLock;
  auto                  worker = [&](unsigned w) {
    (void)hipSetDevice(device);
    hipStream_t ws = nullptr;
    BvhNode*    dNodes = nullptr;
    PtScratch   arena;
    char        msg[256] = "";
    bool        ok = hipStreamCreateWithFlags(&ws, hipStreamNonBlocking) == hipSuccess && hipMalloc(&dNodes, sizeof(BvhNode) * size_t(maxTris)) == hipSuccess;
    if(!ok)
      snprintf(msg, sizeof(msg), "BLAS build: out of device memory");
    // temporaries of one build: < 640 B per triangle (pt_accel_build's lists + the SAH builder's bins); whatever does not fit is allocated singly
    const size_t arenaBytes = size_t(maxTris) * 640 + (size_t(1) << 20);
    if(ok && hipMalloc((void**)&arena.base, arenaBytes) == hipSuccess)
      arena.cap = arenaBytes;
    else
    {
      arena.base = nullptr;  // no arena: every temporary is its own allocation
      (void)hipGetLastError();
    }
    while(ok && !failed.load())
    {
      const uint32_t b = next.fetch_add(1);
      if(b >= numBlas)
        break;
      if(done[b])
        continue;  // built in the forest above
      PtBlasDesc&    d = blas[b];
      const uint32_t n = d.triCount;
      if(pt_accel_build(ws, tune, dPseudo + b, 1, dVertices, dIndices, n, dTris + d.slotBase, dAlpha + d.slotBase, dNodes, dWide + d.nodeBase, &d.numWide, msg, sizeof(msg), nullptr, &arena) != 0)
      {
        ok = false;
        break;
      }
      if(d.numWide == 0 || d.numWide > std::max(1u, n - 1))
      {
        snprintf(msg, sizeof(msg), "BLAS %u: %u wide nodes for %u triangles", b, d.numWide, n);
        ok = false;
        break;
      }
      k_blas_vertex_form<<<(n + 255) / 256, 256, 0, ws>>>(n, dTris + d.slotBase, dVertices, dIndices, d.vertexOffset, d.firstIndex);
      k_blas_rebase<<<(d.numWide + 255) / 256, 256, 0, ws>>>(d.numWide, dWide + d.nodeBase, d.nodeBase, d.slotBase);
    }
    if(ws && (hipStreamSynchronize(ws) != hipSuccess || hipGetLastError() != hipSuccess) && ok)
    {
      snprintf(msg, sizeof(msg), "BLAS build: a kernel failed");
      ok = false;
    }
    if(!ok)
    {
      std::lock_guard<std::mutex> g(errLock);
      if(!failed.exchange(1))
        snprintf(err, errLen, "%s", msg);
    }
    arena.release();
    if(arena.base)
      (void)hipFree(arena.base);
    if(dNodes)
      (void)hipFree(dNodes);
    if(ws)
      (void)hipStreamDestroy(ws);
  };
  std::vector<std::thread> threads;
  for(unsigned w = 1; w < numWorkers; ++w)
    threads.emplace_back(worker, w);
  worker(0);
  for(std::thread& t : threads)
    t.join();
  (void)hipFree(dPseudo);
  return failed.load() ? -1 : 0;
}

// WideNode -> CompactNode (pt_cnode.h cn_encode), one thread per node.  bad[0] counts nodes that cannot be represented (non-finite boxes): the
// caller then keeps the WideNode walk.
// bad[1]: the float bits of max over nodes and axes of |p| + 2047 step, the reach of the nodes' grids (DeviceScene::cnodeBound)
__global__ void k_compact_nodes(uint32_t n, const WideNode* __restrict__ in, CompactNode* __restrict__ out, uint32_t* __restrict__ bad)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  CompactNode c;
  const bool  ok = cn_encode(in[i], c);
  out[i]         = c;
  if(!ok)
    atomicAdd(bad, 1u);
  else
  {
    const float gm = float(CN_GRID_MAX);
    const float m  = fmaxf(fmaxf(fabsf(c.px) + gm * __uint_as_float((c.exps & 0xffu) << 23), fabsf(c.py) + gm * __uint_as_float(((c.exps >> 8) & 0xffu) << 23)),
                           fabsf(c.pz) + gm * __uint_as_float(((c.exps >> 16) & 0xffu) << 23));
    atomicMax(bad + 1, __float_as_uint(m));  // non-negative floats order like their bit patterns
  }
}
// the same over a list of node ranges (the bottom-level structures of the two-level mode sit at their node bases with unused nodes between them):
// one block per range
__global__ void k_compact_ranges(const uint2* __restrict__ ranges, const WideNode* __restrict__ in, CompactNode* __restrict__ out, uint32_t* __restrict__ bad)
{
  const uint2 r = ranges[blockIdx.x];
  for(uint32_t i = threadIdx.x; i < r.y; i += blockDim.x)
  {
    CompactNode c;
    const bool  ok = cn_encode(in[r.x + i], c);
    out[r.x + i]   = c;
    if(!ok)
      atomicAdd(bad, 1u);
  }
}
int pt_compact_node_ranges(hipStream_t stream, const uint32_t* hBaseCount, uint32_t numRanges, const WideNode* in, CompactNode* out)
{
  if(numRanges == 0)
    return 0;
  uint32_t* dBuf = nullptr;  // [0]: bad counter, then the ranges
  uint32_t  bad  = 1;
  if(hipMalloc(&dBuf, 8 + 8 * size_t(numRanges)) != hipSuccess)
  {
    (void)hipGetLastError();
    return -1;
  }
  if(hipMemsetAsync(dBuf, 0, 8, stream) == hipSuccess && hipMemcpyAsync(dBuf + 2, hBaseCount, 8 * size_t(numRanges), hipMemcpyHostToDevice, stream) == hipSuccess)
  {
    k_compact_ranges<<<numRanges, 128, 0, stream>>>((const uint2*)(dBuf + 2), in, out, dBuf);
    if(hipMemcpyAsync(&bad, dBuf, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess)
      bad = 1;
  }
  (void)hipFree(dBuf);
  return bad ? -1 : 0;
}
int pt_compact_nodes(hipStream_t stream, uint32_t n, const WideNode* in, CompactNode* out, float* reachOut)
{
  if(reachOut)
    *reachOut = 0.0f;
  if(n == 0)
    return 0;
  uint32_t* dBad = nullptr;
  uint32_t  bad[2] = {1u, 0u};
  if(hipMalloc(&dBad, 8) != hipSuccess)
  {
    (void)hipGetLastError();
    return -1;
  }
  if(hipMemsetAsync(dBad, 0, 8, stream) == hipSuccess)
  {
    k_compact_nodes<<<(n + 127) / 128, 128, 0, stream>>>(n, in, out, dBad);
    if(hipMemcpyAsync(bad, dBad, 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess)
      bad[0] = 1;
  }
  (void)hipFree(dBad);
  if(reachOut && !bad[0])
    std::memcpy(reachOut, &bad[1], 4);
  return bad[0] ? -1 : 0;
}

// DeviceScene::shadeTris of a flat-format structure: per leaf slot the three vertices' attribute pairs, copied from where the record's instance and
// primitive point
__global__ void k_shade_tris(uint32_t n, const TriRec* __restrict__ tris, const InstanceRec* __restrict__ inst, const float4* __restrict__ vertices,
                             const uint32_t* __restrict__ indices, float4* __restrict__ out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const uint32_t     ii = __float_as_uint(tris[i].e1n.w), prim = __float_as_uint(tris[i].e2p.w);
  const InstanceRec& I = inst[ii];
  const uint32_t*    t = indices + I.firstIndex + 3 * size_t(prim);
  float4*            o = out + size_t(i) * PT_SHADE_REC_QUADS;  // one 128-byte line per slot
  for(int k = 0; k < 3; ++k)
  {
    const size_t v = size_t(I.vertexOffset + t[k]) * 2;
    o[2 * k]     = vertices[v];
    o[2 * k + 1] = vertices[v + 1];
  }
  o[6] = make_float4(__uint_as_float(ii), __uint_as_float(prim), __uint_as_float(uint32_t(I.materialIndex)), 0.f);  // what k_shade read the 48-byte TriRec for + the material index
  o[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}
void pt_launch_shade_tris(hipStream_t stream, uint32_t n, const TriRec* tris, const InstanceRec* inst, const float4* vertices, const uint32_t* indices, float4* out)
{
  if(n)
    k_shade_tris<<<(n + 255) / 256, 256, 0, stream>>>(n, tris, inst, vertices, indices, out);
}

// The merged world-space structure of the two-level mode: the flat build over the instances listed in hInst (copies of the scene's records
// with triBase renumbered 0, n0, n0 + n1, ... so that k_world_tris finds them), then every record gets the identity it has in the scene --
// instance id ids[j] and world index worldBase[j] + primitive -- and the child references become global (k_blas_rebase).
__global__ void k_merged_identity(uint32_t n, TriRec* __restrict__ tris, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ worldBase)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const uint32_t j = __float_as_uint(tris[i].e1n.w), prim = __float_as_uint(tris[i].e2p.w);
  const uint32_t w = __float_as_uint(tris[i].p0w.w);
  tris[i].p0w.w    = __uint_as_float((w & ~TRI_INDEX_MASK) | (worldBase[j] + prim));
  tris[i].e1n.w    = __uint_as_float(ids[j]);
}
int pt_merged_build(hipStream_t stream, const PtTuning& tune, const InstanceRec* hInst, const uint32_t* hIds, const uint32_t* hWorldBase, uint32_t numInst, uint32_t numTris, const float4* dVertices,
                    const uint32_t* dIndices, TriRec* dTris, AlphaRec* dAlpha, WideNode* dWide, uint32_t slotBase, uint32_t nodeBase, uint32_t* numWideOut, float* boxOut6, char* err,
                    size_t errLen)
{
  InstanceRec* dInst = nullptr;
  uint32_t*    dIds  = nullptr;
  BvhNode*     dNodes = nullptr;
  PtScratch    arena;
  int          rc = -1;
  BvhNode      root{};
  const size_t arenaBytes = size_t(numTris) * 640 + (size_t(1) << 20);
  if(hipMalloc(&dInst, sizeof(InstanceRec) * size_t(numInst)) != hipSuccess || hipMalloc(&dIds, 8 * size_t(numInst)) != hipSuccess ||
     hipMalloc(&dNodes, sizeof(BvhNode) * size_t(std::max(1u, numTris))) != hipSuccess)
  {
    snprintf(err, errLen, "merged BLAS build: out of device memory");
    goto done;
  }
  if(hipMalloc((void**)&arena.base, arenaBytes) == hipSuccess)
    arena.cap = arenaBytes;
  else
  {
    arena.base = nullptr;
    (void)hipGetLastError();
  }
  if(hipMemcpyAsync(dInst, hInst, sizeof(InstanceRec) * size_t(numInst), hipMemcpyHostToDevice, stream) != hipSuccess ||
     hipMemcpyAsync(dIds, hIds, 4 * size_t(numInst), hipMemcpyHostToDevice, stream) != hipSuccess ||
     hipMemcpyAsync(dIds + numInst, hWorldBase, 4 * size_t(numInst), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
  {
    snprintf(err, errLen, "merged BLAS build: upload failed");
    goto done;
  }
  if(pt_accel_build(stream, tune, dInst, numInst, dVertices, dIndices, numTris, dTris + slotBase, dAlpha + slotBase, dNodes, dWide + nodeBase, numWideOut, err, errLen, nullptr, &arena) != 0)
    goto done;
  k_merged_identity<<<(numTris + 255) / 256, 256, 0, stream>>>(numTris, dTris + slotBase, dIds, dIds + numInst);
  k_blas_rebase<<<(*numWideOut + 255) / 256, 256, 0, stream>>>(*numWideOut, dWide + nodeBase, nodeBase, slotBase);
  if(hipMemcpyAsync(&root, dNodes, sizeof(BvhNode), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess)
  {
    snprintf(err, errLen, "merged BLAS build: a kernel failed");
    goto done;
  }
  {  // world box = union of the binary root's child boxes (one child when the structure is a single triangle)
    const bool  two   = numTris > 1 && root.d.y != BVH_NONE;
    const float l[6]  = {root.a.x, root.a.y, root.a.z, root.a.w, root.b.x, root.b.y}, r[6] = {root.b.z, root.b.w, root.c.x, root.c.y, root.c.z, root.c.w};
    for(int k = 0; k < 3; ++k)
    {
      boxOut6[k]     = two ? std::min(l[k], r[k]) : l[k];
      boxOut6[3 + k] = two ? std::max(l[3 + k], r[3 + k]) : l[3 + k];
    }
  }
  rc = 0;
done:
  (void)hipStreamSynchronize(stream);
  arena.release();
  if(arena.base)
    (void)hipFree(arena.base);
  if(dInst) (void)hipFree(dInst);
  if(dIds) (void)hipFree(dIds);
  if(dNodes) (void)hipFree(dNodes);
  return rc;
}

int pt_tlas_build(hipStream_t stream, const PtTuning& tune, const InstanceRec* dInst, const uint32_t* dActive, uint32_t numActive, const uint32_t* dInstNodeBase, const float* dInstPad,
                  const float4* dVertices, const uint32_t* dIndices, WideNode* dTlasOut, TlasLeaf* dLeavesOut, BvhNode* rootOut, uint32_t* numWideOut, char* err, size_t errLen,
                  const float* mergedBox, uint32_t mergedNodeBase)
{
  *numWideOut = 0;
  const uint32_t n = numActive + (mergedBox ? 1u : 0u);  // the merged structure is one more primitive of the TLAS
  if(n == 0)
    return 0;
  TriRec *       dProx = nullptr, *dLeafOrder = nullptr;
  AlphaRec*      dAlpha = nullptr;
  BvhNode*       dNodes = nullptr;
  int            rc = -1;
  if(hipMalloc(&dProx, sizeof(TriRec) * size_t(n)) != hipSuccess || hipMalloc(&dLeafOrder, sizeof(TriRec) * size_t(n)) != hipSuccess ||
     hipMalloc(&dAlpha, sizeof(AlphaRec) * size_t(n)) != hipSuccess || hipMalloc(&dNodes, sizeof(BvhNode) * size_t(n)) != hipSuccess)
  {
    snprintf(err, errLen, "TLAS build: out of device memory");
    goto done;
  }
  if(numActive)
    k_instance_proxies<<<numActive, 256, 0, stream>>>(dActive, dInst, dVertices, dIndices, dProx);
  if(mergedBox)
  {
    TriRec r;
    const uint32_t tag = TRI_INDEX_MASK;  // k_tlas_leaves recognises the proxy by this index
    float          tagF;
    std::memcpy(&tagF, &tag, 4);
    r.p0w = make_float4(mergedBox[0], mergedBox[1], mergedBox[2], tagF);
    r.e1n = make_float4(mergedBox[3] - mergedBox[0], mergedBox[4] - mergedBox[1], mergedBox[5] - mergedBox[2], 0.f);
    r.e2p = make_float4(0.f, 0.f, 0.f, 0.f);
    if(hipMemcpyAsync(dProx + numActive, &r, sizeof(r), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
    {
      snprintf(err, errLen, "TLAS build: upload failed");
      goto done;
    }
  }
  if(pt_accel_build(stream, tune, nullptr, 0, nullptr, nullptr, n, dLeafOrder, dAlpha, dNodes, dTlasOut, numWideOut, err, errLen, dProx, nullptr) != 0)
    goto done;
  k_tlas_leaves<<<(n + 255) / 256, 256, 0, stream>>>(n, dLeafOrder, dInst, dInstNodeBase, dInstPad, dLeavesOut, mergedNodeBase);
  if(hipMemcpyAsync(rootOut, dNodes, sizeof(BvhNode), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess)
  {
    snprintf(err, errLen, "TLAS build: a kernel failed");
    goto done;
  }
  rc = 0;
done:
  (void)hipStreamSynchronize(stream);
  void* all[] = {dProx, dLeafOrder, dAlpha, dNodes};
  for(void* q : all)
    if(q)
      (void)hipFree(q);
  return rc;
}
