// Device LBVH build for gfx950.  Replaces AccelStructure::create of the reference
// (src/accelstruct.cpp:55-162: one BLAS per prim-mesh + one TLAS instance per node, built by the
// Vulkan driver) with a single flattened world-space BVH2:
//
//   k_world_tris   instance transforms applied in fp32 (trace contract T1) -> TriRec + centroid
//   k_bounds       wave-reduced atomic min/max of the centroids
//   k_morton       30-bit Morton code of the centroid, value = world triangle index
//   radix sort     4 passes x 8 bits, stable, one wave per 2048-key block (histogram / scan / scatter)
//   k_hierarchy    Karras 2012 binary radix tree over the sorted codes (ties broken by position)
//   k_refit        bottom-up AABB merge with per-node arrival counters (agent-scope fence per hand-off)
//   k_emit         64-byte traversal nodes holding both child boxes
//
// Everything runs on the caller's stream; temporaries are freed before returning.
#include <hip/hip_runtime.h>
#include <vector>
#include <cfloat>
#include <cstdio>
#include "pt_device.h"
#include "pt_internal.h"

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

// leaf reference of sorted slot `i`: non-opaque triangles are tagged so that traversal fetches their AlphaRec up front
PT_DEV uint32_t leaf_ref(const TriRec* __restrict__ tris, uint32_t leaf)
{
  const uint32_t slot  = leaf & ~BVH_LEAF;
  const uint32_t flags = __float_as_uint(tris[slot].p0w.w) >> 29;
  return BVH_LEAF | slot | ((flags & TRI_OPAQUE) ? 0u : BVH_ALPHA);
}

__global__ void k_emit(int numInner, const uint32_t* __restrict__ childL, const uint32_t* __restrict__ childR, const float4* __restrict__ leafLo,
                       const float4* __restrict__ leafHi, const float4* __restrict__ nodeLo, const float4* __restrict__ nodeHi, const TriRec* __restrict__ tris, BvhNode* __restrict__ out)
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
  nd.d   = make_uint4((l & BVH_LEAF) ? leaf_ref(tris, l) : (l | (llo.w > 0.f ? BVH_ALPHA : 0u)), (r & BVH_LEAF) ? leaf_ref(tris, r) : (r | (rlo.w > 0.f ? BVH_ALPHA : 0u)), 0u, 0u);
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


// ---- collapse BVH2 -> wide BVH (level-synchronous; one thread per wide node) -------------------------------------
struct CollapseItem {
  uint32_t b2;    // binary node to expand
  uint32_t wide;  // wide node it becomes
};
PT_DEV float half_area(float4 lo, float4 hi)
{
  float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  return dx * dy + dy * dz + dz * dx;
}
__global__ void k_collapse(const BvhNode* __restrict__ b2, const CollapseItem* __restrict__ qin, uint32_t nIn, CollapseItem* __restrict__ qout, uint32_t* counters /* [0]=nOut [1]=wideCount */,
                           WideNode* __restrict__ out)
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
          uint32_t wid = atomicAdd(&counters[1], 1u);
          uint32_t qi  = atomicAdd(&counters[0], 1u);
          qout[qi]     = CollapseItem{id[c] & BVH_SLOT_MASK, wid};
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
int pt_accel_build(hipStream_t stream, const InstanceRec* dInst, uint32_t numInst, const float4* dVertices, const uint32_t* dIndices, uint32_t numTris,
                   TriRec* dTrisOut, AlphaRec* dAlphaOut, BvhNode* dNodesOut, WideNode* dWideOut, uint32_t* numWideOut, char* err, size_t errLen)
{
  *numWideOut = 0;
  if(numTris == 0)
    return 0;
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
  bool      sah     = false;
  uint32_t  initBounds[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};

  HIPCHK(hipMalloc(&dUnsorted, sizeof(TriRec) * size_t(n)));
  HIPCHK(hipMalloc(&dAlphaUnsorted, sizeof(AlphaRec) * size_t(n)));
  HIPCHK(hipMalloc(&dCen, sizeof(float4) * size_t(n)));
  HIPCHK(hipMalloc(&dLeafLo, sizeof(float4) * size_t(n)));
  HIPCHK(hipMalloc(&dLeafHi, sizeof(float4) * size_t(n)));
  HIPCHK(hipMalloc(&dNodeLo, sizeof(float4) * size_t(n)));
  HIPCHK(hipMalloc(&dNodeHi, sizeof(float4) * size_t(n)));
  HIPCHK(hipMalloc(&dKeysA, 4 * size_t(n)));
  HIPCHK(hipMalloc(&dKeysB, 4 * size_t(n)));
  HIPCHK(hipMalloc(&dValsA, 4 * size_t(n)));
  HIPCHK(hipMalloc(&dValsB, 4 * size_t(n)));
  HIPCHK(hipMalloc(&dHist, 4 * size_t(256) * sortBlocks));
  HIPCHK(hipMalloc(&dBounds, 4 * 6));
  HIPCHK(hipMalloc(&dChildL, 4 * size_t(n)));
  HIPCHK(hipMalloc(&dChildR, 4 * size_t(n)));
  HIPCHK(hipMalloc(&dParI, 4 * size_t(n)));
  HIPCHK(hipMalloc(&dParL, 4 * size_t(n)));
  HIPCHK(hipMalloc(&dArrive, 4 * size_t(n)));
  HIPCHK(hipMemcpyAsync(dBounds, initBounds, sizeof(initBounds), hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemsetAsync(dArrive, 0, 4 * size_t(n), stream));

  k_world_tris<<<G, B, 0, stream>>>(n, dInst, numInst, dVertices, dIndices, dUnsorted, dAlphaUnsorted, dCen, dBounds);
  sah = g_tuning.sahBuild != 0 && n >= 2;
  if(sah)
  {
    // fast-trace build: SAH topology on the host from the world-space triangles (pt_sah.hip); boxes stay on the device
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
  if(n == 1)
  {
    k_single_leaf<<<1, 1, 0, stream>>>(dLeafLo, dLeafHi, dTrisOut, dNodesOut);
  }
  else
  {
    if(!sah)
      k_hierarchy<<<G, B, 0, stream>>>(int(n), dKeysA, dChildL, dChildR, dParI, dParL);
    k_refit<<<G, B, 0, stream>>>(int(n), dChildL, dChildR, dParI, dParL, dLeafLo, dLeafHi, dNodeLo, dNodeHi, dArrive);
    k_emit<<<(n - 1 + B - 1) / B, B, 0, stream>>>(int(n - 1), dChildL, dChildR, dLeafLo, dLeafHi, dNodeLo, dNodeHi, dTrisOut, dNodesOut);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(stream));

  // ---- collapse to the wide layout, one BVH level per launch (the queue sizes come back to the host between levels;
  // the build is not on the timed path)
  {
    CollapseItem* dQ[2] = {nullptr, nullptr};
    uint32_t*     dCnt  = nullptr;
    HIPCHK(hipMalloc(&dQ[0], sizeof(CollapseItem) * size_t(n)));
    HIPCHK(hipMalloc(&dQ[1], sizeof(CollapseItem) * size_t(n)));
    HIPCHK(hipMalloc(&dCnt, 8));
    CollapseItem first{0u, 0u};
    uint32_t     cnt[2] = {0u, 1u};  // next-queue size, wide nodes allocated (root = 0)
    HIPCHK(hipMemcpyAsync(dQ[0], &first, sizeof(first), hipMemcpyHostToDevice, stream));
    uint32_t nIn = 1;
    int      cur = 0;
    bool     ok  = true;
    while(nIn && ok)
    {
      HIPCHK(hipMemcpyAsync(dCnt, cnt, 8, hipMemcpyHostToDevice, stream));
      k_collapse<<<(nIn + 63) / 64, 64, 0, stream>>>(dNodesOut, dQ[cur], nIn, dQ[cur ^ 1], dCnt, dWideOut);
      HIPCHK(hipMemcpyAsync(cnt, dCnt, 8, hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      nIn    = cnt[0];
      cnt[0] = 0;
      cur ^= 1;
    }
    *numWideOut = cnt[1];
    (void)hipFree(dQ[0]);
    (void)hipFree(dQ[1]);
    (void)hipFree(dCnt);
  }

  {
    void* all[] = {dUnsorted, dAlphaUnsorted, dCen, dLeafLo, dLeafHi, dNodeLo, dNodeHi, dKeysA, dKeysB, dValsA, dValsB, dHist, dBounds, dChildL, dChildR, dParI, dParL, dArrive};
    for(void* p : all)
      (void)hipFree(p);
  }
  return 0;
fail:
{
  void* all[] = {dUnsorted, dAlphaUnsorted, dCen, dLeafLo, dLeafHi, dNodeLo, dNodeHi, dKeysA, dKeysB, dValsA, dValsB, dHist, dBounds, dChildL, dChildR, dParI, dParL, dArrive};
  for(void* p : all)
    if(p)
      (void)hipFree(p);
}
  return -1;
}
