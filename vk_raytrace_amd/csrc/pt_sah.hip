// Host-side SAH topology builder for the acceleration structure ("prefer fast trace": the reference asks the Vulkan
// driver for exactly that, src/accelstruct.cpp:125-161 builds once per scene with PREFER_FAST_TRACE).
//
// Top-down binned surface-area heuristic over the world-space triangles produced on the device (k_world_tris),
// one triangle per leaf.  Only the TOPOLOGY comes from here -- leaf order and child / parent links in the layout the
// LBVH path produces (Karras numbering is not required by anything downstream) -- so the device pipeline after it
// (gather into leaf order, bottom-up refit with the padded leaf boxes, node emission, collapse to wide nodes) is the
// same for both builders and the box arithmetic exists once.  Large ranges split across std::async tasks.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <future>
#include <vector>
#include "pt_device.h"
#include "pt_internal.h"
#include "pt_sahdev.h"

namespace {

struct Prim {
  float    lo[3], hi[3], c[3];
  uint32_t id;
};
struct Box {
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  void  grow(const float* l, const float* h)
  {
    for(int a = 0; a < 3; ++a)
    {
      lo[a] = l[a] < lo[a] ? l[a] : lo[a];
      hi[a] = h[a] > hi[a] ? h[a] : hi[a];
    }
  }
  void  grow(const Box& b) { grow(b.lo, b.hi); }
  float area() const
  {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx < 0.f || dy < 0.f || dz < 0.f) ? 0.f : 2.0f * (dx * dy + dy * dz + dz * dx);
  }
};

#ifndef PT_SAH_BINS
#define PT_SAH_BINS 32
#endif
#ifndef PT_SAH_SWEEP
#define PT_SAH_SWEEP 12
#endif
constexpr int      kBins        = PT_SAH_BINS;
constexpr uint32_t kSweepBelow  = PT_SAH_SWEEP;  // ranges this small evaluate every split of every axis
constexpr uint32_t kSpawnAbove  = 1u << 15;

struct Builder {
  std::vector<Prim>& prims;
  uint32_t *         childL, *childR, *parI, *parL;

  // Chooses the SAH split of prims[first, first+count) and partitions the range; returns the size of the left part.
  uint32_t split(uint32_t first, uint32_t count)
  {
    Prim* P = prims.data() + first;
    if(count <= kSweepBelow)
    {
      float    bestCost = 3.0e38f;
      int      bestAxis = -1;
      uint32_t bestK    = 0;
      std::vector<float> rightArea(count);
      for(int a = 0; a < 3; ++a)
      {
        std::sort(P, P + count, [a](const Prim& x, const Prim& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.id < y.id); });
        Box b;
        for(uint32_t i = count; i-- > 1;)
        {
          b.grow(P[i].lo, P[i].hi);
          rightArea[i] = b.area();
        }
        Box l;
        for(uint32_t k = 1; k < count; ++k)
        {
          l.grow(P[k - 1].lo, P[k - 1].hi);
          const float cost = l.area() * float(k) + rightArea[k] * float(count - k);
          if(cost < bestCost)
          {
            bestCost = cost;
            bestAxis = a;
            bestK    = k;
          }
        }
      }
      if(bestAxis != 2)  // the range is currently sorted along axis 2
        std::sort(P, P + count, [bestAxis](const Prim& x, const Prim& y) { return x.c[bestAxis] < y.c[bestAxis] || (x.c[bestAxis] == y.c[bestAxis] && x.id < y.id); });
      return bestK;
    }

    Box cb;
    for(uint32_t i = 0; i < count; ++i)
      cb.grow(P[i].c, P[i].c);
    float bestCost = 3.0e38f, bestPos = 0.f;
    int   bestAxis = -1;
    for(int a = 0; a < 3; ++a)
    {
      const float ext = cb.hi[a] - cb.lo[a];
      if(!(ext > 0.f))
        continue;
      const float scale = float(kBins) / ext;
      Box         bb[kBins];
      uint32_t    bn[kBins] = {};
      for(uint32_t i = 0; i < count; ++i)
      {
        int k = int((P[i].c[a] - cb.lo[a]) * scale);
        k     = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
        bb[k].grow(P[i].lo, P[i].hi);
        bn[k]++;
      }
      float    ra[kBins];
      uint32_t rn[kBins];
      Box      r;
      uint32_t n = 0;
      for(int k = kBins - 1; k >= 1; --k)
      {
        if(bn[k])
          r.grow(bb[k]);
        n += bn[k];
        ra[k] = r.area();
        rn[k] = n;
      }
      Box l;
      n = 0;
      for(int k = 1; k < kBins; ++k)
      {
        if(bn[k - 1])
          l.grow(bb[k - 1]);
        n += bn[k - 1];
        if(n == 0 || rn[k] == 0)
          continue;
        const float cost = l.area() * float(n) + ra[k] * float(rn[k]);
        if(cost < bestCost)
        {
          bestCost = cost;
          bestAxis = a;
          bestPos  = float(k);
        }
      }
    }
    if(bestAxis >= 0)
    {
      const int   a     = bestAxis;
      const float scale = float(kBins) / (cb.hi[a] - cb.lo[a]);
      const float lo    = cb.lo[a];
      const int   kSplit = int(bestPos);
      Prim*       mid   = std::partition(P, P + count, [&](const Prim& p) {
        int k = int((p.c[a] - lo) * scale);
        k     = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
        return k < kSplit;
      });
      const uint32_t nl = uint32_t(mid - P);
      if(nl > 0 && nl < count)
        return nl;
    }
    // all centroids coincide (or binning failed): median split along the widest axis
    int   a   = 0;
    float ext = -1.f;
    for(int k = 0; k < 3; ++k)
      if(cb.hi[k] - cb.lo[k] > ext)
      {
        ext = cb.hi[k] - cb.lo[k];
        a   = k;
      }
    std::nth_element(P, P + count / 2, P + count, [a](const Prim& x, const Prim& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.id < y.id); });
    return count / 2;
  }

  // Inner node `node` covers prims[first, first+count), count >= 2; a subtree over k leaves owns k-1 consecutive ids.
  void build(uint32_t first, uint32_t count, uint32_t node)
  {
    const uint32_t nl = split(first, count), nr = count - nl;
    const uint32_t leftId = node + 1, rightId = node + nl;  // left subtree owns nl-1 ids after `node`
    std::future<void> task;
    if(nl == 1)
    {
      childL[node] = first | BVH_LEAF;
      parL[first]  = node;
    }
    else
    {
      childL[node] = leftId;
      parI[leftId] = node;
      if(nl > kSpawnAbove && nr > kSpawnAbove)
        task = std::async(std::launch::async, [=] { build(first, nl, leftId); });
      else
        build(first, nl, leftId);
    }
    if(nr == 1)
    {
      childR[node]     = (first + nl) | BVH_LEAF;
      parL[first + nl] = node;
    }
    else
    {
      childR[node]  = rightId;
      parI[rightId] = node;
      build(first + nl, nr, rightId);
    }
    if(task.valid())
      task.get();
  }
};

}  // namespace

// tris: n world-space triangle records in input order.  Outputs (host arrays of n entries each): vals = leaf order
// (leaf i holds input triangle vals[i]), childL / childR / parI over the n-1 inner nodes (root 0, BVH_LEAF-tagged
// leaf indices), parL over the leaves.  n >= 2.
void pt_sah_topology(uint32_t n, const TriRec* tris, uint32_t* vals, uint32_t* childL, uint32_t* childR, uint32_t* parI, uint32_t* parL)
{
  std::vector<Prim> prims(n);
  for(uint32_t i = 0; i < n; ++i)
  {
    const TriRec& t = tris[i];
    const float   p0[3] = {t.p0w.x, t.p0w.y, t.p0w.z};
    const float   p1[3] = {t.p0w.x + t.e1n.x, t.p0w.y + t.e1n.y, t.p0w.z + t.e1n.z};
    const float   p2[3] = {t.p0w.x + t.e2p.x, t.p0w.y + t.e2p.y, t.p0w.z + t.e2p.z};
    Prim&         p = prims[i];
    for(int a = 0; a < 3; ++a)
    {
      p.lo[a] = std::fmin(p0[a], std::fmin(p1[a], p2[a]));
      p.hi[a] = std::fmax(p0[a], std::fmax(p1[a], p2[a]));
      if(!std::isfinite(p.lo[a]) || !std::isfinite(p.hi[a]))
        p.lo[a] = p.hi[a] = 0.0f;  // non-finite input must not reach the comparators (strict weak ordering); the triangle can never be hit anyway
      p.c[a] = 0.5f * (p.lo[a] + p.hi[a]);
    }
    p.id = i;
  }
  Builder b{prims, childL, childR, parI, parL};
  parI[0] = BVH_NONE;
  b.build(0, n, 0);
  for(uint32_t i = 0; i < n; ++i)
    vals[i] = prims[i].id;
}

// Test hook (tests/test_sah_cpu.py; not part of include/pt_api.h): the topology builder on plain arrays, no GPU involved.
// tri9: n x 9 floats (p0, e1, e2).
extern "C" __attribute__((visibility("default"))) int pt_debug_sah_topology(uint32_t n, const float* tri9, uint32_t* vals, uint32_t* childL, uint32_t* childR, uint32_t* parI, uint32_t* parL)
{
  if(n < 2 || !tri9)
    return -1;
  std::vector<TriRec> t(n);
  for(uint32_t i = 0; i < n; ++i)
  {
    const float* p = tri9 + 9 * size_t(i);
    t[i].p0w = make_float4(p[0], p[1], p[2], 0.f);
    t[i].e1n = make_float4(p[3], p[4], p[5], 0.f);
    t[i].e2p = make_float4(p[6], p[7], p[8], 0.f);
  }
  pt_sah_topology(n, t.data(), vals, childL, childR, parI, parL);
  return 0;
}

// Test hook: the DEVICE builder (pt_sahdev.h) emulated on the host -- the same per-thread bodies the HIP kernels run, executed one "thread"
// after the other, level by level.  tests/test_sah_cpu.py holds its output to the tree invariants and to the host builder's SAH cost.
extern "C" __attribute__((visibility("default"))) int pt_debug_sahdev_topology(uint32_t n, const float* tri9, uint32_t* vals, uint32_t* childL, uint32_t* childR, uint32_t* parI, uint32_t* parL)
{
  if(n < 2 || !tri9)
    return -1;
  std::vector<TriRec> t(n);
  for(uint32_t i = 0; i < n; ++i)
  {
    const float* p = tri9 + 9 * size_t(i);
    t[i].p0w = make_float4(p[0], p[1], p[2], 0.f);
    t[i].e1n = make_float4(p[3], p[4], p[5], 0.f);
    t[i].e2p = make_float4(p[6], p[7], p[8], 0.f);
  }
  std::vector<float4>   plo(n), phi(n);
  std::vector<uint32_t> idxA(n), idxB(n), pwA(n, 0u), pwB(n, SD_NONE);
  for(uint32_t i = 0; i < n; ++i)
  {
    sd_prim(i, t.data(), plo.data(), phi.data());
    idxA[i] = i;
  }
  const size_t        maxWork = size_t(n) / (SD_SMALL + 1) + 2;
  std::vector<SdWork> workA(maxWork), workB(maxWork), small(size_t(n) / 2 + 2);
  uint32_t            nActive = 0, nextCount = 0, smallCount = 0;
  parI[0] = BVH_NONE;
  if(n <= SD_SMALL)
  {
    sd_init_work(small[smallCount++], 0, n, 0);
    std::fill(pwA.begin(), pwA.end(), SD_NONE);
  }
  else
    sd_init_work(workA[nActive++], 0, n, 0);
  std::vector<uint32_t> binCnt, binBox;
  int                   level = 0;
  while(nActive)
  {
    binCnt.assign(size_t(nActive) * 3 * SD_BINS, 0u);
    binBox.resize(size_t(nActive) * 3 * SD_BINS * 6);
    for(size_t b = 0; b < size_t(nActive) * 3 * SD_BINS; ++b)
      for(int q = 0; q < 6; ++q)
        binBox[b * 6 + q] = q < 3 ? SD_ORD_PLUS_INF : SD_ORD_MINUS_INF;
    if(level == 0)  // deeper levels get their centroid bounds from the partition pass of their parents
      for(uint32_t pos = 0; pos < n; ++pos)
        sd_cbounds(pos, idxA.data(), pwA.data(), workA.data(), plo.data(), phi.data());
    ++level;
    for(uint32_t pos = 0; pos < n; ++pos)
      sd_bin(pos, idxA.data(), pwA.data(), workA.data(), plo.data(), phi.data(), binCnt.data(), binBox.data());
    nextCount = 0;
    SdLists L{workB.data(), &nextCount, small.data(), &smallCount};
    for(uint32_t w = 0; w < nActive; ++w)
      sd_split(w, workA.data(), binCnt.data(), binBox.data(), L, childL, childR, parI, parL);
    for(uint32_t pos = 0; pos < n; ++pos)
      sd_partition(pos, idxA.data(), pwA.data(), workA.data(), workB.data(), plo.data(), phi.data(), idxB.data(), pwB.data());
    idxA.swap(idxB);
    pwA.swap(pwB);
    workA.swap(workB);
    nActive = nextCount;
  }
  for(uint32_t s = 0; s < smallCount; ++s)
    sd_small(s, small.data(), idxA.data(), plo.data(), phi.data(), childL, childR, parI, parL);
  for(uint32_t i = 0; i < n; ++i)
    vals[i] = idxA[i];
  return 0;
}
