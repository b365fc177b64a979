// Level-synchronous binned-SAH topology builder for the device (the default acceleration-structure builder of libptmi.so).
//
// Replaces what the reference asks of the Vulkan driver -- vkCmdBuildAccelerationStructuresKHR with PREFER_FAST_TRACE
// (src/accelstruct.cpp:125-161) -- with the same algorithm as the host builder of pt_sah.hip (top-down surface-area heuristic, 32 bins,
// exact sweeps for ranges of <= 12 triangles, one triangle per leaf), restructured so that every step is a data-parallel pass:
//
//   per level, over all triangles of the still-open nodes:
//     sd_cbounds    centroid bounds of every open node                     (atomic min / max per node)
//     sd_bin        32 bins x 3 axes per open node: counts + boxes          (atomics per bin)
//     sd_split      one thread per open node: SAH over the 3 x 31 bin boundaries -> axis, boundary, left count; children are numbered
//                   (a subtree over k leaves owns k-1 consecutive inner ids) and sent to the next level, to the small-node list, or
//                   finished as leaves
//     sd_partition  every triangle moves to its side of its node's range
//   finally sd_small: one thread per node of <= SD_SMALL triangles builds that whole subtree with exact sweeps.
//
// The bodies below are plain functions of a thread index so that the SAME code runs as HIP kernels (pt_accel.hip) and, sequentially, on the
// host (pt_debug_sahdev_topology in pt_sah.hip: CPU tests hold the emulated build to the tree invariants and to the host builder's SAH cost).
#pragma once
#include <stdint.h>
#include "pt_device.h"

#define SD_BINS 32
#define SD_SMALL 12
#define SD_NONE 0xffffffffu

#include <string.h>
#if defined(__HIPCC__)
#define SD_FN __host__ __device__ inline
#else
#define SD_FN static inline
#endif
// device pass: real atomics; host pass: the emulation runs the "threads" one after the other, so plain read-modify-write is exact
SD_FN uint32_t sd_add(uint32_t* p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(p, v);
#else
  uint32_t o = *p; *p += v; return o;
#endif
}
SD_FN void sd_min(uint32_t* p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
  atomicMin(p, v);
#else
  if(v < *p) *p = v;
#endif
}
SD_FN void sd_max(uint32_t* p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
  atomicMax(p, v);
#else
  if(v > *p) *p = v;
#endif
}
SD_FN uint32_t sd_fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
SD_FN float    sd_bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// order-preserving map float -> uint32 (for atomic min / max on floats)
SD_FN uint32_t sd_order(float f)
{
  uint32_t u = sd_fbits(f);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
SD_FN float sd_unorder(uint32_t u)
{
  u ^= ((u >> 31) ? 0x80000000u : 0xffffffffu);
  return sd_bitsf(u);
}
#define SD_ORD_PLUS_INF 0xff800000u   /* sd_order(+inf) */
#define SD_ORD_MINUS_INF 0x007fffffu  /* sd_order(-inf) */

struct SdWork {
  uint32_t first, count, node;  // prims [first, first+count) of the index array, inner node id
  uint32_t cbLo[3], cbHi[3];    // centroid bounds (ordered bits)
  int32_t  axis, kSplit;        // chosen split: bin boundary kSplit on `axis`; axis < 0: split the range in the middle (degenerate input)
  uint32_t nl;                  // triangles on the left side
  uint32_t leftW, rightW;       // index of the children in the next level's work list, SD_NONE: leaf or small node
  uint32_t curL, curR;          // partition cursors
  float    lo, scale;           // bin = int((centroid[axis] - lo) * scale)
};
struct SdLists {
  SdWork*   next;        // next level's work list
  uint32_t* nextCount;
  SdWork*   small;       // nodes of <= SD_SMALL triangles (finished by sd_small)
  uint32_t* smallCount;
};

SD_FN float sd_area2(const float* lo, const float* hi)  // 2 x half area; empty box -> 0   (pt_sah.hip Box::area)
{
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return (dx < 0.f || dy < 0.f || dz < 0.f) ? 0.f : 2.0f * (dx * dy + dy * dz + dz * dx);
}
SD_FN int sd_bin_of(float c, float lo, float scale)
{
  int k = int((c - lo) * scale);
  return k < 0 ? 0 : (k >= SD_BINS ? SD_BINS - 1 : k);
}
SD_FN void sd_init_work(SdWork& w, uint32_t first, uint32_t count, uint32_t node)
{
  w.first = first; w.count = count; w.node = node;
  for(int a = 0; a < 3; ++a) { w.cbLo[a] = SD_ORD_PLUS_INF; w.cbHi[a] = SD_ORD_MINUS_INF; }
  w.axis = -1; w.kSplit = 0; w.nl = 0; w.leftW = w.rightW = SD_NONE; w.curL = w.curR = 0; w.lo = 0.f; w.scale = 0.f;
}

// prim boxes from the world-space triangle records (pt_sah.hip pt_sah_topology: non-finite input collapses to the origin)
SD_FN void sd_prim(uint32_t i, const TriRec* tris, float4* plo, float4* phi)
{
  const TriRec t = tris[i];
  const float  p0[3] = {t.p0w.x, t.p0w.y, t.p0w.z};
  const float  p1[3] = {t.p0w.x + t.e1n.x, t.p0w.y + t.e1n.y, t.p0w.z + t.e1n.z};
  const float  p2[3] = {t.p0w.x + t.e2p.x, t.p0w.y + t.e2p.y, t.p0w.z + t.e2p.z};
  float        lo[3], hi[3];
  for(int a = 0; a < 3; ++a)
  {
    lo[a] = fminf(p0[a], fminf(p1[a], p2[a]));
    hi[a] = fmaxf(p0[a], fmaxf(p1[a], p2[a]));
    const bool finite = (sd_fbits(lo[a]) & 0x7f800000u) != 0x7f800000u && (sd_fbits(hi[a]) & 0x7f800000u) != 0x7f800000u;
    if(!finite)
      lo[a] = hi[a] = 0.0f;
  }
  plo[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
  phi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

SD_FN void sd_cbounds(uint32_t pos, const uint32_t* idx, const uint32_t* primWork, SdWork* work, const float4* plo, const float4* phi)
{
  const uint32_t w = primWork[pos];
  if(w == SD_NONE)
    return;
  const uint32_t p  = idx[pos];
  const float4   lo = plo[p], hi = phi[p];
  const float    c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
  for(int a = 0; a < 3; ++a)
  {
    sd_min(&work[w].cbLo[a], sd_order(c[a]));
    sd_max(&work[w].cbHi[a], sd_order(c[a]));
  }
}

// bins: binCnt[w][axis][bin], binBox[w][axis][bin][6] (lo xyz, hi xyz as ordered bits; initialised to +inf / -inf)
SD_FN void sd_bin(uint32_t pos, const uint32_t* idx, const uint32_t* primWork, const SdWork* work, const float4* plo, const float4* phi, uint32_t* binCnt, uint32_t* binBox)
{
  const uint32_t w = primWork[pos];
  if(w == SD_NONE)
    return;
  const uint32_t p  = idx[pos];
  const float4   lo = plo[p], hi = phi[p];
  const float    l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
  for(int a = 0; a < 3; ++a)
  {
    const float cl = sd_unorder(work[w].cbLo[a]), ch = sd_unorder(work[w].cbHi[a]);
    const float ext = ch - cl;
    if(!(ext > 0.f))
      continue;
    const int       k = sd_bin_of(0.5f * (l[a] + h[a]), cl, float(SD_BINS) / ext);
    const size_t    b = (size_t(w) * 3 + a) * SD_BINS + k;
    sd_add(&binCnt[b], 1u);
    for(int q = 0; q < 3; ++q)
    {
      sd_min(&binBox[b * 6 + q], sd_order(l[q]));
      sd_max(&binBox[b * 6 + 3 + q], sd_order(h[q]));
    }
  }
}

// one thread per open node (pt_sah.hip Builder::split, binned branch, + the child numbering of Builder::build)
SD_FN void sd_split(uint32_t w, SdWork* work, const uint32_t* binCnt, const uint32_t* binBox, SdLists L, uint32_t* childL, uint32_t* childR, uint32_t* parI, uint32_t* parL)
{
  SdWork&        W = work[w];
  const uint32_t count = W.count;
  float          bestCost = 3.0e38f;
  int            bestAxis = -1, bestK = 0;
  for(int a = 0; a < 3; ++a)
  {
    const float cl = sd_unorder(W.cbLo[a]), ch = sd_unorder(W.cbHi[a]);
    if(!(ch - cl > 0.f))
      continue;
    const uint32_t* cnt = binCnt + (size_t(w) * 3 + a) * SD_BINS;
    const uint32_t* box = binBox + (size_t(w) * 3 + a) * SD_BINS * 6;
    float           ra[SD_BINS];
    uint32_t        rn[SD_BINS];
    float           rl[3] = {3.0e38f, 3.0e38f, 3.0e38f}, rh[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t        n = 0;
    for(int k = SD_BINS - 1; k >= 1; --k)
    {
      if(cnt[k])
        for(int q = 0; q < 3; ++q)
        {
          rl[q] = fminf(rl[q], sd_unorder(box[k * 6 + q]));
          rh[q] = fmaxf(rh[q], sd_unorder(box[k * 6 + 3 + q]));
        }
      n += cnt[k];
      ra[k] = sd_area2(rl, rh);
      rn[k] = n;
    }
    float ll[3] = {3.0e38f, 3.0e38f, 3.0e38f}, lh[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    n = 0;
    for(int k = 1; k < SD_BINS; ++k)
    {
      if(cnt[k - 1])
        for(int q = 0; q < 3; ++q)
        {
          ll[q] = fminf(ll[q], sd_unorder(box[(k - 1) * 6 + q]));
          lh[q] = fmaxf(lh[q], sd_unorder(box[(k - 1) * 6 + 3 + q]));
        }
      n += cnt[k - 1];
      if(n == 0 || rn[k] == 0)
        continue;
      const float cost = sd_area2(ll, lh) * float(n) + ra[k] * float(rn[k]);
      if(cost < bestCost)
      {
        bestCost = cost;
        bestAxis = a;
        bestK    = k;
      }
    }
  }
  uint32_t nl = 0;
  if(bestAxis >= 0)
  {
    const uint32_t* cnt = binCnt + (size_t(w) * 3 + bestAxis) * SD_BINS;
    for(int k = 0; k < bestK; ++k)
      nl += cnt[k];
  }
  if(bestAxis < 0 || nl == 0 || nl >= count)
  {  // all centroids coincide (or binning failed): split the range in the middle
    bestAxis = -1;
    nl       = count / 2;
  }
  else
  {
    const float cl = sd_unorder(W.cbLo[bestAxis]), ch = sd_unorder(W.cbHi[bestAxis]);
    W.lo    = cl;
    W.scale = float(SD_BINS) / (ch - cl);
  }
  W.axis   = bestAxis;
  W.kSplit = bestK;
  W.nl     = nl;
  const uint32_t nr = count - nl, node = W.node, first = W.first;
  const uint32_t leftId = node + 1, rightId = node + nl;  // the left subtree owns nl-1 ids after `node`
  auto           child = [&](uint32_t f, uint32_t c, uint32_t id, uint32_t* childArr, uint32_t& slotW) {
    if(c == 1)
    {
      childArr[node] = f | BVH_LEAF;
      parL[f]        = node;
      return;
    }
    childArr[node] = id;
    parI[id]       = node;
    if(c <= SD_SMALL)
    {
      SdWork s;
      sd_init_work(s, f, c, id);
      L.small[sd_add(L.smallCount, 1u)] = s;
    }
    else
    {
      SdWork nx;
      sd_init_work(nx, f, c, id);
      slotW          = sd_add(L.nextCount, 1u);
      L.next[slotW]  = nx;
    }
  };
  child(first, nl, leftId, childL, W.leftW);
  child(first + nl, nr, rightId, childR, W.rightW);
}

// (also accumulates the centroid bounds of the child the triangle goes to, so that only the root needs a sd_cbounds pass)
SD_FN void sd_partition(uint32_t pos, const uint32_t* idxIn, const uint32_t* primWorkIn, SdWork* work, SdWork* next, const float4* plo, const float4* phi, uint32_t* idxOut,
                        uint32_t* primWorkOut)
{
  const uint32_t w = primWorkIn[pos];
  const uint32_t p = idxIn[pos];
  if(w == SD_NONE)
  {  // finished ranges keep their place
    idxOut[pos]      = p;
    primWorkOut[pos] = SD_NONE;
    return;
  }
  SdWork& W = work[w];
  bool    left;
  if(W.axis < 0)
    left = pos < W.first + W.nl;
  else
  {
    const float4 lo = plo[p], hi = phi[p];
    const float  c = W.axis == 0 ? 0.5f * (lo.x + hi.x) : (W.axis == 1 ? 0.5f * (lo.y + hi.y) : 0.5f * (lo.z + hi.z));
    left           = sd_bin_of(c, W.lo, W.scale) < W.kSplit;
  }
  const uint32_t dst = left ? W.first + sd_add(&W.curL, 1u) : W.first + W.nl + sd_add(&W.curR, 1u);
  const uint32_t cw  = left ? W.leftW : W.rightW;
  idxOut[dst]        = p;
  primWorkOut[dst]   = cw;
  if(cw != SD_NONE)
  {
    const float4 lo = plo[p], hi = phi[p];
    const float  c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
    for(int a = 0; a < 3; ++a)
    {
      sd_min(&next[cw].cbLo[a], sd_order(c[a]));
      sd_max(&next[cw].cbHi[a], sd_order(c[a]));
    }
  }
}

// one thread per small node: the whole subtree over <= SD_SMALL triangles by exact sweeps (pt_sah.hip Builder::split, sweep branch)
SD_FN void sd_small(uint32_t s, const SdWork* small, uint32_t* idx, const float4* plo, const float4* phi, uint32_t* childL, uint32_t* childR, uint32_t* parI, uint32_t* parL)
{
  const SdWork S = small[s];
  uint32_t     id[SD_SMALL];
  float        lo[SD_SMALL][3], hi[SD_SMALL][3], cen[SD_SMALL][3];
  const uint32_t total = S.count;
  for(uint32_t i = 0; i < total; ++i)
  {
    id[i]           = idx[S.first + i];
    const float4 l = plo[id[i]], h = phi[id[i]];
    lo[i][0] = l.x; lo[i][1] = l.y; lo[i][2] = l.z; hi[i][0] = h.x; hi[i][1] = h.y; hi[i][2] = h.z;
    for(int a = 0; a < 3; ++a)
      cen[i][a] = 0.5f * (lo[i][a] + hi[i][a]);
  }
  // local permutation `ord` of [0, total): the subtree is built by sorting sub-ranges of it
  uint32_t ord[SD_SMALL];
  for(uint32_t i = 0; i < total; ++i)
    ord[i] = i;
  struct Item { uint32_t f, c, node; };
  Item stack[SD_SMALL];
  int  sp       = 0;
  stack[sp++]   = Item{0u, total, S.node};
  auto sort_by  = [&](uint32_t f, uint32_t c, int a) {  // insertion sort by (centroid[a], id)
    for(uint32_t i = f + 1; i < f + c; ++i)
    {
      const uint32_t v = ord[i];
      uint32_t       j = i;
      while(j > f && (cen[ord[j - 1]][a] > cen[v][a] || (cen[ord[j - 1]][a] == cen[v][a] && id[ord[j - 1]] > id[v])))
      {
        ord[j] = ord[j - 1];
        --j;
      }
      ord[j] = v;
    }
  };
  while(sp)
  {
    const Item it = stack[--sp];
    uint32_t   k  = 1;
    if(it.c > 2)
    {
      float bestCost = 3.0e38f;
      int   bestAxis = -1;
      float rightArea[SD_SMALL];
      for(int a = 0; a < 3; ++a)
      {
        sort_by(it.f, it.c, a);
        float bl[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bh[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for(uint32_t i = it.c; i-- > 1;)
        {
          for(int q = 0; q < 3; ++q)
          {
            bl[q] = fminf(bl[q], lo[ord[it.f + i]][q]);
            bh[q] = fmaxf(bh[q], hi[ord[it.f + i]][q]);
          }
          rightArea[i] = sd_area2(bl, bh);
        }
        float ll[3] = {3.0e38f, 3.0e38f, 3.0e38f}, lh[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for(uint32_t j = 1; j < it.c; ++j)
        {
          for(int q = 0; q < 3; ++q)
          {
            ll[q] = fminf(ll[q], lo[ord[it.f + j - 1]][q]);
            lh[q] = fmaxf(lh[q], hi[ord[it.f + j - 1]][q]);
          }
          const float cost = sd_area2(ll, lh) * float(j) + rightArea[j] * float(it.c - j);
          if(cost < bestCost)
          {
            bestCost = cost;
            bestAxis = a;
            k        = j;
          }
        }
      }
      if(bestAxis != 2)  // the range is currently sorted along axis 2
        sort_by(it.f, it.c, bestAxis < 0 ? 0 : bestAxis);
    }
    else
      sort_by(it.f, it.c, 0);
    const uint32_t nl = k, nr = it.c - k;
    const uint32_t leftId = it.node + 1, rightId = it.node + nl;
    if(nl == 1)
    {
      childL[it.node]        = (S.first + it.f) | BVH_LEAF;
      parL[S.first + it.f]   = it.node;
    }
    else
    {
      childL[it.node] = leftId;
      parI[leftId]    = it.node;
      stack[sp++]     = Item{it.f, nl, leftId};
    }
    if(nr == 1)
    {
      childR[it.node]           = (S.first + it.f + nl) | BVH_LEAF;
      parL[S.first + it.f + nl] = it.node;
    }
    else
    {
      childR[it.node] = rightId;
      parI[rightId]   = it.node;
      stack[sp++]     = Item{it.f + nl, nr, rightId};
    }
  }
  for(uint32_t i = 0; i < total; ++i)
    idx[S.first + i] = id[ord[i]];
}
