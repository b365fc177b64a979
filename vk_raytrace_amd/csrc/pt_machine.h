// Resumable per-lane BVH traversal ("trace machine") for persistent wavefronts on gfx950.
//
// Measured motivation (profiles/r01_b_pmc_*.txt): with one ray per lane for the lifetime of a wavefront, the
// traversal kernels ran at 23 % (closest hit) and 8 % (shadow) SIMD lane utilisation -- a wave lasts as long as
// its longest ray (rays differ by >10x in node count; alpha-tested rays need a second pass).  Here a wavefront
// is persistent: every lane carries an explicit traversal state, and as soon as fewer than REFILL_BELOW lanes
// are still running, the idle lanes pull new rays from the queue (one atomic per CHUNK rays per wave) and the
// loop continues.  The alpha count pass (pass B of pt_trace.h) is just another state of the same lane, so a
// lane that needs it does not stall the other 63.
//
// Semantics are exactly those of traverse<TM_CLOSEST / TM_SHADOW / TM_COUNT> in pt_trace.h (same arithmetic,
// same candidate rules); rays that need the exact key-ordered fallback are handed to the simple kernels.
#pragma once
#include "pt_trace.h"

// refill threshold: PT_REFILL_BELOW_DEFAULT in pt_internal.h (lanes still running below which idle lanes pull new rays)

struct TraceLane {
  f3       o, d;
#if PT_BVH_WIDTH != 2
  RayBox   rbox;
#else
  f3       idir;
#endif
  float    tmax;           // exclusive upper bound on t (pass B: the limit key's t, inclusive for ties)
  float    bt, bu, bv;     // best CERTAIN hit
  uint32_t bslot, bw;
  uint32_t cur;
  int      sp;
  uint32_t flags, cnt, wLimit;
  float    zeroMaxT, zeroMaxT2, zeroMaxT3;  // pass A: the three largest t among the zero-opacity candidates seen
  int      pass;           // 0: pass A (nearest certain hit), 1: pass B (count zero-opacity candidates in front of it)
  bool     opaqueHit;      // shadow rays: an opaque occluder was found
  bool     done;
#if PT_BVH_WIDTH != 2
  InstCtx  ic;             // two-level instantiations only (the flat ones never touch it): the instance the lane is inside of
  uint32_t steps;          //   and the loop-iteration guard
#endif
};

PT_DEV void lane_begin(TraceLane& L, f3 o, f3 d, float tmax, bool emptyScene)
{
  L.o = o; L.d = d;
#if PT_BVH_WIDTH != 2
  L.rbox = make_raybox(o, d);
#else
  L.idir = f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
#endif
  L.tmax = tmax; L.bt = tmax; L.bu = 0.f; L.bv = 0.f; L.bslot = BVH_NONE; L.bw = 0xffffffffu;
  L.cur = 0; L.sp = 0; L.flags = 0; L.cnt = 0; L.wLimit = 0; L.pass = 0; L.opaqueHit = false; L.done = emptyScene; L.zeroMaxT = -1.0f; L.zeroMaxT2 = -1.0f; L.zeroMaxT3 = -1.0f;
#if PT_BVH_WIDTH != 2
  L.ic = InstCtx{BVH_NONE, 0, 0u}; L.steps = 0;
#endif
}
// pass B over the candidates with key < (best hit | ray end)
template <bool TWO = false>
PT_DEV void lane_begin_count(TraceLane& L)
{
  const bool found = L.bslot != BVH_NONE;
  L.wLimit = found ? (L.bw & TRI_INDEX_MASK) : 0u;
  L.tmax   = found ? L.bt : L.tmax;
  L.cur = 0; L.sp = 0; L.flags = 0; L.cnt = 0; L.pass = 1; L.done = false;
#if PT_BVH_WIDTH != 2
  if(TWO)
  {  // pass A may have ended inside an instance
    L.ic.inst = BVH_NONE;
    L.rbox    = make_raybox(L.o, L.d);
  }
#endif
}

template <bool TWO = false>
PT_DEV void lane_pop(TraceLane& L, const uint32_t* lds, const uint32_t* spill)
{
#if PT_BVH_WIDTH != 2
  if(TWO && L.ic.inst != BVH_NONE && L.sp == L.ic.spBase)
  {  // the instance's subtree is exhausted: back to TLAS level.  The world-space ray constants are recomputed rather than kept
     // (12 VGPRs for the lifetime of the lane against ~40 instructions per instance visit)
    L.ic.inst = BVH_NONE;
    L.rbox    = make_raybox(L.o, L.d);
  }
#endif
  if(L.sp == 0)
  {
    L.done = true;
    return;
  }
  --L.sp;
  L.cur = L.sp < STACK_LDS ? lds[L.sp * TRACE_BLOCK] : spill[L.sp - STACK_LDS];
}

// One inner-node visit.  SHADOW: true for shadow rays (they must keep looking for opaque triangles behind the best
// alpha candidate, so only tmax prunes).
template <bool SHADOW, bool TWO = false>
PT_DEV void lane_inner(const DeviceScene& S, TraceLane& L, uint32_t* lds, uint32_t* spill, Counters* counters)
{
#if PT_BVH_WIDTH != 2
  if(TWO && ++L.steps > PT_TWO_GUARD)
  {
    atomicAdd(&counters->stackOverflow, 1u);
    L.done = true;
    return;
  }
  const float    lim = (SHADOW || L.pass == 1) ? L.tmax : L.bt;
  auto pushChild = [&](uint32_t c) {
    if(L.sp < STACK_LDS)
      lds[L.sp++ * TRACE_BLOCK] = c;
    else if(L.sp < STACK_LDS + STACK_SPILL)
      spill[L.sp++ - STACK_LDS] = c;
    else
      atomicAdd(&counters->stackOverflow, 1u);
  };
  // compact nodes when the structure has them (wave-uniform choice): five requests per node instead of seven
  const bool     atTlas = TWO && L.ic.inst == BVH_NONE;
  const bool     alphaOnly = L.pass == 1;
  const uint32_t nxt    = S.cnodes ? wide_node_step_c(atTlas ? S.ctlas : S.cnodes, L.cur, L.rbox, lim, alphaOnly, pushChild)
                                   : wide_node_step(atTlas ? S.tlas : S.wide, L.cur, L.rbox, lim, alphaOnly, pushChild);
  if(nxt != BVH_NONE)
    L.cur = nxt;
  else
    lane_pop<TWO>(L, lds, spill);
}
#else
  const BvhNode* np = S.bvh + (L.cur & BVH_SLOT_MASK);
  const float4   a = np->a, b = np->b, c = np->c;
  const uint4    ch = np->d;
  const float    lim = (SHADOW || L.pass == 1) ? L.tmax : L.bt;
  const f3       o = L.o, idir = L.idir;
  float lx0 = (a.x - o.x) * idir.x, lx1 = (a.w - o.x) * idir.x;
  float ly0 = (a.y - o.y) * idir.y, ly1 = (b.x - o.y) * idir.y;
  float lz0 = (a.z - o.z) * idir.z, lz1 = (b.y - o.z) * idir.z;
  float rx0 = (b.z - o.x) * idir.x, rx1 = (c.y - o.x) * idir.x;
  float ry0 = (b.w - o.y) * idir.y, ry1 = (c.z - o.y) * idir.y;
  float rz0 = (c.x - o.z) * idir.z, rz1 = (c.w - o.z) * idir.z;
  float lnear = fmaxf(fmaxf(fminf(lx0, lx1), fminf(ly0, ly1)), fmaxf(fminf(lz0, lz1), 0.0f)) * 0.9999996f;
  float lfar  = fminf(fminf(fmaxf(lx0, lx1), fmaxf(ly0, ly1)), fminf(fmaxf(lz0, lz1), lim)) * 1.0000004f;
  float rnear = fmaxf(fmaxf(fminf(rx0, rx1), fminf(ry0, ry1)), fmaxf(fminf(rz0, rz1), 0.0f)) * 0.9999996f;
  float rfar  = fminf(fminf(fmaxf(rx0, rx1), fmaxf(ry0, ry1)), fminf(fmaxf(rz0, rz1), lim)) * 1.0000004f;
  bool  hl = lnear <= lfar, hr = (rnear <= rfar) && (ch.y != BVH_NONE);
  if(hl && hr)
  {
    uint32_t nearC = ch.x, farC = ch.y;
    if(rnear < lnear)
    {
      nearC = ch.y;
      farC  = ch.x;
    }
    if(L.sp < STACK_LDS)
      lds[L.sp++ * TRACE_BLOCK] = farC;
    else if(L.sp < STACK_LDS + STACK_SPILL)
      spill[L.sp++ - STACK_LDS] = farC;
    else
      atomicAdd(&counters->stackOverflow, 1u);
    L.cur = nearC;
  }
  else if(hl || hr)
    L.cur = hl ? ch.x : ch.y;
  else
    lane_pop(L, lds, spill);
}
#endif

template <bool SHADOW, bool TWO>
PT_DEV void lane_leaf_with(const DeviceScene& S, TraceLane& L, uint32_t slot, const TriRec& tr, const AlphaRec& ar, uint32_t* lds, uint32_t* spill);
// One leaf (triangle) visit; same candidate rules as traverse<TM_CLOSEST / TM_SHADOW / TM_COUNT>.
template <bool SHADOW, bool TWO = false>
PT_DEV void lane_leaf(const DeviceScene& S, TraceLane& L, uint32_t* lds, uint32_t* spill)
{
  const uint32_t slot  = L.cur & BVH_SLOT_MASK;
#if PT_BVH_WIDTH != 2
  if(TWO && L.ic.inst == BVH_NONE)
  {  // TLAS leaf: enter the instance (pt_trace.h: enter_instance)
    const TlasLeaf tl = S.tlasLeaves[slot];
    L.ic   = InstCtx{tl.inst, L.sp, tl.wflags};
    if(tl.inst != PT_INST_MERGED)
      L.rbox = enter_instance(S, tl, L.o, L.d);
    L.cur  = tl.nodeBase;
    return;
  }
#endif
  TriRec         tr    = S.tris[slot];
  AlphaRec       ar;
  if(L.cur & BVH_ALPHA)  // with the triangle, not after its test: fetching it only for candidates measured 4 % slower (one more round trip per candidate)
    ar = S.alphaRecs[slot];
#if PT_BVH_WIDTH != 2
  if(TWO && L.ic.inst != PT_INST_MERGED)
    tr = world_tri(S, L.ic, tr);
#endif
  lane_leaf_with<SHADOW, TWO>(S, L, slot, tr, ar, lds, spill);
}
// the triangle step on records already in registers
template <bool SHADOW, bool TWO>
PT_DEV void lane_leaf_with(const DeviceScene& S, TraceLane& L, uint32_t slot, const TriRec& tr, const AlphaRec& ar, uint32_t* lds, uint32_t* spill)
{
  const uint32_t wbits = __float_as_uint(tr.p0w.w);
  const uint32_t flags = wbits >> 29;
  const bool     opq   = (flags & TRI_OPAQUE) != 0;
  if(!(L.pass == 1 && opq))
  {
    float t, u, v;
    if(tri_test(tr, flags, L.o, L.d, t, u, v) && (L.pass == 1 ? t <= L.tmax : t < L.tmax) && t > 0.0f)
    {
      const uint32_t w = wbits & TRI_INDEX_MASK;
      if(L.pass == 1)
      {
        if(key_less(t, w, L.tmax, L.wLimit))
        {
          const float op = opacity_class(S, ar, u, v);
          if(op <= 0.0f)
            L.cnt++;
          else if(op < 1.0f)
            L.flags |= TF_SAW_FRAC;
        }
      }
      else if(SHADOW && opq)
      {
        L.opaqueHit = true;
        L.done      = true;
        return;
      }
      else if(L.bslot == BVH_NONE || key_less(t, w, L.bt, L.bw & TRI_INDEX_MASK))
      {
        bool certain = opq;
        if(!opq)
        {
          const float op = opacity_class(S, ar, u, v);
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
        }
      }
    }
  }
  lane_pop<TWO>(L, lds, spill);
}

// Wave-uniform ray supply: a wave reserves PT_CHUNK consecutive queue entries with one atomic and hands them to
// its idle lanes.  Returns the queue index for this lane or 0xffffffff.
struct RaySupply {
  uint32_t pos = 0, end = 0;
  uint32_t chunk = 64;  // rays a wave reserves per queue atomic
  bool     more = true;
};
PT_DEV uint32_t supply_next(RaySupply& rs, uint32_t* chunkCounter, uint32_t count, bool wants)
{
  uint32_t           got  = 0xffffffffu;
  const int          lane = threadIdx.x & 63;
  unsigned long long need = __ballot(wants && got == 0xffffffffu);
  while(need && rs.more)
  {
    if(rs.pos == rs.end)
    {
      uint32_t c = 0;
      if(lane == 0)
        c = atomicAdd(chunkCounter, 1u);
      c                       = __builtin_amdgcn_readfirstlane(c);
      const unsigned long long base = (unsigned long long)c * rs.chunk;
      if(base >= count)
      {
        rs.more = false;
        break;
      }
      rs.pos = uint32_t(base);
      rs.end = (base + rs.chunk < count) ? uint32_t(base + rs.chunk) : count;
    }
    const uint32_t avail = rs.end - rs.pos;
    const uint32_t want  = (uint32_t)__popcll(need);
    const uint32_t take  = avail < want ? avail : want;
    if(wants && got == 0xffffffffu)
    {
      const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
      if(rank < take)
        got = rs.pos + rank;
    }
    rs.pos += take;
    need = __ballot(wants && got == 0xffffffffu);
  }
  return got;
}
