// The acceleration structure the traversal kernels walk: an 8-wide BVH whose child boxes are quantised to an 11-bit grid per node and stored as
// fp16 integers (after Ylitie, Karras, Laine: "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017 -- their 8-bit
// planes and octant-ordered child slots, adapted to what is cheap on gfx950).
//
// Why (measured on the 4-wide fp32 layout of rounds 1-2, profiles/r03_valu.json / r03_cache.json): the trace kernels spend ~60 % of their wave
// cycles parked on memory -- a wavefront waits for the slowest of its 64 lanes -- and the rest on ~37 VALU instructions per child box.  Here
// one node visit decides 8 children from 128 B (eight aligned 16-byte loads per lane, all in flight together; the 4-wide layout spent the same
// bytes on 4 children), a ray needs ~0.65x the dependent round trips, and a plane costs ONE instruction: v_fma_mix_f32 reads the fp16 grid
// coordinate straight out of a register half and multiplies it with the fp32 grid step (an 8-bit plane costs a conversion and an FMA;
// first measured that way: 1.7x the VALU work per ray of the 4-wide layout and 10 % slower, profiles/r03c_variants_*.txt).
//
//   CwNode, 128 B:
//     p[3]          fp32 origin of the node's quantisation grid (lower corner of the union of its children's boxes)
//     e[3]          one exponent byte per axis: grid step 2^(e - 127); 2047 steps cover the node's extent
//     imask         bit k: child slot k is an inner node
//     childBase     24 bits: index of the first inner child (the inner children of a node are consecutive nodes, in slot order)
//                   | amask << 24: bit k: slot k holds non-opaque triangles (leaf) / its subtree does (inner) -- the alpha-only walks skip the rest
//     triBase       leaf slot of the node's first triangle (the triangles of a node's leaf children are consecutive records, in slot order)
//     l1 | l2 << 8  bit k of l1: slot k is a leaf; of l2: that leaf holds two triangles (else one)
//     qlo/qhi[3][8] the child boxes on the grid as fp16 integers 0 .. 2047: lower planes rounded down, upper planes rounded up (conservative
//                   by construction: the builder checks p + q 2^e against the fp32 box in double precision); empty slots: inverted (2047, 0)
//
// Child slots are not arbitrary: a child is put into the slot whose sign pattern (bit 0 / 1 / 2 = x / y / z on the + side of the node centre)
// matches where it lies, so that visiting the hit slots in increasing (slot XOR ray-direction signs) is a front-to-back order that costs no
// sorting (cw_assign_slots) -- over ALL children, leaves and inner nodes alike: a leaf's triangles are tested when its turn comes, not before
// nearer inner children (testing a node's leaves first cost 2x the triangle tests, tools/steps_experiment.py).
//
// The functions below are plain code of (inputs) -> (node) shared by the device builder (pt_accel.hip k_collapse8), its host emulation
// (pt_debug_cw_collapse, used by the CPU tests) and nothing else.
#pragma once
#include <stdint.h>
#include <string.h>
#include "pt_device.h"

#if defined(__HIPCC__)
#define CW_FN __host__ __device__ inline
#else
#define CW_FN static inline
#endif

#define CW_WIDTH 8
#define CW_LEAF_MAX 2          // triangles per leaf child (both records are fetched together)
#define CW_GRID_MAX 2047       // largest grid coordinate (the fp16 integers 0 .. 2048 are exact)
#define CW_EXP_MIN 27          // exponent byte floor (2^-100): ray-side products with it never underflow, so an inverted (empty) box can never test as hit
#define CW_NODE_BYTES 128
#define CW_CHILD_MASK 0x00ffffffu

struct CwNode {
  float    p[3];
  uint32_t eimask;     // e[0] | e[1] << 8 | e[2] << 16 | imask << 24
  uint32_t childBase;  // first inner child | amask << 24
  uint32_t triBase;
  uint32_t leaves;     // l1 | l2 << 8
  uint32_t _pad;
  uint16_t qlox[8], qloy[8], qloz[8];  // fp16 bit patterns, slot k at [k]; 16 B per plane set: the ray's direction signs pick near / far by ADDRESS
  uint16_t qhix[8], qhiy[8], qhiz[8];
};
static_assert(sizeof(CwNode) == CW_NODE_BYTES, "CwNode is eight 16-byte loads");
#define CW_OFF_QLO 32u  // byte offset of qlox; qloy, qloz follow at +16, +32
#define CW_OFF_QHI 80u  // byte offset of qhix

// fp16 bit pattern of the integer q (0 .. 2048): exact
CW_FN uint16_t cw_half_of_int(uint32_t q)
{
  if(q == 0)
    return 0;
  int e = 0;
  while((q >> (e + 1)) != 0)
    ++e;
  return uint16_t(((uint32_t(e) + 15u) << 10) | ((q << (10 - e)) & 0x3ffu));
}
// and back (normal, non-negative halves only: what the nodes hold)
CW_FN float cw_float_of_half(uint32_t h)
{
  h &= 0xffffu;
  if(h == 0)
    return 0.0f;
  const uint32_t bits = (((h >> 10) + 112u) << 23) | ((h & 0x3ffu) << 13);
  float          f;
  memcpy(&f, &bits, 4);
  return f;
}

// what a collapse step knows about one child of the node it emits
struct CwChild {
  float    lo[3], hi[3];
  uint32_t kind;      // 0: inner node, 1..leafMax: leaf with that many triangles
  uint32_t alpha;     // non-opaque triangles in it / below it
};

CW_FN uint32_t cw_fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
CW_FN float    cw_bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// Slot assignment (see the header comment): greedy over the 8 x 8 table of dot(child centre - node centre, sign pattern of the slot), largest first.
// slotOf[c] = slot of child c.  Deterministic (ties: lowest child, lowest slot).
CW_FN void cw_assign_slots(const CwChild* ch, int n, const float nlo[3], const float nhi[3], int slotOf[CW_WIDTH])
{
  float cost[CW_WIDTH][CW_WIDTH];
  const float cx = 0.5f * (nlo[0] + nhi[0]), cy = 0.5f * (nlo[1] + nhi[1]), cz = 0.5f * (nlo[2] + nhi[2]);
  for(int c = 0; c < n; ++c)
  {
    const float dx = 0.5f * (ch[c].lo[0] + ch[c].hi[0]) - cx, dy = 0.5f * (ch[c].lo[1] + ch[c].hi[1]) - cy, dz = 0.5f * (ch[c].lo[2] + ch[c].hi[2]) - cz;
    for(int s = 0; s < CW_WIDTH; ++s)
      cost[c][s] = ((s & 1) ? dx : -dx) + ((s & 2) ? dy : -dy) + ((s & 4) ? dz : -dz);
  }
  uint32_t usedC = 0, usedS = 0;
  for(int c = 0; c < CW_WIDTH; ++c)
    slotOf[c] = -1;
  for(int round = 0; round < n; ++round)
  {
    int   bc = -1, bs = -1;
    float best = 0.f;
    for(int c = 0; c < n; ++c)
      if(!((usedC >> c) & 1u))
        for(int s = 0; s < CW_WIDTH; ++s)
          if(!((usedS >> s) & 1u) && (bc < 0 || cost[c][s] > best))
          {
            best = cost[c][s];
            bc   = c;
            bs   = s;
          }
    slotOf[bc] = bs;
    usedC |= 1u << bc;
    usedS |= 1u << bs;
  }
}

// Grid of one axis: origin p (fp32), exponent byte e with 2^(e-127) * 2047 >= extent.  Quantisation runs in double so that "the decoded plane
// encloses the fp32 plane" is exact arithmetic, not an argument about rounding.
CW_FN uint32_t cw_axis_exponent(float lo, float hi)
{
  const double ext = double(hi) - double(lo);
  int          e   = CW_EXP_MIN;
  // smallest e with 2047 * 2^(e-127) >= ext
  while(e < 254 && double(CW_GRID_MAX) * ldexp(1.0, e - 127) < ext)
    ++e;
  return uint32_t(e);
}
CW_FN uint32_t cw_quant_lo(float v, float p, uint32_t e)
{
  const double step = ldexp(1.0, int(e) - 127);
  double       q    = floor((double(v) - double(p)) / step);
  if(q < 0.0) q = 0.0;
  if(q > double(CW_GRID_MAX)) q = double(CW_GRID_MAX);
  while(q > 0.0 && double(p) + q * step > double(v))  // (never taken for finite inputs: floor is exact in double here; kept as the stated invariant)
    q -= 1.0;
  return uint32_t(q);
}
CW_FN uint32_t cw_quant_hi(float v, float p, uint32_t e)
{
  const double step = ldexp(1.0, int(e) - 127);
  double       q    = ceil((double(v) - double(p)) / step);
  if(q < 0.0) q = 0.0;
  if(q > double(CW_GRID_MAX)) q = double(CW_GRID_MAX);
  return uint32_t(q);
}

// Writes the node for `n` children already placed: child c sits in slot slotOf[c].  Inner children get consecutive node ids from childBase in
// slot order, leaf children consecutive triangle offsets from triBase in slot order: innerRank[c] / triOffset[c] tell the caller which.
// Returns false if the boxes cannot be represented (non-finite input).
CW_FN bool cw_encode_node(const CwChild* ch, int n, const int slotOf[CW_WIDTH], uint32_t childBase, uint32_t triBase, CwNode* out, uint32_t innerRank[CW_WIDTH],
                          uint32_t triOffset[CW_WIDTH])
{
  float nlo[3] = {ch[0].lo[0], ch[0].lo[1], ch[0].lo[2]}, nhi[3] = {ch[0].hi[0], ch[0].hi[1], ch[0].hi[2]};
  for(int c = 1; c < n; ++c)
    for(int a = 0; a < 3; ++a)
    {
      nlo[a] = ch[c].lo[a] < nlo[a] ? ch[c].lo[a] : nlo[a];
      nhi[a] = ch[c].hi[a] > nhi[a] ? ch[c].hi[a] : nhi[a];
    }
  for(int a = 0; a < 3; ++a)
    if(!(nlo[a] <= nhi[a]) || !(nhi[a] - nlo[a] < 3.0e38f))
      return false;
  uint32_t e[3];
  for(int a = 0; a < 3; ++a)
  {
    e[a] = cw_axis_exponent(nlo[a], nhi[a]);
    // the upper planes must fit into 2047 steps counted from p: bump the exponent if rounding up overflows the grid
    for(;;)
    {
      bool ok = true;
      for(int c = 0; c < n && ok; ++c)
        ok = double(nlo[a]) + double(CW_GRID_MAX) * ldexp(1.0, int(e[a]) - 127) >= double(ch[c].hi[a]);
      if(ok || e[a] >= 254)
        break;
      ++e[a];
    }
  }
  CwNode nd;
  memset(&nd, 0, sizeof(nd));
  nd.p[0] = nlo[0]; nd.p[1] = nlo[1]; nd.p[2] = nlo[2];
  uint16_t* qlo[3] = {nd.qlox, nd.qloy, nd.qloz};
  uint16_t* qhi[3] = {nd.qhix, nd.qhiy, nd.qhiz};
  for(int s = 0; s < CW_WIDTH; ++s)
    for(int a = 0; a < 3; ++a)
    {
      qlo[a][s] = cw_half_of_int(CW_GRID_MAX);  // inverted box: an empty slot is never hit (and if it were, it is neither inner nor leaf: no work)
      qhi[a][s] = cw_half_of_int(0);
    }
  uint32_t imask = 0, amask = 0, l1 = 0, l2 = 0;
  int      childInSlot[CW_WIDTH];
  for(int s = 0; s < CW_WIDTH; ++s)
    childInSlot[s] = -1;
  for(int c = 0; c < n; ++c)
    childInSlot[slotOf[c]] = c;
  uint32_t nInner = 0, nTri = 0;
  for(int s = 0; s < CW_WIDTH; ++s)
  {
    const int c = childInSlot[s];
    if(c < 0)
      continue;
    for(int a = 0; a < 3; ++a)
    {
      qlo[a][s] = cw_half_of_int(cw_quant_lo(ch[c].lo[a], nlo[a], e[a]));
      qhi[a][s] = cw_half_of_int(cw_quant_hi(ch[c].hi[a], nlo[a], e[a]));
    }
    if(ch[c].alpha)
      amask |= 1u << s;
    if(ch[c].kind == 0)
    {
      imask |= 1u << s;
      innerRank[c] = nInner++;
    }
    else
    {
      l1 |= 1u << s;
      if(ch[c].kind == 2)
        l2 |= 1u << s;
      triOffset[c] = nTri;
      nTri += ch[c].kind;
    }
  }
  nd.eimask    = e[0] | (e[1] << 8) | (e[2] << 16) | (imask << 24);
  nd.childBase = (childBase & CW_CHILD_MASK) | (amask << 24);
  nd.triBase   = triBase;
  nd.leaves    = l1 | (l2 << 8);
  *out = nd;
  return true;
}

// ---- collapse: binary tree -> CwNodes ----------------------------------------------------------------------------------------------------
// One work item = (binary node to open, CwNode id it becomes).  The binary tree is the builder's product (BvhNode: both child boxes in the
// parent, child references with BVH_LEAF / BVH_ALPHA tags, d.z / d.w = number of triangles below the left / right child).
//   1. open the child of largest surface area that is an inner binary node with more than leafMax triangles, until 8 children;
//   2. slots left over: open multi-triangle leaves-to-be (largest area first) -- a box test per triangle costs nothing extra in an 8-wide node;
//   3. what remains inner with <= leafMax triangles becomes ONE leaf child holding those triangles.
// leafMax: CW_LEAF_MAX for triangles, 1 for the TLAS (a lane enters one instance at a time).
struct CwItem {
  uint32_t b2;    // binary node (index into the BvhNode array)
  uint32_t node;  // CwNode it becomes
};
struct CwOpen {  // a child during the collapse
  uint32_t ref;   // binary child reference (BVH_LEAF | slot [| BVH_ALPHA], or inner index [| BVH_ALPHA])
  uint32_t count; // triangles below it
  float    lo[3], hi[3];
};
CW_FN float cw_half_area(const CwOpen& o)
{
  const float dx = o.hi[0] - o.lo[0], dy = o.hi[1] - o.lo[1], dz = o.hi[2] - o.lo[2];
  return dx * dy + dy * dz + dz * dx;
}
CW_FN int cw_open_children(const BvhNode* b2, uint32_t node, CwOpen* out)
{
  const BvhNode nd = b2[node & BVH_SLOT_MASK];
  out[0].ref = nd.d.x; out[0].count = nd.d.z;
  out[0].lo[0] = nd.a.x; out[0].lo[1] = nd.a.y; out[0].lo[2] = nd.a.z; out[0].hi[0] = nd.a.w; out[0].hi[1] = nd.b.x; out[0].hi[2] = nd.b.y;
  if(nd.d.y == BVH_NONE)
    return 1;
  out[1].ref = nd.d.y; out[1].count = nd.d.w;
  out[1].lo[0] = nd.b.z; out[1].lo[1] = nd.b.w; out[1].lo[2] = nd.c.x; out[1].hi[0] = nd.c.y; out[1].hi[1] = nd.c.z; out[1].hi[2] = nd.c.w;
  return 2;
}
// the (old, binary-tree order) leaf slots below `ref`, at most CW_LEAF_MAX of them, left to right
CW_FN int cw_leaf_slots(const BvhNode* b2, uint32_t ref, uint32_t* slots)
{
  uint32_t stack[2 * CW_LEAF_MAX + 2];
  int      sp = 0, n = 0;
  stack[sp++] = ref;
  while(sp)
  {
    const uint32_t r = stack[--sp];
    if(r & BVH_LEAF)
    {
      if(n < CW_LEAF_MAX)
        slots[n] = r & BVH_SLOT_MASK;
      ++n;
      continue;
    }
    const BvhNode nd = b2[r & BVH_SLOT_MASK];
    if(nd.d.y != BVH_NONE && sp < 2 * CW_LEAF_MAX + 1)
      stack[sp++] = nd.d.y;
    if(sp < 2 * CW_LEAF_MAX + 2)
      stack[sp++] = nd.d.x;
  }
  return n;
}
// Gathers the (up to 8) children of work item `it`.  Returns their number.
CW_FN int cw_gather_children(const BvhNode* b2, uint32_t b2node, CwOpen* ch, uint32_t leafMax)
{
  int n = cw_open_children(b2, b2node, ch);
  for(int phase = 0; phase < 2; ++phase)
    while(n < CW_WIDTH)
    {
      int   best  = -1;
      float bestA = -1.f;
      for(int k = 0; k < n; ++k)
      {
        const bool inner = !(ch[k].ref & BVH_LEAF);
        if(inner && (phase == 0 ? ch[k].count > leafMax : true))
        {
          const float a = cw_half_area(ch[k]);
          if(a > bestA)
          {
            bestA = a;
            best  = k;
          }
        }
      }
      if(best < 0)
        break;
      const uint32_t node = ch[best].ref;
      ch[best]            = ch[n - 1];
      --n;
      n += cw_open_children(b2, node, ch + n);
    }
  return n;
}
