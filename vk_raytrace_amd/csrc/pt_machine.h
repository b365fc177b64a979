// Ray supply of the refilling "trace machine" (the persistent wavefronts of pt_render.hip: k_closest_p, k_shadow_p, k_tail).
//
// Measured motivation (profiles/r01_b_pmc_*.txt): with one ray per lane for the lifetime of a wavefront, the
// traversal kernels ran at 23 % (closest hit) and 8 % (shadow) SIMD lane utilisation -- a wave lasts as long as
// its longest ray (rays differ by >10x in node count; alpha-tested rays need a second pass).  There a wavefront
// is persistent: every lane carries an explicit traversal state (TraceLane, pt_trace.h), and as soon as fewer than REFILL_BELOW lanes
// are still running, the idle lanes pull new rays from the queue (one atomic per CHUNK rays per wave) and the
// loop continues.  The alpha count pass (pass B of pt_trace.h) is just another state of the same lane, so a
// lane that needs it does not stall the other 63.
#pragma once
#include "pt_trace.h"

// refill threshold: PT_REFILL_BELOW_DEFAULT in pt_internal.h (lanes still running below which idle lanes pull new rays)

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
