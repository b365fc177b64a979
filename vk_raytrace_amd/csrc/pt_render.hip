// Wavefront path tracing pipeline for gfx950.
//
// The reference runs one megakernel invocation per pixel (shaders/pathtrace.comp:87-134 ->
// shaders/pathtrace.glsl:193-343).  Here the same per-pixel computation is cut at the two ray-trace
// calls into stages that communicate through an SoA path state in HBM and compacted queues of path
// slots, so that every stage runs with full 64-lane wavefronts of live paths:
//
//   k_generate      seed (pathtrace.comp:97), jitter + camera ray (pathtrace.glsl:348-372)      all paths of the frame batch
//   per bounce:
//   k_closest_k     bounce 0: ClosestHit incl. stochastic alpha (traceray_rq.glsl:108-147) as one packet traversal per
//                   wavefront (pt_packet.h); rays it cannot settle go to queueR                     queue[in] -> queueR
//   k_closest_p     bounce >= 1 and queueR: the same per lane, persistent wavefronts on the refilling
//                   trace machine (pt_machine.h)                                                   queue[in] | queueR
//   k_shade         miss/env, GetShadeState, material, emission, absorption, DirectLight,
//                   BSDF sample, throughput, next ray (pathtrace.glsl:201-325); Russian roulette
//                   right away for paths without a shadow ray                                    queue[in] -> queueS | queue[out]
//   k_shadow_p      AnyHit for the deferred NEE contribution, then Russian roulette
//                   (pathtrace.glsl:327-338); trace machine                                       queueS -> queue[out]
//   k_closest_x / k_shadow_x   one ray per lane, exact key-ordered alpha loop, on the (normally almost empty) queues of
//                   rays the two-pass scheme cannot settle
//   k_accumulate    firefly clamp (pathtrace.glsl:379-384) + running mean over the frames of the batch, in frame order
//                   (pathtrace.comp:122-133)
//
// A "frame batch" is up to 64 consecutive frames traced as one wavefront (path slot = frame x pixel slot); queue sizes live
// on the device in a per-bounce counter block (16 words per bounce, zeroed once per sample pass), so a batch is enqueued
// without any host synchronisation.  Queue appends never issue one returning atomic per wave on a shared counter (~11 ns each,
// serialised): they are aggregated per workgroup through LDS (k_generate, k_shade) or staged in LDS and flushed every ~200
// entries (persistent kernels).
//
// Random numbers are drawn in the reference's order from one PCG state per
// path that travels with the path state.  Queue order never influences a pixel's value.
#include <hip/hip_runtime.h>
#include <functional>
#include <vector>
#include <cstring>
#include "pt_bsdf.h"
#include "pt_internal.h"
#include "pt_sky.h"
#include "pt_machine.h"
#include "pt_packet.h"
#include "pt_settle.h"
#include "pt_shade.h"

namespace {

// resident waves per SIMD the trace kernels are compiled for (bounds their VGPR budget: 512 / waves)
#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 5  // 96 VGPRs; 6 (80 VGPRs) spills once the fused slab test's per-ray constants are live: 996 vs 1070 Msamples/s (r01)
#endif
#ifndef PT_TRACE_WAVES_TWO
#define PT_TRACE_WAVES_TWO 4  // two-level instantiations: object-space ray constants + instance context are live on top of the flat state (128 VGPRs)
#endif
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4  // 128 VGPRs.  The first bounce's shade launch streams ~700 B per path and is HBM-bound: a fourth wave per SIMD keeps more
                          // loads in flight (+5 % on the 96-step bench against 3 waves / 137 VGPRs, profiles/r03h_*)
#endif
constexpr int SHADE_BLOCK = 256;

// ---- wave-aggregated helpers ---------------------------------------------------------------------------
PT_DEV void count_event(unsigned long long* ctr)
{
  unsigned long long m = __ballot(1);
  if(int(threadIdx.x & 63) == __ffsll((long long)m) - 1)
    atomicAdd(ctr, (unsigned long long)__popcll(m));
}
// Appends `slot` for every active lane with one atomic per wave (ballot + mbcnt-style rank).
PT_DEV void enqueue(uint32_t* queue, uint32_t* count, uint32_t slot)
{
  unsigned long long m      = __ballot(1);
  int                lane   = threadIdx.x & 63;
  int                leader = __ffsll((long long)m) - 1;
  uint32_t           base   = 0;
  if(lane == leader)
    base = atomicAdd(count, (uint32_t)__popcll(m));
  base                   = __shfl(base, leader);
  queue[base + __popcll(m & ((1ull << lane) - 1ull))] = slot;
}

// Block-wide variant for kernels whose lanes all reach the call (no early return): one atomic per workgroup.
// `valid` lanes append `slot`.  Same-address atomics that return a value are latency-serialised per address, so a
// launch of a few 100 k waves must not issue one per wave.
PT_DEV void enqueue_block(uint32_t* queue, uint32_t* count, uint32_t slot, bool valid)
{
  __shared__ uint32_t sBase, sCount;
  if(threadIdx.x == 0)
    sCount = 0;
  __syncthreads();
  const unsigned long long m    = __ballot(valid);
  const int                lane = threadIdx.x & 63;
  uint32_t                 wbase = 0;
  if(lane == 0 && m)
    wbase = atomicAdd(&sCount, (uint32_t)__popcll(m));  // LDS atomic
  wbase = __shfl(wbase, 0);
  __syncthreads();
  if(threadIdx.x == 0 && sCount)
    sBase = atomicAdd(count, sCount);
  __syncthreads();
  if(valid)
    queue[sBase + wbase + __popcll(m & ((1ull << lane) - 1ull))] = slot;
}

// Wave-private staging of queue appends in LDS (persistent kernels): one global atomic per ~200 entries instead of
// one per service round.  `n` is wave-uniform.
#define STAGE_CAP 256
PT_DEV void stage_flush(uint32_t* stage, uint32_t& n, uint32_t* __restrict__ queue, uint32_t* count)
{
  if(n == 0)
    return;
  const int lane = threadIdx.x & 63;
  uint32_t  base = 0;
  if(lane == 0)
    base = atomicAdd(count, n);
  base = __builtin_amdgcn_readfirstlane(base);
  for(uint32_t k = lane; k < n; k += 64)
    queue[base + k] = stage[k];
  n = 0;
}
PT_DEV void stage_push(uint32_t* stage, uint32_t& n, bool valid, uint32_t slot, uint32_t* __restrict__ queue, uint32_t* count)
{
  const unsigned long long m = __ballot(valid);
  if(!m)
    return;
  if(n + 64 > STAGE_CAP)
    stage_flush(stage, n, queue, count);
  if(valid)
    stage[n + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = slot;
  n += (uint32_t)__popcll(m);
}

// ---- k_generate -----------------------------------------------------------------------------------------
// heat-map support: nanoseconds since t0 (low 32 bits of the 100 MHz wall clock)
PT_DEV float heat_ns(uint32_t t0) { return float(uint32_t(wall_clock64()) - t0) * 10.0f; }
__global__ void __launch_bounds__(1024) k_generate(DeviceScene S, RenderBuffers rb, FrameParams fp)
{
  uint32_t       slot  = blockIdx.x * blockDim.x + threadIdx.x;
  const bool     inRange = slot < fp.numSlots * fp.batch;
  const uint32_t fb    = inRange ? slot / fp.numSlots : 0u;  // frame of the batch
  int            px = 0, py = 0;
  const bool     valid = inRange && slot_pixel(fp, rb.slotTile, slot - fb * fp.numSlots, px, py);
  enqueue_block(rb.queueA, &rb.counts[CNT_IN], slot, valid);  // bounce 0 reads queueA
  if(!valid)
    return;
  if(fp.regen)
  {  // the packet kernel of bounce 0 computes the camera ray of its paths itself (k_closest_k): nothing but the queue is written here
    if(fp.st.maxDepth == 0)
      rb.ps.rad[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  generate_ray(S, rb, fp, slot, fb, px, py);
}

// ---- closest hit ----------------------------------------------------------------------------------------------
PT_DEV void wave_add(unsigned long long* ctr, uint32_t v)
{
  for(int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off);
  if((threadIdx.x & 63) == 0 && v)
    atomicAdd(ctr, (unsigned long long)v);
}

// Persistent wavefronts on the trace machine (pt_machine.h).  The loop alternates between
//   service: lanes whose ray has finished settle it (pass A -> pass B transition, RNG draws, hit record) and every
//            idle lane pulls the next ray from the queue;
//   run:     the traversal steps, executed until fewer than `minRun` lanes are still traversing (or, once the
//            queue is empty, until all are done).
// Settling rays outside the run loop keeps the hot loop to the node step and the triangle test; a finished lane
// waits for the next service round instead of dragging ~200 instructions of epilogue into every iteration.
// HEAT: the heat-map debug mode (shaders/pathtrace.comp:89,108-119 colours a pixel by the real time its invocation took): the instrumented
// instantiation stamps every ray with the wall-clock time it spent in this kernel (fetch -> settled), added to the path's cost in rayO.w
template <bool HEAT, bool TWO>
__global__ void __launch_bounds__(TRACE_BLOCK, TWO ? PT_TRACE_WAVES_TWO : PT_TRACE_WAVES) k_closest_p(DeviceScene S, RenderBuffers rb, const uint32_t* __restrict__ queueIn, int bounce, int minRun, int chunk, int cntIn, int cntChunk)
{
  uint32_t heatT0 = 0;
  __shared__ uint32_t stack[STACK_LDS * TRACE_BLOCK];
  uint32_t            spill[STACK_SPILL];
  uint32_t*           C     = rb.counts + bounce * CNT_STRIDE;
  const uint32_t      count = C[cntIn];
  if(blockIdx.x > 0 && (unsigned long long)blockIdx.x * (TRACE_BLOCK * PT_MIN_GENERATIONS) >= count)
    return;  // small queue: fewer waves with full lanes.  Spreading such a queue over MORE waves (8 .. 56 rays each, so that the SIMDs hold more
             // resident waves and a wave waits for fewer rays) measured 4-7 % slower on the 20-step run and on an 8-GPU rank's shard
             // (profiles/r04d_*): a wave-instruction costs the same with 16 lanes as with 64
  uint32_t*           lds   = stack + threadIdx.x;
  TraceLane           L;
  RaySupply           rs;
  // rays reserved per queue atomic: ~8 reservations per wave over the launch, at least `chunk`
  rs.chunk = max(uint32_t(chunk), min(2048u, (count / (gridDim.x * 8u)) & ~63u));
  uint32_t            pslot = 0, seed = 0, nRays = 0, nAlpha = 0;
  bool                alive = false;
  L.done                    = true;
  L.cur                     = 0;
#ifdef PT_HIST
  unsigned long long hIter = 0, hInner = 0, hLeaf = 0, hService = 0, hBoth = 0, hInnerIt = 0, hLeafIt = 0;
#endif
  for(;;)
  {
    // ---- service
    if(alive && L.done)
    {
      bool fallback = (L.flags & TF_SAW_FRAC) != 0;
      if(!fallback && L.pass == 0 && (L.flags & TF_SAW_ZERO) && !pass_a_settles(L.bslot, L.bt, L.zeroMaxT, L.zeroMaxT2, L.zeroMaxT3, L.cnt))
      {
        lane_begin_count<TWO>(L);  // stay alive: pass B runs in the same loop
      }
      else
      {
        if(!fallback)
        {
          uint32_t nDraw = L.cnt;  // pass A's count when it is final, else pass B's
          if(L.bslot != BVH_NONE && !((L.bw >> 29) & TRI_OPAQUE))
            ++nDraw;  // the certain non-opaque hit consumes its own (always passing) draw
          uint32_t s2 = seed;
          if(consume_rejected_draws(s2, nDraw))
          {
            store_hit(rb, pslot, L.bslot, L.bw, TWO, L.bt, L.bu, L.bv);
            if(nDraw)
              rb.ps.rayD[pslot].w = __uint_as_float(s2);
            nAlpha += nDraw;
          }
          else
            fallback = true;
        }
        if(fallback)
          enqueue(rb.queueX, &C[CNT_X_CLOSEST], pslot);
        if(HEAT)
          rb.ps.rayO[pslot].w += heat_ns(heatT0);
        alive = false;
      }
    }
    const uint32_t qi = supply_next(rs, &C[cntChunk], count, !alive);
    if(qi != 0xffffffffu)
    {
      pslot           = queueIn[qi];
      const float4 dw = rb.ps.rayD[pslot];
      seed            = __float_as_uint(dw.w);
      lane_begin(L, xyz(rb.ps.rayO[pslot]), xyz(dw), PT_INFINITY, S.numTris == 0);
      alive = true;
      ++nRays;
      if(HEAT)
        heatT0 = uint32_t(wall_clock64());
    }
    if(!__ballot(alive))
      break;
    // ---- run
    const int target = rs.more ? minRun : 1;
#ifdef PT_HIST
    ++hService;
#endif
    while(__popcll(__ballot(!L.done)) >= target)
    {
#ifdef PT_HIST
      const uint32_t ni = __popcll(__ballot(!L.done && !(L.cur & BVH_LEAF)));
#endif
      if(!L.done && !(L.cur & BVH_LEAF))
        lane_inner<false, TWO>(S, L, lds, spill, rb.counters);
#ifdef PT_HIST
      const uint32_t nl = __popcll(__ballot(!L.done && (L.cur & BVH_LEAF)));
      ++hIter; hInner += ni; hLeaf += nl; hBoth += (ni && nl) ? 1 : 0; hInnerIt += ni ? 1 : 0; hLeafIt += nl ? 1 : 0;
#endif
      if(!L.done && (L.cur & BVH_LEAF))
        lane_leaf<false, TWO>(S, L, lds, spill);
    }
  }
#ifdef PT_HIST
  if((threadIdx.x & 63) == 0)
  {
    atomicAdd(&g_hist[5][0], hIter); atomicAdd(&g_hist[5][1], hInner); atomicAdd(&g_hist[5][2], hLeaf); atomicAdd(&g_hist[5][3], hService);
    atomicAdd(&g_hist[5][4], hBoth); atomicAdd(&g_hist[5][5], hInnerIt); atomicAdd(&g_hist[5][6], hLeafIt);
  }
#endif
  wave_add(&rb.counters->closestRays, nRays);
  wave_add(&rb.counters->alphaTests, nAlpha);
}

// Packet kernel for coherent rays (bounce 0, pt_packet.h): persistent wavefronts walk the queue 64 rays (one 8x8 pixel block)
// at a time with ONE traversal per wave.  Rays it cannot settle on the spot -- packets whose lanes disagree on a direction
// sign, rays that need pass B or the exact fallback -- are staged in LDS and appended to queueR, which the refilling trace
// machine (k_closest_p) then redoes per lane.  Keeping those paths out of this kernel keeps it at ~64 VGPRs: the packet
// traversal is a serial chain of scalar loads, so it lives on resident waves, not on instruction throughput.
#ifndef PT_PACKET_WAVES
#define PT_PACKET_WAVES 8
#endif
#ifndef PT_PACKET_WAVES_TWO
#define PT_PACKET_WAVES_TWO 6  // two-level packets carry the object-space ray constants of the instance they are in on top of the world-space ones
#endif
// fp.regen (bounce 0 only): the camera rays are computed HERE from (path slot -> pixel, frame) instead of being written by k_generate and read back --
// 32 B per sample less written, 32 B less read; what later stages need is written from here: the direction and RNG state (k_shade), and the whole
// ray of a path that goes on to the refilling trace machine (queueR).
template <bool TWO>
__global__ void __launch_bounds__(TRACE_BLOCK, TWO ? PT_PACKET_WAVES_TWO : PT_PACKET_WAVES) k_closest_k(DeviceScene S, RenderBuffers rb, FrameParams fp, const uint32_t* __restrict__ queueIn, int bounce)
{
  __shared__ uint32_t wstack[PACKET_STACK];
  __shared__ uint32_t stage[STAGE_CAP];
  uint32_t            nStage = 0, nRays = 0, nAlpha = 0;
  uint32_t*           C     = rb.counts + bounce * CNT_STRIDE;
  const uint32_t      count = C[CNT_IN];
#pragma unroll 1
  for(uint32_t base = blockIdx.x * TRACE_BLOCK; base < count; base += gridDim.x * TRACE_BLOCK)
  {
    const uint32_t i     = base + threadIdx.x;
    const bool     valid = i < count;
    uint32_t       slot = 0, seed = 0;
    f3             o = f3{0.f, 0.f, 0.f}, d = f3{0.f, 0.f, 1.f};
    const bool regen = fp.regen != 0 && bounce == 0;  // (later bounces of a packetClosest >= 2 policy trace the rays k_shade wrote)
    if(valid)
    {
      slot = queueIn[i];
      if(regen)
      {
        const uint32_t fb = slot / fp.numSlots;
        int            px = 0, py = 0;
        (void)slot_pixel(fp, rb.slotTile, slot - fb * fp.numSlots, px, py);  // (valid slots only are queued)
        camera_ray(S, fp, fb, px, py, seed, o, d);
      }
      else
      {
        o               = xyz(rb.ps.rayO[slot]);
        const float4 dw = rb.ps.rayD[slot];
        d               = xyz(dw);
        seed            = __float_as_uint(dw.w);
      }
    }
    RayHit     h;
    const bool packet = TWO ? traverse_packet_two(S, valid, o, d, wstack, h, rb.counters) : traverse_packet_closest(S, valid, o, d, wstack, h, rb.counters);
    bool       redo   = valid && !packet;
    uint32_t   seedOut = seed;
    if(valid && packet)
    {
      redo = (h.flags & TF_SAW_FRAC) != 0 || ((h.flags & TF_SAW_ZERO) && !pass_a_settles(h.slot, h.t, h.zeroMaxT, h.zeroMaxT2, h.zeroMaxT3, h.count));
      if(!redo)
      {
        uint32_t nDraw = h.count;
        if(h.slot != BVH_NONE && !((h.w >> 29) & TRI_OPAQUE))
          ++nDraw;
        uint32_t s2 = seed;
        if(consume_rejected_draws(s2, nDraw))
        {
          store_hit(rb, slot, h.slot, h.w, TWO, h.t, h.u, h.v);
          if(nDraw && !regen)
            rb.ps.rayD[slot].w = __uint_as_float(s2);
          seedOut = s2;
          nAlpha += nDraw;
          ++nRays;
        }
        else
          redo = true;
      }
    }
    if(regen && valid)
    {  // what the next stages read of the ray: k_shade the direction + RNG state; the trace machine (redo) the whole ray with the untouched seed
      rb.ps.rayD[slot] = make_float4(d.x, d.y, d.z, __uint_as_float(redo ? seed : seedOut));
      if(redo)
        rb.ps.rayO[slot] = make_float4(o.x, o.y, o.z, 0.f);
    }
    stage_push(stage, nStage, redo, slot, rb.queueR, &C[CNT_REDO]);
  }
  stage_flush(stage, nStage, rb.queueR, &C[CNT_REDO]);
  wave_add(&rb.counters->closestRays, nRays);
  wave_add(&rb.counters->alphaTests, nAlpha);
}

// Exact fallback: one ray per lane, key-ordered stochastic alpha (trace contract T5).  Runs on the rays the
// machine could not settle (fractional opacity in front of the hit, or a rejected-candidate draw of exactly 0.0).
template <bool TWO>
__global__ void __launch_bounds__(TRACE_BLOCK, TWO ? PT_TRACE_WAVES_TWO : PT_TRACE_WAVES) k_closest_x(DeviceScene S, RenderBuffers rb, int bounce)
{
  __shared__ uint32_t stack[STACK_LDS * TRACE_BLOCK];
  const uint32_t      count = rb.counts[bounce * CNT_STRIDE + CNT_X_CLOSEST];
  for(uint32_t i = blockIdx.x * TRACE_BLOCK + threadIdx.x; i < count; i += gridDim.x * TRACE_BLOCK)
  {
    const uint32_t slot = rb.queueX[i];
    const float4   dw   = rb.ps.rayD[slot];
    uint32_t       nAlpha = 0;
    settle_closest_exact<TWO>(S, rb, slot, xyz(rb.ps.rayO[slot]), xyz(dw), __float_as_uint(dw.w), stack + threadIdx.x, nAlpha);  // pt_settle.h: THE key-ordered loop (T5)
    if(nAlpha)
      atomicAdd(&rb.counters->alphaTests, (unsigned long long)nAlpha);
  }
}

#ifdef PT_SHADE_VGPRS
__attribute__((amdgpu_num_vgpr(PT_SHADE_VGPRS)))
#endif
template <int MODE>
__global__ void __launch_bounds__(SHADE_BLOCK, PT_SHADE_WAVES) k_shade(DeviceScene S, RenderBuffers rb, FrameParams fp, const uint32_t* __restrict__ queueIn, uint32_t* __restrict__ queueOut, int depth, int cntNext)
{
  __shared__ uint32_t sCnt[5], sBase[2];  // shadow, next, misses, hits, nee lookups
  uint32_t*      C     = rb.counts + depth * CNT_STRIDE;
  const uint32_t count = C[CNT_IN];
  if(blockIdx.x * SHADE_BLOCK >= count)
    return;  // the grid covers every path slot of the batch; from bounce 1 on most of its workgroups lie beyond the queue
  if(threadIdx.x < 5)
    sCnt[threadIdx.x] = 0;
  __syncthreads();
  // one path per lane, no loop: a grid-stride loop around shade_path costs ~120 extra VGPRs (256 vs 136)
  const uint32_t i    = blockIdx.x * SHADE_BLOCK + threadIdx.x;
  uint32_t       slot = 0, events = 0;
  int            to   = SHADE_DONE;
  if(i < count)
  {
    slot = queueIn[i];
    if(MODE < 0 && fp.st.debugging_mode == PT_DEBUG_HEATMAP)
    {  // heat map: the path's cost travels in rayO.w (shade_path rewrites rayO for the next bounce: carry the old value over)
      const uint32_t t0   = uint32_t(wall_clock64());
      const float    cost = rb.ps.rayO[slot].w;
      to                  = shade_path<MODE>(S, rb, fp, slot, depth, events);
      rb.ps.rayO[slot].w  = cost + heat_ns(t0);
    }
    else
      to = shade_path<MODE>(S, rb, fp, slot, depth, events);
  }
  // queue appends and statistics: wave totals into LDS, one global atomic per workgroup and counter (a returning
  // atomic per wave on one address costs ~11 ns each, serialised: 2.9 ms for the 260 k waves of a bounce-0 batch)
  const int                lane = threadIdx.x & 63;
  const unsigned long long mS = __ballot(to == SHADE_TO_SHADOW), mN = __ballot(to == SHADE_TO_NEXT);
  const unsigned long long e0 = __ballot(events & EV_MISS), e1 = __ballot(events & EV_HIT), e2 = __ballot(events & EV_NEE);
  uint32_t                 wS = 0, wN = 0;
  if(lane == 0)
  {
    if(mS) wS = atomicAdd(&sCnt[0], (uint32_t)__popcll(mS));
    if(mN) wN = atomicAdd(&sCnt[1], (uint32_t)__popcll(mN));
    if(e0) atomicAdd(&sCnt[2], (uint32_t)__popcll(e0));
    if(e1) atomicAdd(&sCnt[3], (uint32_t)__popcll(e1));
    if(e2) atomicAdd(&sCnt[4], (uint32_t)__popcll(e2));
  }
  wS = __shfl(wS, 0);
  wN = __shfl(wN, 0);
  __syncthreads();
  if(threadIdx.x == 0)
  {
    if(sCnt[0]) sBase[0] = atomicAdd(&C[CNT_SHADOW], sCnt[0]);
    if(sCnt[1]) sBase[1] = atomicAdd(&C[cntNext], sCnt[1]);  // CNT_STRIDE + CNT_IN: the next bounce's input queue; CNT_NEXT: the fused stage's second queue
    if(sCnt[2]) atomicAdd(&rb.counters->misses, (unsigned long long)sCnt[2]);
    if(sCnt[3]) atomicAdd(&rb.counters->shadedHits, (unsigned long long)sCnt[3]);
    if(sCnt[4]) atomicAdd(&rb.counters->neeLookups, (unsigned long long)sCnt[4]);
  }
  __syncthreads();
  if(to == SHADE_TO_SHADOW)
    rb.queueS[sBase[0] + wS + __popcll(mS & ((1ull << lane) - 1ull))] = slot;
  else if(to == SHADE_TO_NEXT)
    queueOut[sBase[1] + wN + __popcll(mN & ((1ull << lane) - 1ull))] = slot;
}

PT_DEV void finish_bounce(const RenderBuffers& rb, uint32_t slot, bool inShadow, uint32_t seed, uint32_t* __restrict__ queueOut, uint32_t* nextCount, bool lastBounce)
{
  if(finish_bounce_core(rb, slot, inShadow, seed) && !lastBounce)
    enqueue(queueOut, nextCount, slot);
}

// Shadow rays (trace contract T6): the closest-hit walk bounded by the light distance -- the nearest certain hit, opaque or not, ends the ray;
// zero-opacity candidates in front of it consume their draws (pass A / pass B like k_closest_p)
template <bool HEAT, bool TWO>
__global__ void __launch_bounds__(TRACE_BLOCK, TWO ? PT_TRACE_WAVES_TWO : PT_TRACE_WAVES) k_shadow_p(DeviceScene S, RenderBuffers rb, const uint32_t* __restrict__ queueIn, uint32_t* __restrict__ queueOut, int bounce, int lastBounce, int minRun, int chunk, int variant,
                                                                             int cntIn, int cntChunk)
{
  uint32_t heatT0 = 0;
  __shared__ uint32_t stack[STACK_LDS * TRACE_BLOCK];
  __shared__ uint32_t stage[STAGE_CAP];
  uint32_t            nStage = 0;
  uint32_t            spill[STACK_SPILL];
  uint32_t*           C     = rb.counts + bounce * CNT_STRIDE;
  const uint32_t      count = C[cntIn];
  if(blockIdx.x > 0 && (unsigned long long)blockIdx.x * (TRACE_BLOCK * PT_MIN_GENERATIONS) >= count)
    return;  // small queue: fewer waves with full lanes.  Spreading such a queue over MORE waves (8 .. 56 rays each, so that the SIMDs hold more
             // resident waves and a wave waits for fewer rays) measured 4-7 % slower on the 20-step run and on an 8-GPU rank's shard
             // (profiles/r04d_*): a wave-instruction costs the same with 16 lanes as with 64
  uint32_t*           lds   = stack + threadIdx.x;
  TraceLane           L;
  RaySupply           rs;
  // rays reserved per queue atomic: ~8 reservations per wave over the launch, at least `chunk`
  rs.chunk = max(uint32_t(chunk), min(2048u, (count / (gridDim.x * 8u)) & ~63u));
  uint32_t            pslot = 0, seed = 0, nRays = 0, nAlpha = 0;
  bool                alive = false;
  L.done                    = true;
  L.cur                     = 0;
#ifdef PT_HIST
  unsigned long long hIter = 0, hInner = 0, hLeaf = 0, hService = 0, hBoth = 0, hInnerIt = 0, hLeafIt = 0;
#endif
  for(;;)
  {
    // ---- service (see k_closest_p)
    bool survivor = false;
    if(alive && L.done)
    {
      bool fallback = (L.flags & TF_SAW_FRAC) != 0;
      if(!fallback && L.pass == 0 && (L.flags & TF_SAW_ZERO) && !pass_a_settles(L.bslot, L.bt, L.zeroMaxT, L.zeroMaxT2, L.zeroMaxT3, L.cnt))
      {
        lane_begin_count<TWO>(L);
      }
      else
      {
        bool inShadow = false;
        if(!fallback)
        {
          uint32_t nDraw = L.cnt;  // zero-opacity candidates in front of the hit: one rejected draw each
          if(L.bslot != BVH_NONE && !((L.bw >> 29) & TRI_OPAQUE))
            ++nDraw;               // a certain non-opaque hit consumes its own (always passing) draw; an opaque one commits without
          uint32_t s2 = seed;
          if(consume_rejected_draws(s2, nDraw))
          {
            seed     = variant == PT_VARIANT_RTX ? seed : s2;  // RTX: the any-hit shader draws from a copy (traceray_rtx.glsl:54-55)
            inShadow = L.bslot != BVH_NONE;
            nAlpha += nDraw;
          }
          else
            fallback = true;
        }
        if(fallback)
          enqueue(rb.queueX2, &C[CNT_X_SHADOW], pslot);
        else
          survivor = finish_bounce_core(rb, pslot, inShadow, seed) && !lastBounce;
        if(HEAT)
          rb.ps.rayO[pslot].w += heat_ns(heatT0);
        alive = false;
      }
    }
    stage_push(stage, nStage, survivor, pslot, queueOut, &C[CNT_STRIDE + CNT_IN]);
    const uint32_t qi = supply_next(rs, &C[cntChunk], count, !alive);
    if(qi != 0xffffffffu)
    {
      pslot = queueIn[qi];
      seed  = __float_as_uint(rb.ps.rayD[pslot].w);
      lane_begin(L, xyz(rb.ps.rayO[pslot]), xyz(rb.ps.neeDir[pslot]), rb.ps.absorb[pslot].w, S.numTris == 0);
      alive = true;
      ++nRays;
      if(HEAT)
        heatT0 = uint32_t(wall_clock64());
    }
    if(!__ballot(alive))
      break;
    // ---- run
    const int target = rs.more ? minRun : 1;
#ifdef PT_HIST
    ++hService;
#endif
    while(__popcll(__ballot(!L.done)) >= target)
    {
#ifdef PT_HIST
      const uint32_t ni = __popcll(__ballot(!L.done && !(L.cur & BVH_LEAF)));
#endif
      if(!L.done && !(L.cur & BVH_LEAF))
        lane_inner<false, TWO>(S, L, lds, spill, rb.counters);
#ifdef PT_HIST
      const uint32_t nl = __popcll(__ballot(!L.done && (L.cur & BVH_LEAF)));
      ++hIter; hInner += ni; hLeaf += nl; hBoth += (ni && nl) ? 1 : 0; hInnerIt += ni ? 1 : 0; hLeafIt += nl ? 1 : 0;
#endif
      if(!L.done && (L.cur & BVH_LEAF))
      {
        lane_leaf<false, TWO>(S, L, lds, spill);
        if(S.allOpaque && L.bslot != BVH_NONE)
          L.done = true;  // all-opaque scene: any hit inside (0, tmax) occludes and nothing draws -- the nearest one need not be found
      }
    }
  }
#ifdef PT_HIST
  if((threadIdx.x & 63) == 0)
  {
    atomicAdd(&g_hist[6][0], hIter); atomicAdd(&g_hist[6][1], hInner); atomicAdd(&g_hist[6][2], hLeaf); atomicAdd(&g_hist[6][3], hService);
    atomicAdd(&g_hist[6][4], hBoth); atomicAdd(&g_hist[6][5], hInnerIt); atomicAdd(&g_hist[6][6], hLeafIt);
  }
#endif
  stage_flush(stage, nStage, queueOut, &C[CNT_STRIDE + CNT_IN]);
  wave_add(&rb.counters->shadowRays, nRays);
  wave_add(&rb.counters->alphaTests, nAlpha);
}

// Exact fallback for shadow rays (trace contract T6 with the key-ordered alpha loop).
template <bool TWO>
__global__ void __launch_bounds__(TRACE_BLOCK, TWO ? PT_TRACE_WAVES_TWO : PT_TRACE_WAVES) k_shadow_x(DeviceScene S, RenderBuffers rb, uint32_t* __restrict__ queueOut, int bounce, int lastBounce, int variant)
{
  __shared__ uint32_t stack[STACK_LDS * TRACE_BLOCK];
  uint32_t*           C     = rb.counts + bounce * CNT_STRIDE;
  const uint32_t      count = C[CNT_X_SHADOW];
  for(uint32_t i = blockIdx.x * TRACE_BLOCK + threadIdx.x; i < count; i += gridDim.x * TRACE_BLOCK)
  {
    const uint32_t slot     = rb.queueX2[i];
    uint32_t       seed     = __float_as_uint(rb.ps.rayD[slot].w), nAlpha = 0;
    const bool     inShadow = settle_shadow_exact<TWO>(S, xyz(rb.ps.rayO[slot]), xyz(rb.ps.neeDir[slot]), rb.ps.absorb[slot].w, variant, seed, stack + threadIdx.x, nAlpha, rb.counters);  // pt_settle.h (T6)
    if(nAlpha)
      atomicAdd(&rb.counters->alphaTests, (unsigned long long)nAlpha);
    finish_bounce(rb, slot, inShadow, seed, queueOut, &C[CNT_STRIDE + CNT_IN], lastBounce != 0);
  }
}

// ---- shadow ray of bounce b and closest-hit ray of bounce b + 1 in ONE launch (round 5) --------------------------------------------------------
// The reference runs AnyHit -> rand (Russian roulette) -> the next ClosestHit back to back in one invocation (shaders/pathtrace.glsl:319-338,
// traceray_rq.glsl:108-185).  The staged chain cut that into k_shadow_p -> queue -> k_closest_p: two launches of the SAME trace machine on the same
// path with a kernel boundary (every wave drained, as long as the slowest ray) and a queue round trip in between.  Here a lane that settles a
// shadow ray adds the NEE contribution, draws the roulette (finish_bounce_core) and, if the path lives, begins the path's next closest-hit ray on
// the spot -- the lane never goes idle, nothing is queued.  Paths that k_shade sent on WITHOUT a shadow ray (queueN) are pulled after the shadow
// rays.  Closest-hit rays that settle are appended to the hit queue k_shade of bounce b + 1 reads.  A bounce then costs two launch drains (trace,
// shade) instead of three.  Ray kinds differ only in their upper bound and in what the service round does with the result: the walk is the same
// (trace contract T5 / T6).  Bit-identical to the staged chain by construction: the same functions on the same per-path state in the same order.
template <bool TWO>
__global__ void __launch_bounds__(TRACE_BLOCK, TWO ? PT_TRACE_WAVES_TWO : PT_TRACE_WAVES) k_trace_p(DeviceScene S, RenderBuffers rb, const uint32_t* __restrict__ queueS, const uint32_t* __restrict__ queueN,
                                                                                                    uint32_t* __restrict__ queueHit, int bounce, int minRun, int chunk, int variant)
{
  __shared__ uint32_t stack[STACK_LDS * TRACE_BLOCK];
  __shared__ uint32_t stage[STAGE_CAP];
  uint32_t            nStage = 0;
  uint32_t            spill[STACK_SPILL];
  uint32_t*           C  = rb.counts + bounce * CNT_STRIDE;  // this bounce: shadow rays, paths without one
  uint32_t*           C1 = C + CNT_STRIDE;                   // the next bounce: its hit queue, its closest-hit fallbacks
  const uint32_t      nS = C[CNT_SHADOW], count = nS + C[CNT_NEXT];
  if(blockIdx.x > 0 && (unsigned long long)blockIdx.x * (TRACE_BLOCK * PT_MIN_GENERATIONS) >= count)
    return;
  uint32_t* lds = stack + threadIdx.x;
  TraceLane L;
  RaySupply rs;
  rs.chunk = max(uint32_t(chunk), min(2048u, (count / (gridDim.x * 8u)) & ~63u));
  uint32_t pslot = 0, seed = 0, nShadow = 0, nClosest = 0, nAlpha = 0;
  bool     alive = false, shadowRay = false;
  L.done         = true;
  L.cur          = 0;
  for(;;)
  {
    // ---- service
    bool settled = false;  // a closest-hit ray of bounce + 1 whose hit record is written: the path goes to k_shade
    if(alive && L.done)
    {
      bool fallback = (L.flags & TF_SAW_FRAC) != 0;
      if(!fallback && L.pass == 0 && (L.flags & TF_SAW_ZERO) && !pass_a_settles(L.bslot, L.bt, L.zeroMaxT, L.zeroMaxT2, L.zeroMaxT3, L.cnt))
        lane_begin_count<TWO>(L);  // stay alive: pass B runs in the same loop
      else
      {
        uint32_t s2 = seed, nDraw = L.cnt;
        if(L.bslot != BVH_NONE && !((L.bw >> 29) & TRI_OPAQUE))
          ++nDraw;  // the certain non-opaque hit consumes its own (always passing) draw
        if(!fallback && !consume_rejected_draws(s2, nDraw))
          fallback = true;
        if(!fallback)
          nAlpha += nDraw;
        if(shadowRay)
        {
          if(fallback)
          {
            enqueue(rb.queueX2, &C[CNT_X_SHADOW], pslot);
            alive = false;
          }
          else
          {
            seed = variant == PT_VARIANT_RTX ? seed : s2;  // RTX: the any-hit shader draws from a copy (traceray_rtx.glsl:54-55)
            if(finish_bounce_core(rb, pslot, L.bslot != BVH_NONE, seed))
            {  // the path lives: its next closest-hit ray (written by k_shade) starts in this lane right away
              lane_begin(L, xyz(rb.ps.rayO[pslot]), xyz(rb.ps.rayD[pslot]), PT_INFINITY, S.numTris == 0);
              shadowRay = false;
              ++nClosest;
            }
            else
              alive = false;
          }
        }
        else
        {
          if(fallback)
            enqueue(rb.queueX, &C1[CNT_X_CLOSEST], pslot);
          else
          {
            store_hit(rb, pslot, L.bslot, L.bw, TWO, L.bt, L.bu, L.bv);
            if(nDraw)
              rb.ps.rayD[pslot].w = __uint_as_float(s2);
            settled = true;
          }
          alive = false;
        }
      }
    }
    stage_push(stage, nStage, settled, pslot, queueHit, &C1[CNT_IN]);
    const uint32_t qi = supply_next(rs, &C[CNT_CHUNK_SHADOW], count, !alive);
    if(qi != 0xffffffffu)
    {
      shadowRay = qi < nS;
      if(shadowRay)
      {
        pslot = queueS[qi];
        seed  = __float_as_uint(rb.ps.rayD[pslot].w);
        lane_begin(L, xyz(rb.ps.rayO[pslot]), xyz(rb.ps.neeDir[pslot]), rb.ps.absorb[pslot].w, S.numTris == 0);
        ++nShadow;
      }
      else
      {
        pslot           = queueN[qi - nS];
        const float4 dw = rb.ps.rayD[pslot];
        seed            = __float_as_uint(dw.w);
        lane_begin(L, xyz(rb.ps.rayO[pslot]), xyz(dw), PT_INFINITY, S.numTris == 0);
        ++nClosest;
      }
      alive = true;
    }
    if(!__ballot(alive))
      break;
    // ---- run
    const int target = rs.more ? minRun : 1;
    while(__popcll(__ballot(!L.done)) >= target)
    {
      if(!L.done && !(L.cur & BVH_LEAF))
        lane_inner<false, TWO>(S, L, lds, spill, rb.counters);
      if(!L.done && (L.cur & BVH_LEAF))
      {
        lane_leaf<false, TWO>(S, L, lds, spill);
        if(shadowRay && S.allOpaque && L.bslot != BVH_NONE)
          L.done = true;  // all-opaque scene: any hit inside (0, tmax) occludes and nothing draws -- the nearest one need not be found
      }
    }
  }
  stage_flush(stage, nStage, queueHit, &C1[CNT_IN]);
  wave_add(&rb.counters->shadowRays, nShadow);
  wave_add(&rb.counters->closestRays, nClosest);
  wave_add(&rb.counters->alphaTests, nAlpha);
}

// Exact fallbacks of the fused stage: the shadow rays of bounce b that k_trace_p could not settle (queueX2) -- a survivor of the roulette goes straight on
// to the exact closest-hit loop of its next ray -- and the closest-hit rays of bounce b + 1 it could not settle (queueX).  Normally almost empty.
template <bool TWO>
__global__ void __launch_bounds__(TRACE_BLOCK, TWO ? PT_TRACE_WAVES_TWO : PT_TRACE_WAVES) k_trace_x(DeviceScene S, RenderBuffers rb, uint32_t* __restrict__ queueHit, int bounce, int variant)
{
  __shared__ uint32_t stack[STACK_LDS * TRACE_BLOCK];
  uint32_t*           C  = rb.counts + bounce * CNT_STRIDE;
  uint32_t*           C1 = C + CNT_STRIDE;
  const uint32_t      nXS = C[CNT_X_SHADOW], count = nXS + C1[CNT_X_CLOSEST];
  for(uint32_t i = blockIdx.x * TRACE_BLOCK + threadIdx.x; i < count; i += gridDim.x * TRACE_BLOCK)
  {
    const uint32_t slot = i < nXS ? rb.queueX2[i] : rb.queueX[i - nXS];
    uint32_t       seed = __float_as_uint(rb.ps.rayD[slot].w), nAlpha = 0;
    bool           alive = true;
    if(i < nXS)
    {  // the shadow ray (T6), the roulette, and -- for a survivor -- straight on to its next closest-hit ray
      const bool inShadow = settle_shadow_exact<TWO>(S, xyz(rb.ps.rayO[slot]), xyz(rb.ps.neeDir[slot]), rb.ps.absorb[slot].w, variant, seed, stack + threadIdx.x, nAlpha, rb.counters);
      alive               = finish_bounce_core(rb, slot, inShadow, seed);
      if(alive)
        atomicAdd(&rb.counters->closestRays, 1ull);  // (the shadow ray was counted when k_trace_p fetched it; the next ray starts here)
    }
    if(alive)
      settle_closest_exact<TWO>(S, rb, slot, xyz(rb.ps.rayO[slot]), xyz(rb.ps.rayD[slot]), seed, stack + threadIdx.x, nAlpha);  // (T5; stores the hit and the RNG state)
    if(nAlpha)
      atomicAdd(&rb.counters->alphaTests, (unsigned long long)nAlpha);
    if(!alive)
      continue;
    enqueue(queueHit, &C1[CNT_IN], slot);
  }
}

// ---- tail of a batch: the late bounces in ONE launch ------------------------------------------------------------------------------------
// Russian roulette from depth 0 shrinks the queues ~4x per bounce, so after two or three bounces a stage no longer fills the chip and its
// duration is that of its slowest ray (200-300 us for the trace stages, measured: profiles/r02_trace_batch.txt) -- three such floors per
// bounce, five or six bounces in a row, on every launch sequence.  Once a bounce's queue is small (host decision, pt_capi.hip: queue sizes
// of earlier batches come back asynchronously) the rest of the path runs here: persistent wavefronts, one path per lane carried through
// closest hit -> shade -> shadow ray -> Russian roulette -> next bounce, a lane whose path ended pulls the next one from the queue.  Nothing
// waits for the slowest ray of a stage any more, only for the longest remaining PATH.  Every step is the function the staged kernels call
// (traverse<>, shade_path, finish_bounce_core) on the same per-path state in the same order: results are bit-identical by construction
// (paths never interact), which the launch-policy tests assert.
template <bool TWO>
__global__ void __launch_bounds__(TRACE_BLOCK, 2) k_tail(DeviceScene S, RenderBuffers rb, FrameParams fp, const uint32_t* __restrict__ queueIn, int depth0)
{
  __shared__ uint32_t stack[STACK_LDS * TRACE_BLOCK];
  uint32_t*           C     = rb.counts + depth0 * CNT_STRIDE;
  const uint32_t      count = C[CNT_IN];
  if(blockIdx.x > 0 && (unsigned long long)blockIdx.x * TRACE_BLOCK >= count)
    return;
  uint32_t* lds = stack + threadIdx.x;
  RaySupply rs;
  rs.chunk       = 64;
  uint32_t slot  = 0, nClosest = 0, nShadow = 0, nAlpha = 0, nMiss = 0, nHit = 0, nNee = 0;
  int      depth = depth0;
  bool     alive = false;
  const int lastDepth = fp.st.maxDepth - 1;
#pragma unroll 1
  for(;;)
  {
    const uint32_t qi = supply_next(rs, &C[CNT_CHUNK_TAIL], count, !alive);
    if(qi != 0xffffffffu)
    {
      slot  = queueIn[qi];
      depth = depth0;
      alive = true;
    }
    if(!__ballot(alive))
      break;
    if(alive)
    {
      tail_closest<TWO>(S, rb, slot, lds, nAlpha);
      ++nClosest;
    }
    int      to     = SHADE_DONE;
    uint32_t events = 0;
    if(alive)
      to = shade_path<-1>(S, rb, fp, slot, depth, events);
    nMiss += (events & EV_MISS) ? 1u : 0u;
    nHit += (events & EV_HIT) ? 1u : 0u;
    nNee += (events & EV_NEE) ? 1u : 0u;
    bool survive = to == SHADE_TO_NEXT;
    if(to == SHADE_TO_SHADOW)
    {
      uint32_t   seed;
      const bool inShadow = tail_shadow<TWO>(S, rb, slot, lds, fp.variant, seed, nAlpha);
      ++nShadow;
      survive = finish_bounce_core(rb, slot, inShadow, seed) && depth != lastDepth;
    }
    alive = survive;
    ++depth;
  }
  wave_add(&rb.counters->closestRays, nClosest);
  wave_add(&rb.counters->shadowRays, nShadow);
  wave_add(&rb.counters->alphaTests, nAlpha);
  wave_add(&rb.counters->misses, nMiss);
  wave_add(&rb.counters->shadedHits, nHit);
  wave_add(&rb.counters->neeLookups, nNee);
  // the same again as "of which in the tail": per-stage byte / ray accounting of bench.py
  wave_add(&rb.counters->tailClosestRays, nClosest);
  wave_add(&rb.counters->tailShadowRays, nShadow);
  wave_add(&rb.counters->tailAlphaTests, nAlpha);
  wave_add(&rb.counters->tailMisses, nMiss);
  wave_add(&rb.counters->tailShadedHits, nHit);
}

// ---- k_accumulate ---------------------------------------------------------------------------------------------
// The last kernel of a sample pass also hands the pass's per-bounce counter block over (queue-size feedback reads the copy) and leaves it zeroed for
// the next pass on this frame slot: no runtime fill kernel per launch sequence (profiles/r04_final_kernel_stats_bench20.csv: one fillBufferAligned each).
__global__ void __launch_bounds__(256) k_accumulate(RenderBuffers rb, FrameParams fp)
{
  if(blockIdx.x == 0)
  {
    const uint32_t words = CNT_STRIDE * uint32_t((fp.st.maxDepth < PT_MAX_DEPTH ? fp.st.maxDepth : PT_MAX_DEPTH) + 2);  // (pt_render_frame rejects maxDepth > PT_MAX_DEPTH; the buffers hold PT_MAX_DEPTH + 2 blocks)
    for(uint32_t i = threadIdx.x; i < words; i += blockDim.x)
    {
      rb.countsDone[i] = rb.counts[i];
      rb.counts[i]     = 0u;
    }
  }
  uint32_t pslot = blockIdx.x * blockDim.x + threadIdx.x;
  if(pslot >= fp.numSlots)
    return;
  int px, py;
  if(!slot_pixel(fp, rb.slotTile, pslot, px, py))
    return;
  accumulate_pixel(rb, fp, pslot);
}

// ---- ray picker (src/sample_example.cpp:468-511; nvvk::RayPickerKHR shoots a flag-less ray: no culling, no any-hit) ------------------
template <bool TWO>
__global__ void __launch_bounds__(64) k_pick(DeviceScene S, float pickX, float pickY, pt_SceneCamera cam, pt_PickResult* out, Counters* counters)
{
  const f2 d         = f2{pickX * 2.0f - 1.0f, pickY * 2.0f - 1.0f};
  const f4 origin    = mat4_mul(cam.viewInverse, f4{0, 0, 0, 1});
  const f4 target    = mat4_mul(cam.projInverse, f4{d.x, d.y, 1, 1});
  const f3 tn        = unit(xyz(target));
  const f4 direction = mat4_mul(cam.viewInverse, f4{tn.x, tn.y, tn.z, 0});
  const f3 o = xyz(origin), dir = xyz(direction);
  // nearest triangle in key order (t, world index), no culling: one BVH traversal (the 64 lanes of the wave walk the same ray; a pick is a
  // rare query, what matters is that it does not scale with the triangle count: 3.8 M records per click on C5 with the old brute force)
  __shared__ uint32_t stack[STACK_LDS * TRACE_BLOCK];
  RayHit              h;
  bool                dummy;
  traverse<TM_PICK, TWO>(S, o, dir, PT_INFINITY, 0.0f, 0xffffffffu, 0u, stack + threadIdx.x, h, dummy, counters);
  const float    bt = h.t, bu = h.u, bv = h.v;
  const uint32_t bs = h.slot;
  if(threadIdx.x != 0)
    return;
  pt_PickResult r;
  r.worldRayOrigin[0] = o.x; r.worldRayOrigin[1] = o.y; r.worldRayOrigin[2] = o.z;
  r.worldRayDirection[0] = dir.x; r.worldRayDirection[1] = dir.y; r.worldRayDirection[2] = dir.z;
  r.hitT = 0.f; r.primitiveID = -1; r.instanceID = 0xffffffffu; r.instanceCustomIndex = -1;
  r.baryCoord[0] = r.baryCoord[1] = r.baryCoord[2] = 0.f;
  if(bs != BVH_NONE)
  {
    r.hitT = bt;
    if(TWO)
    {
      const uint32_t w = h.w & TRI_INDEX_MASK;
      r.instanceID     = instance_of_world_tri(S, w);
      r.primitiveID    = int(w - S.instTriBase[r.instanceID]);
    }
    else
    {
      const TriRec tr = S.tris[bs];
      r.instanceID    = __float_as_uint(tr.e1n.w);
      r.primitiveID   = int(__float_as_uint(tr.e2p.w));
    }
    r.instanceCustomIndex = S.instances[r.instanceID].primMesh;
    r.baryCoord[0] = 1.0f - bu - bv; r.baryCoord[1] = bu; r.baryCoord[2] = bv;
  }
  *out = r;
}

// ---- framebuffer plumbing ---------------------------------------------------------------------------------------
__global__ void k_untile(const float4* __restrict__ tiles, const uint32_t* __restrict__ slotTile, uint32_t numSlots, int tilesX, int width, int height, float4* __restrict__ out)
{
  uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if(slot >= numSlots)
    return;
  uint32_t gt = slotTile[slot >> 10];
  uint32_t in = slot & 1023u, blk = in >> 6, lane = in & 63u;
  int      px = int(gt % uint32_t(tilesX)) * PT_TILE + int(blk & 3u) * 8 + int(lane & 7u);
  int      py = int(gt / uint32_t(tilesX)) * PT_TILE + int(blk >> 2) * 8 + int(lane >> 3);
  if(px < width && py < height)
    out[size_t(py) * width + px] = tiles[slot];
}

// inverse of k_untile: row-major image -> this rank's accumulation tiles (checkpoint restore)
__global__ void k_retile(const float4* __restrict__ in, const uint32_t* __restrict__ slotTile, uint32_t numSlots, int tilesX, int width, int height, float4* __restrict__ tiles)
{
  uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if(slot >= numSlots)
    return;
  uint32_t gt = slotTile[slot >> 10];
  uint32_t in_ = slot & 1023u, blk = in_ >> 6, lane = in_ & 63u;
  int      px = int(gt % uint32_t(tilesX)) * PT_TILE + int(blk & 3u) * 8 + int(lane & 7u);
  int      py = int(gt / uint32_t(tilesX)) * PT_TILE + int(blk >> 2) * 8 + int(lane >> 3);
  tiles[slot] = (px < width && py < height) ? in[size_t(py) * width + px] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// gathered: [nranks][maxTilesPerRank][1024] float4; rank r's j-th tile is the j-th global tile with (tx+ty)%nranks == r
__global__ void k_scatter_tiles(const float4* __restrict__ gathered, int nranks, int maxTilesPerRank, int tilesX, int tilesY, const uint32_t* __restrict__ tileLocalIndex,
                                float4* __restrict__ full)
{
  uint32_t gslot = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t total = uint32_t(tilesX) * uint32_t(tilesY) * 1024u;
  if(gslot >= total)
    return;
  uint32_t gt = gslot >> 10;
  int      tx = int(gt % uint32_t(tilesX)), ty = int(gt / uint32_t(tilesX));
  int      r  = (tx + ty) % nranks;
  uint32_t j  = tileLocalIndex[gt];
  full[gslot] = gathered[(size_t(r) * maxTilesPerRank + j) * 1024u + (gslot & 1023u)];
}

// shaders/post.frag:98-147 (TONEMAP_UNCHARTED) + shaders/tonemapping.glsl:29-65
PT_DEV f3 uncharted2(f3 c)
{
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  return ((c * (c * A + C * B) + D * E) / (c * (c * A + B) + D * F)) - E / F;
}
PT_DEV uint32_t pcg3d_x(uint32_t* v)
{
  v[0] = v[0] * 1664525u + 1013904223u;
  v[1] = v[1] * 1664525u + 1013904223u;
  v[2] = v[2] * 1664525u + 1013904223u;
  v[0] += v[1] * v[2];
  v[1] += v[2] * v[0];
  v[2] += v[0] * v[1];
  v[0] ^= v[0] >> 16u;
  v[1] ^= v[1] >> 16u;
  v[2] ^= v[2] >> 16u;
  v[0] += v[1] * v[2];
  v[1] += v[2] * v[0];
  v[2] += v[0] * v[1];
  return v[0];
}
// ---- display pass (shaders/post.frag:98-147) on the offscreen image as RenderOutput binds it ------------------------------------------
// MipView: level 0 is the offscreen image (the accumulation image, or a viewport-sized image holding it in the top-left corner while the
// viewer de-scales); further levels are the vkCmdBlitImage(VK_FILTER_LINEAR) chain of RenderOutput::genMipmap (src/render_output.cpp:188-193).
// vkCmdBlitImage, whole level to whole level: dst texel centre scaled into src space, unnormalised linear filtering, clamp-to-edge;
// horizontal lerps first, then the vertical one (the association the parity oracle uses).
__global__ void k_blit_linear(const float4* __restrict__ src, int sw, int sh, float4* __restrict__ dst, int dw, int dh)
{
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if(x >= dw || y >= dh)
    return;
  const float su = float(sw) / float(dw), sv = float(sh) / float(dh);
  float u = (float(x) + 0.5f) * su - 0.5f, v = (float(y) + 0.5f) * sv - 0.5f;
  float fu = floorf(u), fv = floorf(v);
  float a = u - fu, b = v - fv;
  int   x0 = min(max(int(fu), 0), sw - 1), x1 = min(max(int(fu) + 1, 0), sw - 1);
  int   y0 = min(max(int(fv), 0), sh - 1), y1 = min(max(int(fv) + 1, 0), sh - 1);
  float4 t00 = src[size_t(y0) * sw + x0], t10 = src[size_t(y0) * sw + x1], t01 = src[size_t(y1) * sw + x0], t11 = src[size_t(y1) * sw + x1];
  auto   bil = [&](float p00, float p10, float p01, float p11) {
    float top = p00 * (1.0f - a) + p10 * a, bot = p01 * (1.0f - a) + p11 * a;
    return top * (1.0f - b) + bot * b;
  };
  dst[size_t(y) * dw + x] = make_float4(bil(t00.x, t10.x, t01.x, t11.x), bil(t00.y, t10.y, t01.y, t11.y), bil(t00.z, t10.z, t01.z, t11.z), bil(t00.w, t10.w, t01.w, t11.w));
}
// the accumulation image placed in the top-left corner of a viewport-sized image (texels outside: zero)
__global__ void k_pad_corner(const float4* __restrict__ src, int w, int h, float4* __restrict__ dst, int dw, int dh)
{
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if(x >= dw || y >= dh)
    return;
  dst[size_t(y) * dw + x] = (x < w && y < h) ? src[size_t(y) * w + x] : make_float4(0.f, 0.f, 0.f, 0.f);
}
// the sampler RenderOutput creates for the offscreen image: NEAREST texel, NEAREST mip, REPEAT (zeroed VkSamplerCreateInfo, render_output.cpp:98-100)
PT_DEV float4 mip_fetch(const MipView& mv, float u, float v, int lod)
{
  lod   = min(max(lod, 0), mv.n - 1);
  int w = mv.w[lod], h = mv.h[lod];
  int i = int(floorf(u * float(w))), j = int(floorf(v * float(h)));
  i %= w; if(i < 0) i += w;
  j %= h; if(j < 0) j += h;
  return mv.level[lod][size_t(j) * w + i];
}
PT_DEV float lum709(float4 c) { return dot3(f3{c.x, c.y, c.z}, f3{0.2126f, 0.7152f, 0.0722f}); }

__global__ void k_tonemap(MipView mv, pt_Tonemapper tm, uint32_t* __restrict__ out)
{
  const int width = mv.w[0], height = mv.h[0];
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if(x >= width || y >= height)
    return;
  f2     uvc = f2{(float(x) + 0.5f) / float(width), (float(y) + 0.5f) / float(height)};  // passthrough.vert at the pixel centre
  f2     uvz = uvc * tm.zoom;
  float4 p   = mip_fetch(mv, uvz.x, uvz.y, 0);  // post.frag:101
  f3     hdr = xyz(p);
  if(tm.autoExposure & 1)
  {
    float avgLum2 = lum709(mip_fetch(mv, 0.5f, 0.5f, 20));  // :105-106, lod 20 clamps to the 1x1 level
    float XYZy    = (0.3575761f * hdr.x + 0.7151522f * hdr.y) + 0.1191920f * hdr.z;
    float Y       = (tm.key / avgLum2) * XYZy;
    float Yd;
    if(tm.autoExposure & 2)
    {
      // toneLocalExposure :72-96
      float       La = 0.0f;
      const float factor = tm.key / avgLum2, epsilon = 0.05f, phi = 2.0f;
      for(int i = 0; i < 7; ++i)
      {
        float v1 = lum709(mip_fetch(mv, uvz.x, uvz.y, i)) * factor;
        float v2 = lum709(mip_fetch(mv, uvz.x, uvz.y, i + 1)) * factor;
        float sc = float(1 << i);
        if(fabsf(v1 - v2) / ((tm.key * pt_pow(2.0f, phi) / (sc * sc)) + v1) > epsilon)
        {
          La = v1;
          break;
        }
        else
          La = v2;
      }
      Yd = Y / (1.0f + La);
    }
    else
      Yd = (Y * (1.0f + Y / (tm.Ywhite * tm.Ywhite))) / (1.0f + Y);  // toneExposure :64-70
    hdr = hdr / XYZy * Yd;
  }
  f3 color = uncharted2(hdr * tm.avgLum * 2.0f);
  f3 white = splat3(1.0f) / uncharted2(splat3(11.2f));
  color    = pow3(color * white, 1.0f / 2.2f);
  if(tm.dither > 0)
  {
    uint32_t r[3] = {uint32_t(x), uint32_t(y), 0u};
    pcg3d_x(r);
    f3 noise = f3{__uint_as_float(0x3f800000u | (r[0] >> 9)) - 1.0f, __uint_as_float(0x3f800000u | (r[1] >> 9)) - 1.0f, __uint_as_float(0x3f800000u | (r[2] >> 9)) - 1.0f};
    const float q = 1.f / 255.f;
    f3 lin   = pow3(color, 2.2f);
    f3 s     = pow3(lin, 1.0f / 2.2f) / q;
    f3 c0    = f3{floorf(s.x), floorf(s.y), floorf(s.z)} * q;
    f3 c1    = c0 + q;
    f3 l0 = pow3(c0, 2.2f), l1 = pow3(c1, 2.2f);
    f3 discr = f3{lerp(l0.x, l1.x, noise.x), lerp(l0.y, l1.y, noise.y), lerp(l0.z, l1.z, noise.z)};
    color    = f3{discr.x < lin.x ? c1.x : c0.x, discr.y < lin.y ? c1.y : c0.y, discr.z < lin.z ? c1.z : c0.z};
  }
  color    = f3{clampf(lerp(0.5f, color.x, tm.contrast), 0.f, 1.f), clampf(lerp(0.5f, color.y, tm.contrast), 0.f, 1.f), clampf(lerp(0.5f, color.z, tm.contrast), 0.f, 1.f)};
  color    = pow3(color, 1.0f / tm.brightness);
  float i  = dot3(color, f3{0.299f, 0.587f, 0.114f});
  color    = lerp(splat3(i), color, tm.saturation);
  f2 uv    = f2{(uvc.x * tm.renderingRatio[0] - 0.5f) * 2.0f, (uvc.y * tm.renderingRatio[1] - 0.5f) * 2.0f};
  color *= 1.0f - (uv.x * uv.x + uv.y * uv.y) * tm.vignette;
  auto q8 = [](float v) { return uint32_t(floorf(clampf(v, 0.f, 1.f) * 255.0f + 0.5f)); };
  out[size_t(y) * width + x] = q8(color.x) | (q8(color.y) << 8) | (q8(color.z) << 16) | (q8(p.w) << 24);
}

__global__ void k_mean(const float4* __restrict__ img, size_t n, double* out3)
{
  __shared__ double sh[3][256];
  double            s[3] = {0, 0, 0};
  for(size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
  {
    float4 p = img[i];
    s[0] += p.x;
    s[1] += p.y;
    s[2] += p.z;
  }
  for(int k = 0; k < 3; ++k)
    sh[k][threadIdx.x] = s[k];
  __syncthreads();
  for(int off = 128; off > 0; off >>= 1)
  {
    if(threadIdx.x < (unsigned)off)
      for(int k = 0; k < 3; ++k)
        sh[k][threadIdx.x] += sh[k][threadIdx.x + off];
    __syncthreads();
  }
  if(threadIdx.x == 0)
    for(int k = 0; k < 3; ++k)
      atomicAdd(&out3[k], sh[k][0]);
}

}  // namespace

// ---- host-side launchers ----------------------------------------------------------------------------------------

// One launch sequence (a batch of frames, all bounces) as a list of STEPS.  A step only enqueues work on `stream`.  The caller either runs
// the steps back to back (pt_launch_frame) or interleaves the steps of several sequences that go to different streams (pt_capi.hip
// flush_pending): enqueueing the launches costs the host ~1 ms, so with one sequence submitted after the other the fourth stream would start
// 2-3 ms late -- a third of the whole run when a short run is cut into four pieces (profiles/r02b_shard_timeline_0of8.txt).
//   generate | closest(0) | shade(0) | trace(0 -> 1) | shade(1) | trace(1 -> 2) | ... | shade(t-1) | shadow(t-1) | k_tail(t ...) | accumulate
// trace(b -> b+1) = k_trace_p + k_trace_x: the shadow rays of bounce b and the closest-hit rays of bounce b + 1 in one launch (PT_TUNE fuse=0: the
// round-4 chain closest | shade | shadow per bounce).  The bounce before k_tail (and the last bounce of the path) ends with the plain shadow stage.
// The counter block is zero on entry: pt_resize clears it once, k_accumulate leaves it cleared.
// TWO: the kernels instantiated for the two-level acceleration structure
template <bool TWO>
static void plan_frame(std::vector<PtStep>& steps, hipStream_t stream, const PtTuning& tune, const DeviceScene& scene, const RenderBuffers& rb, const FrameParams& fpIn, StageTimers* tm, hipEvent_t waitBeforeAccum,
                       hipEvent_t recordAfterAccum, int tailFrom)
{
  FrameParams    fp        = fpIn;
  const uint32_t n         = fp.numSlots * fp.batch;  // path slots of the batch
  const uint32_t wavesAll  = (n + TRACE_BLOCK - 1) / TRACE_BLOCK;
  const uint32_t pw        = PT_PERSISTENT_WAVES;
  const uint32_t gridTrace = wavesAll < pw ? wavesAll : pw;
  const uint32_t gridX     = wavesAll < 512u ? wavesAll : 512u;
  const bool     heat      = fp.st.debugging_mode == PT_DEBUG_HEATMAP;  // instrumented instantiations of the staged machine kernels; no packet stage, no fused stage, no k_tail
  // measured (profiles/r05a_*, r05b_*, r05c_*): serialised, the fused launch takes exactly the time of the two launches it replaces (21.3 ms per 32-frame
  // batch either way) and k_shade reads a hit queue scrambled by two traversals instead of one (+6 %): -4 % on batches, +2.5 % on a frame the host waits for
  const bool     fuse      = !heat && (tune.fuse == 2 || (tune.fuse == 1 && fp.batch == 1));
  // camera rays computed by the packet kernel instead of written by k_generate: one sample per frame (the RNG stream of a second sample continues
  // from the stored state), a packet stage at bounce 0, no heat map (it keeps the path's cost in rayO.w), bounce 0 not already in k_tail
  const int  packetBounces = tune.packetClosestBounces;
  const bool packetStage = !TWO || tune.packetTwo;  // the two-level structure has a packet stage of its own since round 4 (pt_packet.h traverse_packet_two)
  fp.regen = (tune.regen && packetStage && !heat && fp.st.maxSamples == 1 && packetBounces >= 1 && tailFrom > 0 && fp.st.maxDepth > 0) ? 1 : 0;
  for(int s = 0; s < fp.st.maxSamples; ++s)
  {
    fp.sample = s;
    steps.push_back(PtStep{[=]() {
      pt_timers_begin(tm, stream, 0);
      k_generate<<<(n + 1023) / 1024, 1024, 0, stream>>>(scene, rb, fp);
      pt_timers_end(tm, stream, 0);
    }, false});
    uint32_t* qIn     = rb.queueA;
    uint32_t* qOut    = rb.queueB;
    bool      fusedIn = false;  // this bounce's closest-hit rays were traced by the previous bounce's fused stage (qIn is its hit queue)
    for(int depth = 0; depth < fp.st.maxDepth; ++depth)
    {
      const int last = depth == fp.st.maxDepth - 1 ? 1 : 0;
      if(depth >= tailFrom && !heat)
      {  // the remaining bounces in one launch (k_tail): the queue is small, a staged bounce would cost its latency floors
        steps.push_back(PtStep{[=]() {
          pt_timers_begin(tm, stream, 5);
          k_tail<TWO><<<wavesAll < 2048u ? wavesAll : 2048u, TRACE_BLOCK, 0, stream>>>(scene, rb, fp, qIn, depth);
          pt_timers_end(tm, stream, 5);
        }, false});
        break;
      }
      // the next bounce is a staged one too: this bounce's shadow rays and its closest-hit rays share a launch
      const bool fusedOut = fuse && !last && depth + 1 < tailFrom;
      if(!fusedIn)
        steps.push_back(PtStep{[=]() {
          pt_timers_begin(tm, stream, 1);
          if(heat)
            k_closest_p<true, TWO><<<gridTrace, TRACE_BLOCK, 0, stream>>>(scene, rb, qIn, depth, PT_REFILL_BELOW_DEFAULT, PT_SUPPLY_CHUNK, CNT_IN, CNT_CHUNK_CLOSEST);
          else if(packetStage && depth < packetBounces)
          {
            const uint32_t kwAll = PT_PACKET_WAVES_LAUNCH;
            const uint32_t kw    = TWO ? kwAll * PT_PACKET_WAVES_TWO / PT_PACKET_WAVES : kwAll;
            k_closest_k<TWO><<<wavesAll < kw ? wavesAll : kw, TRACE_BLOCK, 0, stream>>>(scene, rb, fp, qIn, depth);
            k_closest_p<false, TWO><<<gridTrace, TRACE_BLOCK, 0, stream>>>(scene, rb, rb.queueR, depth, PT_REFILL_BELOW_DEFAULT, PT_SUPPLY_CHUNK, CNT_REDO, CNT_CHUNK_REDO);
          }
          else
            k_closest_p<false, TWO><<<gridTrace, TRACE_BLOCK, 0, stream>>>(scene, rb, qIn, depth, PT_REFILL_BELOW_DEFAULT, PT_SUPPLY_CHUNK, CNT_IN, CNT_CHUNK_CLOSEST);
          k_closest_x<TWO><<<gridX, TRACE_BLOCK, 0, stream>>>(scene, rb, depth);
          pt_timers_end(tm, stream, 1);
        }, false});
      steps.push_back(PtStep{[=]() {
        pt_timers_begin(tm, stream, 2);
        // paths that go on without a shadow ray: the fused stage pulls them from qOut behind the shadow rays (CNT_NEXT); else they open the next bounce's queue
        k_shade<-1><<<(n + SHADE_BLOCK - 1) / SHADE_BLOCK, SHADE_BLOCK, 0, stream>>>(scene, rb, fp, qIn, qOut, depth, fusedOut ? CNT_NEXT : CNT_STRIDE + CNT_IN);
        pt_timers_end(tm, stream, 2);
      }, false});
      if(fusedOut)
      {  // qIn has been consumed by k_shade: it becomes the hit queue of bounce depth + 1 (no swap)
        steps.push_back(PtStep{[=]() {
          pt_timers_begin(tm, stream, 6);
          k_trace_p<TWO><<<gridTrace, TRACE_BLOCK, 0, stream>>>(scene, rb, rb.queueS, qOut, qIn, depth, PT_REFILL_BELOW_DEFAULT, PT_SUPPLY_CHUNK, fp.variant);
          k_trace_x<TWO><<<gridX, TRACE_BLOCK, 0, stream>>>(scene, rb, qIn, depth, fp.variant);
          pt_timers_end(tm, stream, 6);
        }, false});
      }
      else
      {
        steps.push_back(PtStep{[=]() {
          pt_timers_begin(tm, stream, 3);
          if(heat)
            k_shadow_p<true, TWO><<<gridTrace, TRACE_BLOCK, 0, stream>>>(scene, rb, rb.queueS, qOut, depth, last, PT_REFILL_BELOW_DEFAULT, PT_SUPPLY_CHUNK, fp.variant, CNT_SHADOW, CNT_CHUNK_SHADOW);
          else
            k_shadow_p<false, TWO><<<gridTrace, TRACE_BLOCK, 0, stream>>>(scene, rb, rb.queueS, qOut, depth, last, PT_REFILL_BELOW_DEFAULT, PT_SUPPLY_CHUNK, fp.variant, CNT_SHADOW, CNT_CHUNK_SHADOW);
          k_shadow_x<TWO><<<gridX, TRACE_BLOCK, 0, stream>>>(scene, rb, qOut, depth, last, fp.variant);
          pt_timers_end(tm, stream, 3);
        }, false});
        std::swap(qIn, qOut);
      }
      fusedIn = fusedOut;
    }
    const bool waitHere = waitBeforeAccum && s == 0, recordHere = recordAfterAccum && s == fp.st.maxSamples - 1;
    steps.push_back(PtStep{[=]() {
      // the running mean folds frames in order: wait for the previous frame's accumulate (another stream)
      if(waitHere)
        (void)hipStreamWaitEvent(stream, waitBeforeAccum, 0);
      pt_timers_begin(tm, stream, 4);
      k_accumulate<<<(fp.numSlots + 255) / 256, 256, 0, stream>>>(rb, fp);
      pt_timers_end(tm, stream, 4);
      if(recordHere)
        (void)hipEventRecord(recordAfterAccum, stream);
    }, true});
  }
}

void pt_plan_frame(std::vector<PtStep>& steps, hipStream_t stream, const PtTuning& tune, const DeviceScene& scene, const RenderBuffers& rb, const FrameParams& fp, StageTimers* tm, hipEvent_t waitBeforeAccum,
                   hipEvent_t recordAfterAccum, int tailFrom)
{
  if(scene.twoLevel)
    plan_frame<true>(steps, stream, tune, scene, rb, fp, tm, waitBeforeAccum, recordAfterAccum, tailFrom);
  else
    plan_frame<false>(steps, stream, tune, scene, rb, fp, tm, waitBeforeAccum, recordAfterAccum, tailFrom);
}

void pt_launch_frame(hipStream_t stream, const PtTuning& tune, const DeviceScene& scene, const RenderBuffers& rb, const FrameParams& fp, StageTimers* tm, hipEvent_t waitBeforeAccum, hipEvent_t recordAfterAccum,
                     int tailFrom)
{
  std::vector<PtStep> steps;
  pt_plan_frame(steps, stream, tune, scene, rb, fp, tm, waitBeforeAccum, recordAfterAccum, tailFrom);
  for(PtStep& st : steps)
    st.fn();
}

void pt_launch_pick(hipStream_t stream, const DeviceScene& scene, float px, float py, const float* viewInv, const float* projInv, pt_PickResult* dOut, Counters* counters)
{
  pt_SceneCamera cam = scene.camera;
  std::memcpy(cam.viewInverse, viewInv, sizeof(cam.viewInverse));
  std::memcpy(cam.projInverse, projInv, sizeof(cam.projInverse));
  if(scene.twoLevel)
    k_pick<true><<<1, 64, 0, stream>>>(scene, px, py, cam, dOut, counters);
  else
    k_pick<false><<<1, 64, 0, stream>>>(scene, px, py, cam, dOut, counters);
}

void pt_launch_untile(hipStream_t stream, const float4* frameTiles, const uint32_t* slotTile, uint32_t numLocalTiles, int tilesX, int width, int height, float4* outRowMajor)
{
  uint32_t n = numLocalTiles * 1024u;
  k_untile<<<(n + 255) / 256, 256, 0, stream>>>(frameTiles, slotTile, n, tilesX, width, height, outRowMajor);
}

void pt_launch_retile(hipStream_t stream, const float4* rowMajor, const uint32_t* slotTile, uint32_t numLocalTiles, int tilesX, int width, int height, float4* frameTiles)
{
  uint32_t n = numLocalTiles * 1024u;
  if(n)
    k_retile<<<(n + 255) / 256, 256, 0, stream>>>(rowMajor, slotTile, n, tilesX, width, height, frameTiles);
}

void pt_launch_scatter_tiles(hipStream_t stream, const float4* gathered, int nranks, int maxTilesPerRank, int tilesX, int tilesY, const uint32_t* tileLocalIndex, float4* fullTiles)
{
  uint32_t n = uint32_t(tilesX) * uint32_t(tilesY) * 1024u;
  k_scatter_tiles<<<(n + 255) / 256, 256, 0, stream>>>(gathered, nranks, maxTilesPerRank, tilesX, tilesY, tileLocalIndex, fullTiles);
}

void pt_launch_tonemap(hipStream_t stream, const MipView& mv, const pt_Tonemapper& tm, uint32_t* outRgba8)
{
  dim3 b(16, 16), g((mv.w[0] + 15) / 16, (mv.h[0] + 15) / 16);
  k_tonemap<<<g, b, 0, stream>>>(mv, tm, outRgba8);
}
void pt_launch_blit_linear(hipStream_t stream, const float4* src, int sw, int sh, float4* dst, int dw, int dh)
{
  dim3 b(16, 16), g((dw + 15) / 16, (dh + 15) / 16);
  k_blit_linear<<<g, b, 0, stream>>>(src, sw, sh, dst, dw, dh);
}
void pt_launch_pad_corner(hipStream_t stream, const float4* src, int w, int h, float4* dst, int dw, int dh)
{
  dim3 b(16, 16), g((dw + 15) / 16, (dh + 15) / 16);
  k_pad_corner<<<g, b, 0, stream>>>(src, w, h, dst, dw, dh);
}
void pt_launch_mean(hipStream_t stream, const float4* rowMajor, size_t n, double* out3)
{
  (void)hipMemsetAsync(out3, 0, 3 * sizeof(double), stream);
  k_mean<<<256, 256, 0, stream>>>(rowMajor, n, out3);
}

#ifdef PT_HIST
// measurement build only: copies (and optionally clears) the traversal histograms of pt_trace.h
extern "C" __attribute__((visibility("default"))) int pt_debug_hist(unsigned long long* out, int reset)
{
  if(hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hist), sizeof(g_hist)) != hipSuccess)
    return -1;
  if(reset)
  {
    static unsigned long long zero[8][40];
    if(hipMemcpyToSymbol(HIP_SYMBOL(g_hist), zero, sizeof(zero)) != hipSuccess)
      return -1;
  }
  return 0;
}
#endif

