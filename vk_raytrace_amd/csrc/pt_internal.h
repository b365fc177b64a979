// Internal (non-ABI) interfaces between the translation units of libptmi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <functional>
#include <string>
#include <vector>
#include "pt_device.h"
#define PT_REFILL_BELOW_DEFAULT 48
#ifndef PT_MIN_GENERATIONS
#define PT_MIN_GENERATIONS 1  // persistent kernels: rays per lane below which a launch uses fewer waves (1: only waves beyond the queue
                              // exit; 2-32 measured slower: kernel latency matters more than the lane utilisation of small launches)
#endif

// pt_accel.hip
// Bump arena for a build's temporaries (one device allocation for many builds); what does not fit is allocated individually and freed by release().
struct PtScratch {
  char*  base = nullptr;
  size_t cap = 0, off = 0;
  void*  owned[96];
  int    numOwned = 0;
  hipError_t get(void** p, size_t bytes)
  {
    bytes = bytes ? ((bytes + 255) & ~size_t(255)) : 256;
    if(base && off + bytes <= cap)
    {
      *p = base + off;
      off += bytes;
      return hipSuccess;
    }
    if(numOwned >= 96)
      return hipErrorOutOfMemory;
    hipError_t e = hipMalloc(p, bytes);
    if(e == hipSuccess)
      owned[numOwned++] = *p;
    return e;
  }
  void release()
  {
    for(int i = 0; i < numOwned; ++i)
      (void)hipFree(owned[i]);
    numOwned = 0;
    off      = 0;
  }
};
// Launch-policy knobs (performance only, never results): one set PER CONTEXT, parsed key by key from the PT_TUNE environment variable when the context is
// created ("tail=0,batch=4,build=sah"; unknown keys are reported on stderr once).  Defaults chosen from measurements.  Knobs whose sweeps said "the default is
// best" for two rounds became constants in round 6 (PT_PERSISTENT_WAVES, PT_REFILL_BELOW_DEFAULT, PT_SUPPLY_CHUNK, PT_PACKET_WAVES_LAUNCH, PT_ROTATE_PASSES,
// PT_PLOC_RADIUS; the pieces of a cut batch are always interleaved, a full batch is never split): profiles/README.md has their sweeps.
struct PtTuning {
  int packetClosestBounces = 1;   // bounces whose closest-hit stage walks one traversal per wavefront (pt_packet.h)
  int framesInFlight       = 4;    // independent frame batches overlapped on separate streams (accumulate stays ordered)
  int stateGB              = 0;    // cap of the in-flight path state in GB (0: 85 % of the free device memory); the batch shrinks to fit
  int stateMB              = 0;    // the same cap in MB (tests of the shrink path: a budget smaller than one default batch)
  int sahBuild             = 3;    // 3: device binned SAH (default; pt_sahdev.h), 2: device PLOC, 1: host SAH topology (the cross-check of 3), 0: device LBVH (Karras radix tree)
  int tailBelow            = 65536;  // a launch sequence hands the remaining bounces to k_tail (one launch, paths carried to their end) from the first bounce whose
                                   // queue is expected to hold at most this many paths (0: never)
  int warm                 = 1;    // pt_resize with scene, camera and environment in place: write every frame slot's path state once and run one throw-away launch
                                   // sequence per slot (Renderer::create is where the reference builds its pipelines; 0: the first frames pay instead)
  int packetTwo            = 1;    // two-level structure: bounce 0 walks one traversal per wavefront through TLAS and BLASes (pt_packet.h traverse_packet_two); 0: per lane
  int bandTiles            = 64;   // a band of a single frame holds at least this many 32x32 tiles (65 k pixels); the tests lower it to reach the band path on small images
  int bands                = 3;    // a single frame launched on an idle GPU is cut into up to this many bands of its tiles, one launch sequence each (1 = off); 3: +7 %, 6: -7 % (profiles/r04z_*)
  int displaySlots         = 2;    // extra frame slots holding ONE frame each, used only by single-frame launches (the display loop); 0 = none
  int fuse                 = 1;    // shadow rays of bounce b and closest-hit rays of bounce b + 1 share one persistent launch (k_trace_p): 1 = launch sequences of ONE frame
                                   // (the display loop), 2 = always, 0 = never (the round-4 chain)
  int regen                = 1;    // bounce 0: the packet kernel computes the camera rays itself (k_generate only builds the queue); 0: k_generate writes them
  int texGroups            = 1;    // the textures a material samples with one (u, v) are also stored interleaved when they share size and sampler (pt_device.h TexRec::tiled)
  int texTile              = 1;    // RGBA8 images whose size allows it are stored block-linear (8 x 4-texel tiles = one 128-byte line; pt_device.h tex_index)
  int blasWorkers          = 8;    // two-level build: host threads (own stream + arena each) that build the BLASes concurrently
  int shadeTris            = 1;    // flat-format structures: per-slot shading line for k_shade: the triangle's vertex attributes + (instance, primitive), 128 B per triangle (0: none)
  int cnodes               = 1;    // flat-format structures: 80-byte compact nodes for the persistent trace kernels (see pt_device.h CompactNode)
  int mergeSingles         = 1;    // two-level structure: prim-meshes instantiated once share one world-space bottom-level structure (0: a BLAS each)
  int accelTwoLevel        = 0;    // 1: the context starts with the two-level acceleration structure (PT_TUNE accel=two; pt_set_accel_mode overrides)
  int batch                = 64;   // upper bound of the frames traced as one wavefront; the per-context value also keeps a batch below 2^26 paths (32 frames at 1080p)
};
// Parses a PT_TUNE string key by key ("key=value" tokens separated by commas) into `t`; every token whose key is not a knob is appended to `unknown`
// (comma separated).  Exact key matches only: "waves=" no longer matches inside "packetWaves=".
void pt_parse_tuning(const char* tune, PtTuning& t, std::string& unknown);
#define PT_SUPPLY_CHUNK 64            // rays a persistent wave reserves per queue atomic (at least)
#define PT_PACKET_WAVES_LAUNCH 8192u  // persistent waves of the packet kernel (8 per SIMD)
#define PT_ROTATE_PASSES 2            // device builders: bottom-up tree-rotation passes after the topology is built
#define PT_PLOC_RADIUS 16             // PLOC: clusters examined on either side of a cluster per round

// A FOREST (round 6): several hierarchies in ONE level-synchronous device-SAH build -- root r owns the primitives [first[r], first[r] + count[r]) (count >= 2;
// the instances are numbered like the roots and their triBase is first[r]), its binary nodes take the ids first[r] .. first[r] + count[r] - 2, its wide nodes are
// allocated from wideBase[r] on (global ids: dWideOut is then the base of the shared array) and its leaf references are offset by leafOffset.  All bottom-level
// structures of the two-level mode are built this way in one pass instead of one build (and ~20 host round trips) per prim-mesh.
struct PtForest {
  uint32_t        numRoots;
  const uint32_t* first;      // host arrays, numRoots entries each
  const uint32_t* count;
  const uint32_t* wideBase;
  uint32_t        leafOffset;
  uint32_t*       numWide;    // out: wide nodes of every root
};
int pt_accel_build(hipStream_t stream, const PtTuning& tune, const InstanceRec* dInst, uint32_t numInst, const float4* dVertices, const uint32_t* dIndices, uint32_t numTris,
                   TriRec* dTrisOut, AlphaRec* dAlphaOut, BvhNode* dNodesOut, WideNode* dWideOut, uint32_t* numWideOut, char* err, size_t errLen,
                   const TriRec* dProxies = nullptr, PtScratch* scratch = nullptr, const PtForest* forest = nullptr);
// Two-level structure (reference: src/accelstruct.cpp:110-162).  One BLAS per prim-mesh in object space ...
struct PtBlasDesc {
  uint32_t primMesh, vertexOffset, firstIndex, triCount, flags;  // flags: TRI_OPAQUE / TRI_NOCULL of the mesh's material (no TRI_FLIP: that is per instance)
  int32_t  materialIndex;
  uint32_t slotBase, nodeBase;  // where its leaf records / wide nodes start in the shared arrays (nodeBase + max(1, triCount - 1) nodes reserved)
  uint32_t numWide;             // out: wide nodes used
};
int pt_blas_build(hipStream_t stream, const PtTuning& tune, PtBlasDesc* blas, uint32_t numBlas, const float4* dVertices, const uint32_t* dIndices, TriRec* dTris, AlphaRec* dAlpha, WideNode* dWide,
                  char* err, size_t errLen);
// ... and one TLAS over the world boxes of the `numActive` non-empty instances listed in dActive (exact bounds of the T1 world triangles).
// dInstNodeBase[inst]: root node of the instance's BLAS; dInstPad[2 * inst + {0,1}]: TlasLeaf::padC0 / padC1.  rootOut: the binary root (world bounds).
int pt_tlas_build(hipStream_t stream, const PtTuning& tune, const InstanceRec* dInst, const uint32_t* dActive, uint32_t numActive, const uint32_t* dInstNodeBase, const float* dInstPad,
                  const float4* dVertices, const uint32_t* dIndices, WideNode* dTlasOut, TlasLeaf* dLeavesOut, BvhNode* rootOut, uint32_t* numWideOut, char* err, size_t errLen,
                  const float* mergedBox = nullptr, uint32_t mergedNodeBase = 0);
// ... with mergedBox (lo xyz, hi xyz) one more TLAS primitive: the merged world-space structure of the prim-meshes instantiated once
// (pt_trace.h PT_INST_MERGED), built by pt_merged_build into dTris / dAlpha / dWide at slotBase / nodeBase like a BLAS.
// WideNode -> CompactNode for the first n nodes; -1 when a node cannot be represented (the caller keeps the WideNode walk)
void pt_launch_shade_tris(hipStream_t stream, uint32_t n, const TriRec* tris, const InstanceRec* inst, const float4* vertices, const uint32_t* indices, float4* out);
int pt_compact_nodes(hipStream_t stream, uint32_t n, const WideNode* in, CompactNode* out, float* reachOut = nullptr);  // reachOut: max |p| + 2047 step over the nodes
// ... over numRanges node ranges given as (base, count) pairs
int pt_compact_node_ranges(hipStream_t stream, const uint32_t* hBaseCount, uint32_t numRanges, const WideNode* in, CompactNode* out);
int pt_merged_build(hipStream_t stream, const PtTuning& tune, const InstanceRec* hInst, const uint32_t* hIds, const uint32_t* hWorldBase, uint32_t numInst, uint32_t numTris, const float4* dVertices,
                    const uint32_t* dIndices, TriRec* dTris, AlphaRec* dAlpha, WideNode* dWide, uint32_t slotBase, uint32_t nodeBase, uint32_t* numWideOut, float* boxOut6, char* err,
                    size_t errLen);

// pt_render.hip -- one frame of the wavefront pipeline, enqueued on `stream`
// Per-bounce counter block (CNT_STRIDE words per bounce; zero when a sample pass starts: pt_resize clears it, k_accumulate leaves it cleared):
#define CNT_STRIDE 16
#define CNT_IN 0             // size of the bounce's input queue (bounce b+1's lives at +CNT_STRIDE)
#define CNT_SHADOW 1         // size of queueS (paths with a shadow ray)
#define CNT_X_CLOSEST 2      // rays handed to the exact closest-hit fallback
#define CNT_X_SHADOW 3       // rays handed to the exact shadow fallback
#define CNT_CHUNK_CLOSEST 4  // ray-supply chunk counter of k_closest_p
#define CNT_CHUNK_SHADOW 5   // ray-supply chunk counter of k_shadow_p
#define CNT_REDO 6           // size of queueR
#define CNT_CHUNK_REDO 7     // ray-supply chunk counter of k_closest_p on queueR
#define CNT_NEXT 8                // fused stage (k_trace_p): size of the queue of paths k_shade sent on without a shadow ray
#define CNT_CHUNK_TAIL 10         // path-supply chunk counter of k_tail (the bounce it starts at)
#define PT_MAX_DEPTH 256
#define PT_MAX_INFLIGHT 8
#ifndef PT_DISPLAY_RING
#define PT_DISPLAY_RING 8  // images pt_tonemap_begin may have in flight before pt_tonemap_end collects the oldest
#endif
#ifndef PT_PERSISTENT_WAVES
#define PT_PERSISTENT_WAVES (256u * 20u)  // persistent trace kernels: waves per launch = what the chip holds at 5 waves / SIMD (profiles/r03h_tune_96.txt, r03i_*)
#endif

// POL: cache policy of the path-state accesses (pt_device.h)
template <int POL>
struct RenderBuffersT {
  PathStateT<POL> ps;
  uint32_t* queueA;    // path-slot queues of the bounces (ping-pong)
  uint32_t* queueB;
  uint32_t* queueS;    // paths with a shadow ray
  uint32_t* queueX;    // exact-fallback queues (normally empty)
  uint32_t* queueX2;
  uint32_t* queueR;    // rays the packet kernel could not settle (redone per lane on the trace machine)
  uint32_t* counts;    // (PT_MAX_DEPTH + 2) x CNT_STRIDE device counters
  uint32_t* countsDone;  // the counter block of the latest finished sample pass (k_accumulate copies it here and clears `counts`)
  float4*   frame;     // accumulation tiles, slot order
  uint32_t* slotTile;  // local tile -> global tile id
  Counters* counters;
};
typedef RenderBuffersT<PT_STATE_POLICY> RenderBuffers;
void pt_sah_topology(uint32_t n, const struct TriRec* tris, uint32_t* vals, uint32_t* childL, uint32_t* childR, uint32_t* parI, uint32_t* parL);  // pt_sah.hip
struct StageTimers;  // pt_capi.hip
// waitBeforeAccum (may be null): accumDone event of the previous frame; recordAfterAccum: this frame's
// A launch sequence as steps that each enqueue one stage on the sequence's stream (pt_render.hip plan_frame).  `accum`: the step that folds the
// batch into the running mean -- it waits on the previous sequence's event, so sequences must issue their accum steps in order.
struct PtStep {
  std::function<void()> fn;
  bool                  accum;
};
void pt_plan_frame(std::vector<PtStep>& steps, hipStream_t stream, const PtTuning& tune, const DeviceScene& scene, const RenderBuffers& rb, const FrameParams& fp, StageTimers* timers, hipEvent_t waitBeforeAccum,
                   hipEvent_t recordAfterAccum, int tailFrom);
// tailFrom: first bounce handed to k_tail (>= maxDepth: none)
void pt_launch_frame(hipStream_t stream, const PtTuning& tune, const DeviceScene& scene, const RenderBuffers& rb, const FrameParams& fp, StageTimers* timers, hipEvent_t waitBeforeAccum,
                     hipEvent_t recordAfterAccum, int tailFrom);
void pt_launch_retile(hipStream_t stream, const float4* rowMajor, const uint32_t* slotTile, uint32_t numLocalTiles, int tilesX, int width, int height, float4* frameTiles);
void pt_launch_pick(hipStream_t stream, const DeviceScene& scene, float px, float py, const float* viewInv, const float* projInv, pt_PickResult* dOut, Counters* counters);
void pt_launch_untile(hipStream_t stream, const float4* frameTiles, const uint32_t* slotTile, uint32_t numLocalTiles, int tilesX, int width, int height, float4* outRowMajor);
void pt_launch_scatter_tiles(hipStream_t stream, const float4* gathered, int nranks, int maxTilesPerRank, int tilesX, int tilesY, const uint32_t* tileLocalIndex, float4* fullTiles);
// the offscreen image with its mip chain as the display pass samples it (level 0 = the image; src/render_output.cpp:188-193)
struct MipView {
  const float4* level[20];
  int           w[20], h[20];
  int           n;
};
void pt_launch_tonemap(hipStream_t stream, const MipView& mv, const pt_Tonemapper& tm, uint32_t* outRgba8);
void pt_launch_blit_linear(hipStream_t stream, const float4* src, int sw, int sh, float4* dst, int dw, int dh);
void pt_launch_pad_corner(hipStream_t stream, const float4* src, int w, int h, float4* dst, int dw, int dh);
void pt_launch_mean(hipStream_t stream, const float4* rowMajor, size_t n, double* out3);

struct StageTimers {
  bool       enabled = false;
  hipEvent_t ev[2] = {nullptr, nullptr};
  double     ms[7] = {0, 0, 0, 0, 0, 0, 0};  // generate, closest, shade, shadow, accumulate, tail (k_tail: the late bounces of a launch sequence), fused trace (k_trace_p)
  uint64_t   launchesClosest = 0, launchesTail = 0, launchesFused = 0;
  hipStream_t stream = nullptr;
  // pending (start,stop) pairs are resolved lazily to keep the stream asynchronous
  struct Pending { hipEvent_t a, b; int stage; };
  Pending*   pend = nullptr;
  size_t     npend = 0, cap = 0;
};
void pt_timers_begin(StageTimers* t, hipStream_t s, int stage);
void pt_timers_end(StageTimers* t, hipStream_t s, int stage);
void pt_timers_collect(StageTimers* t);
