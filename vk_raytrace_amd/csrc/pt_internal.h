// Internal (non-ABI) interfaces between the translation units of libptmi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "pt_device.h"

// pt_accel.hip
int pt_accel_build(hipStream_t stream, const InstanceRec* dInst, uint32_t numInst, const float4* dVertices, const uint32_t* dIndices, uint32_t numTris,
                   TriRec* dTrisOut, BvhNode* dNodesOut, char* err, size_t errLen);

// pt_render.hip -- one frame of the wavefront pipeline, enqueued on `stream`
struct RenderBuffers {
  PathState ps;
  uint32_t* queueA;    // path-slot queues (ping-pong) + one for the shadow stage
  uint32_t* queueB;
  uint32_t* queueS;
  uint32_t* counts;    // device counters: [0] queueA size, [1] queueB size, [2] queueS size
  float4*   frame;     // accumulation tiles, slot order
  uint32_t* slotTile;  // local tile -> global tile id
  Counters* counters;
};
struct StageTimers;  // pt_capi.hip
void pt_launch_frame(hipStream_t stream, const DeviceScene& scene, const RenderBuffers& rb, const FrameParams& fp, StageTimers* timers);
void pt_launch_untile(hipStream_t stream, const float4* frameTiles, const uint32_t* slotTile, uint32_t numLocalTiles, int tilesX, int width, int height, float4* outRowMajor);
void pt_launch_scatter_tiles(hipStream_t stream, const float4* gathered, int nranks, int maxTilesPerRank, int tilesX, int tilesY, const uint32_t* tileLocalIndex, float4* fullTiles);
void pt_launch_tonemap(hipStream_t stream, const float4* rowMajor, int width, int height, const pt_Tonemapper& tm, const float avg[3], uint32_t* outRgba8);
void pt_launch_mean(hipStream_t stream, const float4* rowMajor, size_t n, double* out3);

struct StageTimers {
  bool       enabled = false;
  hipEvent_t ev[2] = {nullptr, nullptr};
  double     ms[5] = {0, 0, 0, 0, 0};  // generate, closest, shade, shadow, accumulate
  uint64_t   launchesClosest = 0;
  hipStream_t stream = nullptr;
  // pending (start,stop) pairs are resolved lazily to keep the stream asynchronous
  struct Pending { hipEvent_t a, b; int stage; };
  Pending*   pend = nullptr;
  size_t     npend = 0, cap = 0;
};
void pt_timers_begin(StageTimers* t, hipStream_t s, int stage);
void pt_timers_end(StageTimers* t, hipStream_t s, int stage);
void pt_timers_collect(StageTimers* t);
