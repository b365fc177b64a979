// C ABI of libptmi.so (include/pt_api.h): context, device memory, call sequencing.
// Mirrors the division of labour of the reference's host classes -- Scene (src/scene.cpp), AccelStructure
// (src/accelstruct.cpp), HdrSampling (src/hdr_sampling.cpp), RenderOutput (src/render_output.cpp) and the
// Renderer implementations (src/rayquery.cpp, src/rtx_pipeline.cpp) -- behind one opaque pt_context.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <string>
#include <vector>
#include "../../include/pt_api.h"
#include "pt_internal.h"

namespace {
std::string g_createError;

struct DevBuf {
  void*  p     = nullptr;
  size_t bytes = 0;
};
}  // namespace

struct pt_context {
  PtTuning    tune;  // launch-policy knobs of THIS context (PT_TUNE at pt_create)
  int         device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // scene (host copies kept only for what build_accel needs)
  DevBuf   dMatLines;  // one 128-byte line per material (DeviceScene::matLines)
  DevBuf   dVertices, dIndices, dInstances, dMaterials, dLights, dTexRecs, dTexels, dBvh, dWide, dTris, dAlphaRecs, dAlphaMats, dAlphaMaps, dEnv, dEnvAccel;
  DevBuf   dShadeTris;
  bool     haveShadeTris = false;
  DevBuf   dInstBlock;  // DeviceScene::instBlock
  DevBuf   dCTlas;   // DeviceScene::ctlas
  std::vector<uint32_t> hBlasRanges;  // two-level mode: (node base, wide nodes) of every object-space BLAS
  uint32_t nodeCapacity = 0;          // nodes dWide was sized for (two-level mode: the BLASes sit at their node bases)
  DevBuf   dCNodes;  // DeviceScene::cnodes (flat-format structures, PT_TUNE cnodes=1)
  bool     haveCNodes = false;
  uint32_t numTris = 0, numInstances = 0, numBvhNodes = 0, numWideNodes = 0, numLights = 0;
  // two-level acceleration structure (pt_set_accel_mode): dWide / dTris / dAlphaRecs hold the concatenated BLASes, dTlas the instance hierarchy
  int      accelMode = PT_ACCEL_FLAT;
  DevBuf   dTlas, dTlasLeaves, dInstTriBase, dActive, dInstNodeBase, dInstPad;
  uint32_t numBlas = 0, numTlasNodes = 0, numActive = 0;
  // two-level mode: the instances whose prim-mesh is instantiated exactly once live in one world-space structure (PT_INST_MERGED) at slot 0 /
  // node 0 of the BLAS arrays; mergedOnly: nothing else exists, the structure IS the flat one and the flat kernels run on it
  std::vector<uint32_t> hMerged;
  uint32_t mergedTris = 0, mergedWide = 0;
  float    mergedBox[6] = {0, 0, 0, 0, 0, 0};
  bool     mergedOnly = false;
  std::vector<uint32_t> hInstNodeBase;  // per instance: root node of its BLAS (two-level mode, after the BLAS build)
  std::vector<float>    hPrimBound;     // per prim-mesh: max |coordinate| of its vertices (object space); bounds the rounding of the ray transform
  double   msBuildTlas = 0;
  bool     renderedSinceCheck = false;
  bool     anyHit = true;               // RtxPipeline::useAnyHit (src/rtx_pipeline.cpp:269-276); false: every triangle is opaque
  std::vector<InstanceRec> hInstances;  // as built by pt_set_scene (flags without the useAnyHit override)  // frames were launched since the traversal-stack overflow counter was last looked at
  bool     haveScene = false, haveAccel = false, haveEnv = false, haveCamera = false;
  bool     warmPending = true;  // the next pt_resize warms the frame slots (once per acceleration structure: not on the resizes of an interactive session)
  DeviceScene scene{};

  // output / path state
  int      width = 0, height = 0, tilesX = 0, tilesY = 0;
  int      rank = 0, nranks = 1;
  uint32_t numLocalTiles = 0, maxTilesPerRank = 0, numSlots = 0;
  uint64_t localPixels = 0;
  // Frames in flight: each has its own path state, queues, counter block and stream, so that the long tail of one
  // frame's stage (a launch lasts as long as its slowest ray) is filled with the work of other frames.  Only the
  // running-mean accumulate is ordered across frames (events).
  struct FrameSlot {
    DevBuf        dState[9], dQueueA, dQueueB, dQueueS, dQueueX, dQueueX2, dQueueR, dCounts, dCountsDone;
    RenderBuffers rb{};
    hipStream_t   stream    = nullptr;
    hipEvent_t    accumDone = nullptr;
    bool          launched  = false;  // a launch sequence was enqueued on this slot since the last synchronisation
    // queue-size feedback: the per-bounce counters of the slot's latest launch sequence come back asynchronously (pinned memory)
    uint32_t*     hCounts     = nullptr;
    hipEvent_t    countsDone  = nullptr;
    uint64_t      countsSeq   = 0;   // sequence number of the launch the copy belongs to (0: none)
    uint32_t      countsPaths = 0;   // paths of that launch (frames of the batch x local pixels)
    int           countsDepths = 0;  // bounces it ran staged (the counters of later bounces are not produced: k_tail took over)
  };
  FrameSlot slots[PT_MAX_INFLIGHT];
  int       inflight     = 1;  // frame slots in use (<= inflightMax: pt_resize drops slots when the device memory is short)
  int       inflightMax  = 1;  // frame slots created (streams / events exist for these)
  // Display slots: slots[inflightMax .. inflightMax + displaySlots) hold the path state of ONE frame each and join the ring only for launches of a
  // single frame -- the display loop (render, tonemap, present per frame), where six short sequences in flight beat four by 12 %, while batches
  // are fastest on four full slots (profiles/r04y_display_slots.txt)
  int       displaySlots    = 0;  // in use after pt_resize
  int       displaySlotsMax = 0;  // created
  uint64_t  displayCounter  = 0;  // ring position of the single-frame launches
  // frames handed to pt_render_frame but not launched yet: consecutive frames with identical state are traced as one
  // batch (flushed when full and by every call that reads results or changes inputs)
  pt_RtxState pendState{};
  int         pendCount = 0;
  int         batchMax  = 1;
  int         variant   = PT_VARIANT_RAYQUERY;
  uint64_t  frameCounter = 0;
  // fraction of a launch's paths still alive at the start of bounce d, from the most recent finished launch (queue-size feedback; decides
  // where k_tail takes over -- performance only)
  double    qRatio[PT_MAX_DEPTH + 1];
  int       qRatioDepths = 0;   // entries of qRatio that were observed (0: nothing observed yet)
  uint64_t  qRatioSeq    = 0;   // launch they come from
  uint64_t  launchSeq    = 0;
  hipEvent_t lastAccum   = nullptr;  // accumDone of the most recent frame (nullptr: none pending)
  DevBuf   dFrame, dSlotTile, dCounters;
  DevBuf   dPick;
  DevBuf   dRowMajor, dRgba8, dMean, dMips, dGather, dFullTiles, dFullSlotTile, dTileLocalIndex;
  bool     haveFull = false;
  // pipelined display (pt_tonemap_begin / pt_tonemap_end): a ring of pinned host images, each with the event that says its copy has landed
  // and the event after which the accumulation image may be written again (the untile pass has read it)
  struct DisplaySlot {
    uint8_t*   host  = nullptr;
    size_t     bytes = 0, used = 0;
    hipEvent_t done = nullptr, read = nullptr;
  };
  DisplaySlot display[PT_DISPLAY_RING];
  uint64_t    displayHead = 0, displayTail = 0;  // oldest image not collected yet / next one to fill
  bool     gatherEnqueued = false;  // pt_gather_shards ran on this context as the root and pt_gather_finish has not consumed it yet
  StageTimers timers;
  pt_Stats    stats{};
  double      msBuild = 0;

  int fail(int code, const char* fmt, ...)
  {
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
    if(code == PT_ERR_HIP)
      countsDirty = true;  // a launch sequence may have died before its k_accumulate cleared the per-bounce counters: flush_pending repairs them first
    return code;
  }
  bool countsDirty = false;
};

#define CTX_CHECK(ctx)       \
  if(!(ctx))                 \
    return PT_ERR_INVALID;
#define HIP_TRY(ctx, call)                                                                                   \
  do                                                                                                         \
  {                                                                                                          \
    hipError_t e_ = (call);                                                                                  \
    if(e_ != hipSuccess)                                                                                     \
      return (ctx)->fail(e_ == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while(0)

__attribute__((visibility("hidden"))) int flush_pending(pt_context* c);  // defined next to pt_render_frame (internal: not part of the ABI)
namespace {

int dev_alloc(pt_context* c, DevBuf& b, size_t bytes)
{
  if(b.p && b.bytes >= bytes && b.bytes <= bytes * 2 + 4096)
    return PT_OK;
  if(b.p)
    (void)hipFree(b.p);
  b.p     = nullptr;
  b.bytes = 0;
  if(bytes == 0)
    bytes = 16;
  HIP_TRY(c, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return PT_OK;
}
void dev_free(DevBuf& b);
// like dev_alloc, but a failed allocation is an answer (false), not an error
bool dev_alloc_quiet(DevBuf& b, size_t bytes)
{
  if(b.p && b.bytes >= bytes && b.bytes <= bytes * 2 + 4096)
    return true;
  dev_free(b);
  if(bytes == 0)
    bytes = 16;
  if(hipMalloc(&b.p, bytes) != hipSuccess)
  {
    b.p = nullptr;
    return false;
  }
  b.bytes = bytes;
  return true;
}
void dev_free(DevBuf& b)
{
  if(b.p)
    (void)hipFree(b.p);
  b.p     = nullptr;
  b.bytes = 0;
}
int upload(pt_context* c, DevBuf& b, const void* src, size_t bytes)
{
  int rc = dev_alloc(c, b, bytes);
  if(rc != PT_OK)
    return rc;
  if(bytes)
    HIP_TRY(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  return PT_OK;
}

// inverse of an affine column-major 4x4 (last row forced to 0 0 0 1), computed in double and rounded once
bool affine_inverse(const float* m, double inv[12], double& det3)
{
  const double a = m[0], b = m[4], c = m[8], d = m[1], e = m[5], f = m[9], g = m[2], h = m[6], i = m[10];
  const double tx = m[12], ty = m[13], tz = m[14];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  det3 = a * A + b * B + c * C;
  if(det3 == 0.0)
    return false;
  const double r = 1.0 / det3;
  // rows of the inverse 3x3
  const double i00 = A * r, i01 = -(b * i - c * h) * r, i02 = (b * f - c * e) * r;
  const double i10 = B * r, i11 = (a * i - c * g) * r, i12 = -(a * f - c * d) * r;
  const double i20 = C * r, i21 = -(a * h - b * g) * r, i22 = (a * e - b * d) * r;
  // column-major 3x4: columns 0..2 then translation
  inv[0] = i00; inv[1] = i10; inv[2] = i20;
  inv[3] = i01; inv[4] = i11; inv[5] = i21;
  inv[6] = i02; inv[7] = i12; inv[8] = i22;
  inv[9]  = -(i00 * tx + i01 * ty + i02 * tz);
  inv[10] = -(i10 * tx + i11 * ty + i12 * tz);
  inv[11] = -(i20 * tx + i21 * ty + i22 * tz);
  return true;
}

static inline int slot_total(const pt_context* c) { return c->inflight + c->displaySlots; }
static inline pt_context::FrameSlot& slot_at(pt_context* c, int k) { return c->slots[k < c->inflight ? k : c->inflightMax + (k - c->inflight)]; }

hipError_t sync_all(pt_context* c)
{
  if(flush_pending(c) != PT_OK)
    return hipErrorUnknown;
  for(int i = 0; i < PT_MAX_INFLIGHT; ++i)
    if(c->slots[i].stream)
    {
      hipError_t e = hipStreamSynchronize(c->slots[i].stream);
      if(e != hipSuccess)
        return e;
    }
  c->lastAccum = nullptr;
  for(int i = 0; i < PT_MAX_INFLIGHT; ++i)
    c->slots[i].launched = false;
  return hipStreamSynchronize(c->stream);
}
// After a synchronisation: a traversal that ran out of stack (STACK_LDS + STACK_SPILL entries) dropped a subtree, so the image is wrong --
// every call that hands results to the caller reports it instead of returning PT_OK with missing geometry.
int check_traversal(pt_context* c)
{
  if(!c->renderedSinceCheck)
    return PT_OK;
  unsigned int n = 0;
  HIP_TRY(c, hipMemcpy(&n, (const char*)c->dCounters.p + offsetof(Counters, stackOverflow), sizeof(n), hipMemcpyDeviceToHost));
  c->renderedSinceCheck = false;
  if(n)
  {
    return c->fail(PT_ERR_STATE, "BVH traversal stack overflowed %u times (the image is invalid: the acceleration structure is deeper than the traversal stack)", n);
  }
  return PT_OK;
}

// the instance records as the kernels see them: with useAnyHit(false) every instance carries FORCE_OPAQUE, which is what a hit group
// without an any-hit shader amounts to (src/rtx_pipeline.cpp:186-195)
int upload_instances(pt_context* c)
{
  std::vector<InstanceRec> inst = c->hInstances;
  if(!c->anyHit)
    for(InstanceRec& I : inst)
      I.flags |= TRI_OPAQUE;
  InstanceRec dummy{};
  return upload(c, c->dInstances, inst.empty() ? &dummy : inst.data(), sizeof(InstanceRec) * (inst.empty() ? 1 : inst.size()));
}

void refresh_scene_ptrs(pt_context* c)
{
  DeviceScene& s = c->scene;
  s.vertices     = (const float4*)c->dVertices.p;
  s.indices      = (const uint32_t*)c->dIndices.p;
  s.instances    = (const InstanceRec*)c->dInstances.p;
  s.materials    = (const pt_GltfShadeMaterial*)c->dMaterials.p;
  s.lights       = (const pt_Light*)c->dLights.p;
  s.texRecs      = (const TexRec*)c->dTexRecs.p;
  s.matLines     = (const uint4*)c->dMatLines.p;
  s.texels       = (const uint32_t*)c->dTexels.p;
  s.bvh          = (const BvhNode*)c->dBvh.p;
  s.wide         = (const WideNode*)c->dWide.p;
  s.tris         = (const TriRec*)c->dTris.p;
  s.alphaRecs    = (const AlphaRec*)c->dAlphaRecs.p;
  s.cnodes       = c->haveCNodes ? (const CompactNode*)c->dCNodes.p : nullptr;
  s.ctlas        = c->haveCNodes ? (const CompactNode*)c->dCTlas.p : nullptr;
  s.shadeTris    = c->haveShadeTris ? (const float4*)c->dShadeTris.p : nullptr;
  s.alphaMats    = (const AlphaMat*)c->dAlphaMats.p;
  s.alphaMaps    = (const uint32_t*)c->dAlphaMaps.p;
  s.env          = (const float4*)c->dEnv.p;
  s.envAccel     = (const pt_EnvAccel*)c->dEnvAccel.p;
  s.numTris      = c->numTris;
  s.numInstances = c->numInstances;
  const bool two = c->accelMode == PT_ACCEL_TWO_LEVEL && c->haveAccel && !c->mergedOnly;
  s.tlas         = two ? (const WideNode*)c->dTlas.p : nullptr;
  s.tlasLeaves   = two ? (const TlasLeaf*)c->dTlasLeaves.p : nullptr;
  s.instTriBase  = two ? (const uint32_t*)c->dInstTriBase.p : nullptr;
  s.instBlock    = two ? (const uint32_t*)c->dInstBlock.p : nullptr;
  s.twoLevel     = two ? 1u : 0u;
  s.allOpaque    = 1u;
  for(const InstanceRec& I : c->hInstances)
    if(I.triCount && c->anyHit && !(I.flags & TRI_OPAQUE))
      s.allOpaque = 0u;
}


// Where k_tail takes over (flush_pending): the first bounce whose queue is expected to hold <= tailBelow paths (maxDepth: never).
// Expectation = this launch's paths x the alive fraction observed at that bounce (ratio[0 .. numObserved), from the newest finished launch
// sequence); bounces beyond the observed ones continue the last observed shrink factor; before anything was observed a shrink of 0.3 per bounce
// is assumed (Russian roulette from depth 0 gives ~0.25 on the stand-in scenes).  A wrong guess costs time, never results.
int tail_from_depth(double paths, int maxDepth, int tailBelow, const double* ratio, int numObserved)
{
  if(tailBelow <= 0)
    return maxDepth;
  double r = 1.0, step = 0.3;
  for(int d = 0; d < maxDepth; ++d)
  {
    if(d < numObserved)
    {
      if(d > 0 && ratio[d - 1] > 0.0)
        step = std::min(1.0, ratio[d] / ratio[d - 1]);
      r = ratio[d];
    }
    else if(d > 0)
      r *= step;
    if(paths * r <= double(tailBelow))
      return d;
  }
  return maxDepth;
}

// fills the per-instance part of an InstanceRec that depends on the node's world matrix (pt_set_scene, pt_update_instances)
bool set_instance_transform(InstanceRec& I, const float* m, uint32_t materialFlags)
{
  I.objectToWorld.r0 = make_float4(m[0], m[4], m[8], m[12]);
  I.objectToWorld.r1 = make_float4(m[1], m[5], m[9], m[13]);
  I.objectToWorld.r2 = make_float4(m[2], m[6], m[10], m[14]);
  double inv[12], det3;
  if(!affine_inverse(m, inv, det3))
    return false;
  I.worldToObject.r0 = make_float4(float(inv[0]), float(inv[3]), float(inv[6]), float(inv[9]));
  I.worldToObject.r1 = make_float4(float(inv[1]), float(inv[4]), float(inv[7]), float(inv[10]));
  I.worldToObject.r2 = make_float4(float(inv[2]), float(inv[5]), float(inv[8]), float(inv[11]));
  I.flags = (materialFlags & ~TRI_FLIP) | (det3 < 0.0 ? TRI_FLIP : 0u);
  return true;
}

// Two-level walk: how far the object-space image of a world-space hit point can lie from the transformed ray (TlasLeaf::padC0 / padC1,
// pt_trace.h enter_instance).  With u = 2^-24, A = max abs row sum of the 3x3 parts, T = max |translation|, Bo = max |object coordinate|
// of the mesh, |p| <= Am Bo + Tm for every world point of the instance:
//   ray transform          <= u (7 Ainv |o| + 3 Ainv |p| + 4 Tinv)          (4-term dot products for o', 3-term for d', scaled by t |d| <= |p| + |o|)
//   inverse rounded to f32 <= u (Ainv |p| + Tinv)
//   T1 rounding of the world triangle, seen from object space <= 4 u Ainv (Am Bo + Tm)
//   the triangle test accepts points a few ulps of |p| off the triangle (the flat structure pads its leaf boxes by 67 u |p| for that)
// eps = 2^-17 (Ainv |o|  +  Ainv (Am Bo + 2 Tm) + Tinv + Bo) = 128 u (...) covers their sum with room to spare and is still ~1e-3 of a
// world unit for a scene 50 units across.
void two_level_pad(const InstanceRec& I, float Bo, float& c0, float& c1)
{
  auto rs = [](const float4& r) { return double(std::fabs(r.x)) + std::fabs(r.y) + std::fabs(r.z); };
  const double Am   = std::max(rs(I.objectToWorld.r0), std::max(rs(I.objectToWorld.r1), rs(I.objectToWorld.r2)));
  const double Tm   = std::max(std::fabs(double(I.objectToWorld.r0.w)), std::max(std::fabs(double(I.objectToWorld.r1.w)), std::fabs(double(I.objectToWorld.r2.w))));
  const double Ainv = std::max(rs(I.worldToObject.r0), std::max(rs(I.worldToObject.r1), rs(I.worldToObject.r2)));
  const double Tinv = std::max(std::fabs(double(I.worldToObject.r0.w)), std::max(std::fabs(double(I.worldToObject.r1.w)), std::fabs(double(I.worldToObject.r2.w))));
  const double k    = 1.0 / 131072.0;  // 2^-17
  const double v1 = k * Ainv, v0 = k * (Ainv * (Am * double(Bo) + 2.0 * Tm) + Tinv + double(Bo));
  c1 = std::nextafter(float(std::min(v1, 1e30)), INFINITY);
  c0 = std::nextafter(float(std::min(v0, 1e30)), INFINITY);
}

// world bounds (origin cells of the ray-sort keys) from the binary root of a hierarchy
void bounds_from_root(pt_context* c, const BvhNode& root, bool two)
{
  const float lmin[3] = {root.a.x, root.a.y, root.a.z}, lmax[3] = {root.a.w, root.b.x, root.b.y};
  const float rmin[3] = {root.b.z, root.b.w, root.c.x}, rmax[3] = {root.c.y, root.c.z, root.c.w};
  for(int k = 0; k < 3; ++k)
  {
    const float mn = two ? std::min(lmin[k], rmin[k]) : lmin[k], mx = two ? std::max(lmax[k], rmax[k]) : lmax[k];
    c->scene.boundsMin[k]    = std::isfinite(mn) ? mn : 0.f;
    c->scene.boundsInvExt[k] = (std::isfinite(mx - mn) && mx > mn) ? 1.0f / (mx - mn) : 0.f;
  }
}

// the instance records as the kernels see them (see upload_instances)
std::vector<InstanceRec> effective_instances(const pt_context* c)
{
  std::vector<InstanceRec> inst = c->hInstances;
  if(!c->anyHit)
    for(InstanceRec& I : inst)
      I.flags |= TRI_OPAQUE;
  return inst;
}

void build_cnodes(pt_context* c, uint32_t n);
void build_cnodes_two_level(pt_context* c);
void build_shade_tris(pt_context* c, uint32_t n);
// TLAS of the two-level structure over the current instance transforms (also the refit after pt_update_instances: the BLASes stay)
int build_tlas(pt_context* c)
{
  const std::vector<InstanceRec> inst = effective_instances(c);
  std::vector<uint32_t>          active, triBase(inst.empty() ? 1 : inst.size(), 0u);
  std::vector<float>             pad(inst.empty() ? 2 : 2 * inst.size(), 0.f);
  std::vector<char> isMerged(inst.size(), 0);
  for(uint32_t i : c->hMerged)
    isMerged[i] = 1;
  for(uint32_t i = 0; i < inst.size(); ++i)
  {
    triBase[i] = inst[i].triBase;
    if(inst[i].triCount == 0 || isMerged[i])
      continue;
    active.push_back(i);
    two_level_pad(inst[i], c->hPrimBound[inst[i].primMesh], pad[2 * i], pad[2 * i + 1]);
  }
  c->numActive = uint32_t(active.size());
  int rc;
  const uint32_t none = 0;
  if((rc = upload(c, c->dActive, active.empty() ? &none : active.data(), 4 * std::max<size_t>(1, active.size()))) != PT_OK) return rc;
  if((rc = upload(c, c->dInstTriBase, triBase.data(), 4 * triBase.size())) != PT_OK) return rc;
  {  // block table of instance_of_world_tri (pt_trace.h): entry e = the last instance whose triBase <= e << PT_INST_BLOCK_SHIFT
    const size_t          entries = (size_t(c->numTris) >> PT_INST_BLOCK_SHIFT) + 2;
    std::vector<uint32_t> block(entries, 0u);
    uint32_t              at = 0;
    for(size_t e = 0; e < entries; ++e)
    {
      const uint64_t first = uint64_t(e) << PT_INST_BLOCK_SHIFT;
      while(at + 1 < triBase.size() && uint64_t(triBase[at + 1]) <= first)
        ++at;
      block[e] = at;
    }
    if((rc = upload(c, c->dInstBlock, block.data(), 4 * block.size())) != PT_OK) return rc;
  }
  if((rc = upload(c, c->dInstPad, pad.data(), 4 * pad.size())) != PT_OK) return rc;
  if((rc = upload(c, c->dInstNodeBase, c->hInstNodeBase.empty() ? &none : c->hInstNodeBase.data(), 4 * std::max<size_t>(1, c->hInstNodeBase.size()))) != PT_OK) return rc;
  const uint32_t numPrims = c->numActive + (c->mergedTris ? 1u : 0u);
  if((rc = dev_alloc(c, c->dTlas, sizeof(WideNode) * size_t(std::max(1u, numPrims)))) != PT_OK) return rc;
  if((rc = dev_alloc(c, c->dTlasLeaves, sizeof(TlasLeaf) * size_t(std::max(1u, numPrims)))) != PT_OK) return rc;
  auto    t0 = std::chrono::steady_clock::now();
  char    msg[256];
  BvhNode root{};
  if(pt_tlas_build(c->stream, c->tune, (const InstanceRec*)c->dInstances.p, (const uint32_t*)c->dActive.p, c->numActive, (const uint32_t*)c->dInstNodeBase.p, (const float*)c->dInstPad.p,
                   (const float4*)c->dVertices.p, (const uint32_t*)c->dIndices.p, (WideNode*)c->dTlas.p, (TlasLeaf*)c->dTlasLeaves.p, &root, &c->numTlasNodes, msg, sizeof(msg),
                   c->mergedTris ? c->mergedBox : nullptr, 0u) != 0)
    return c->fail(PT_ERR_HIP, "TLAS build: %s", msg);
  c->msBuildTlas = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for(int k = 0; k < 3; ++k)
    c->scene.boundsMin[k] = c->scene.boundsInvExt[k] = 0.f;
  if(numPrims > 0)
    bounds_from_root(c, root, numPrims > 1 && root.d.y != BVH_NONE);
  c->mergedOnly = c->mergedTris > 0 && c->numActive == 0;
  if(c->mergedOnly)
    build_cnodes(c, c->mergedWide);  // the flat kernels run on the merged structure
  else
    build_cnodes_two_level(c);
  build_shade_tris(c, c->mergedOnly ? c->mergedTris : 0u);
  return PT_OK;
}

// compact nodes of a real two-level structure: every bottom-level structure at its node base, and the TLAS
void build_cnodes_two_level(pt_context* c)
{
  c->haveCNodes = false;
  if(!c->tune.cnodes || c->nodeCapacity == 0 || c->numTlasNodes == 0)
  {
    dev_free(c->dCNodes);
    dev_free(c->dCTlas);
    return;
  }
  if(dev_alloc(c, c->dCNodes, sizeof(CompactNode) * size_t(c->nodeCapacity)) != PT_OK || dev_alloc(c, c->dCTlas, sizeof(CompactNode) * size_t(c->numTlasNodes)) != PT_OK)
  {
    (void)hipGetLastError();
    return;
  }
  std::vector<uint32_t> ranges = c->hBlasRanges;
  if(c->mergedWide)
  {
    ranges.push_back(0u);
    ranges.push_back(c->mergedWide);
  }
  c->haveCNodes = pt_compact_node_ranges(c->stream, ranges.data(), uint32_t(ranges.size() / 2), (const WideNode*)c->dWide.p, (CompactNode*)c->dCNodes.p) == 0 &&
                  pt_compact_nodes(c->stream, c->numTlasNodes, (const WideNode*)c->dTlas.p, (CompactNode*)c->dCTlas.p) == 0;
}

// DeviceScene::cnodes over the first n wide nodes of a flat-format structure (best effort: without it the kernels walk the WideNodes)
void build_cnodes(pt_context* c, uint32_t n)
{
  c->haveCNodes = false;
  if(!c->tune.cnodes || n == 0)
  {
    dev_free(c->dCNodes);
    return;
  }
  if(dev_alloc(c, c->dCNodes, sizeof(CompactNode) * size_t(n)) != PT_OK)
  {
    (void)hipGetLastError();
    return;
  }
  c->haveCNodes = pt_compact_nodes(c->stream, n, (const WideNode*)c->dWide.p, (CompactNode*)c->dCNodes.p) == 0;
}
// DeviceScene::shadeTris over the first n leaf records of a flat-format structure (best effort: without the memory k_shade takes the indexed route)
void build_shade_tris(pt_context* c, uint32_t n)
{
  c->haveShadeTris = false;
  if(!c->tune.shadeTris || n == 0)
  {
    dev_free(c->dShadeTris);
    return;
  }
  if(dev_alloc(c, c->dShadeTris, sizeof(float4) * PT_SHADE_REC_QUADS * size_t(n)) != PT_OK)
  {
    (void)hipGetLastError();
    return;
  }
  pt_launch_shade_tris(c->stream, n, (const TriRec*)c->dTris.p, (const InstanceRec*)c->dInstances.p, (const float4*)c->dVertices.p, (const uint32_t*)c->dIndices.p,
                       (float4*)c->dShadeTris.p);
  c->haveShadeTris = hipStreamSynchronize(c->stream) == hipSuccess && hipGetLastError() == hipSuccess;
}

// (re)builds the merged world-space structure over c->hMerged with the current transforms, in place at slot 0 / node 0 of the BLAS arrays
int build_merged(pt_context* c)
{
  const uint32_t before = c->mergedWide;
  struct KeepStat {  // the wide-node statistic follows the merged structure's size on every (re)build, refits included
    pt_context* c; uint32_t before;
    ~KeepStat() { c->numWideNodes = c->numWideNodes - std::min(c->numWideNodes, before) + c->mergedWide; }
  } keep{c, before};
  c->mergedWide = 0;
  if(c->hMerged.empty())
    return PT_OK;
  const std::vector<InstanceRec> inst = effective_instances(c);
  std::vector<InstanceRec>       sub;
  std::vector<uint32_t>          worldBase;
  uint32_t                       n = 0;
  for(uint32_t i : c->hMerged)
  {
    InstanceRec I = inst[i];
    worldBase.push_back(I.triBase);
    I.triBase = n;
    n += I.triCount;
    sub.push_back(I);
  }
  char msg[256];
  if(pt_merged_build(c->stream, c->tune, sub.data(), c->hMerged.data(), worldBase.data(), uint32_t(sub.size()), n, (const float4*)c->dVertices.p, (const uint32_t*)c->dIndices.p, (TriRec*)c->dTris.p,
                     (AlphaRec*)c->dAlphaRecs.p, (WideNode*)c->dWide.p, 0u, 0u, &c->mergedWide, c->mergedBox, msg, sizeof(msg)) != 0)
    return c->fail(PT_ERR_HIP, "pt_build_accel (two-level, merged structure): %s", msg);
  return PT_OK;
}

// AccelStructure::create as the reference does it [src/accelstruct.cpp:110-162]: one BLAS per prim-mesh that some node instantiates, one TLAS
// instance per node
int build_two_level(pt_context* c)
{
  const std::vector<InstanceRec> inst = effective_instances(c);
  std::map<int32_t, uint32_t>    blasOf;
  std::vector<PtBlasDesc>        blas;
  uint64_t                       slots = 0, nodes = 0;
  c->hInstNodeBase.assign(inst.size(), 0u);
  // prim-meshes instantiated once: their instances share one world-space structure, first in the arrays
  c->hMerged.clear();
  c->mergedTris = 0;
  c->mergedOnly = false;
  std::vector<char> isMerged(inst.size(), 0);
  if(c->tune.mergeSingles)
  {
    std::map<int32_t, uint32_t> uses;
    for(const InstanceRec& I : inst)
      if(I.triCount)
        uses[I.primMesh]++;
    for(uint32_t i = 0; i < inst.size(); ++i)
      if(inst[i].triCount && uses[inst[i].primMesh] == 1)
      {
        c->hMerged.push_back(i);
        isMerged[i] = 1;
        c->mergedTris += inst[i].triCount;
      }
    slots = c->mergedTris;
    nodes = c->mergedTris ? std::max(1u, c->mergedTris - 1) : 0;
  }
  for(uint32_t i = 0; i < inst.size(); ++i)
  {
    const InstanceRec& I = inst[i];
    if(I.triCount == 0 || isMerged[i])
      continue;
    auto it = blasOf.find(I.primMesh);
    if(it == blasOf.end())
    {
      PtBlasDesc d{};
      d.primMesh = uint32_t(I.primMesh); d.vertexOffset = I.vertexOffset; d.firstIndex = I.firstIndex; d.triCount = I.triCount;
      d.flags = I.flags & ~TRI_FLIP; d.materialIndex = I.materialIndex;
      d.slotBase = uint32_t(slots); d.nodeBase = uint32_t(nodes);
      slots += I.triCount;
      nodes += std::max(1u, I.triCount - 1);
      it = blasOf.emplace(I.primMesh, uint32_t(blas.size())).first;
      blas.push_back(d);
    }
    c->hInstNodeBase[i] = blas[it->second].nodeBase;
  }
  if(slots > BVH_SLOT_MASK || nodes > BVH_SLOT_MASK)
    return c->fail(PT_ERR_INVALID, "two-level structure: %llu distinct triangles exceed the reference range", (unsigned long long)slots);
  int rc;
  if((rc = dev_alloc(c, c->dTris, sizeof(TriRec) * size_t(std::max<uint64_t>(1, slots)))) != PT_OK) return rc;
  if((rc = dev_alloc(c, c->dAlphaRecs, sizeof(AlphaRec) * size_t(std::max<uint64_t>(1, slots)))) != PT_OK) return rc;
  if((rc = dev_alloc(c, c->dWide, sizeof(WideNode) * size_t(std::max<uint64_t>(1, nodes)))) != PT_OK) return rc;
  dev_free(c->dBvh);  // the binary nodes are a build temporary here
  auto t0 = std::chrono::steady_clock::now();
  char msg[256];
  if(pt_blas_build(c->stream, c->tune, blas.data(), uint32_t(blas.size()), (const float4*)c->dVertices.p, (const uint32_t*)c->dIndices.p, (TriRec*)c->dTris.p, (AlphaRec*)c->dAlphaRecs.p,
                   (WideNode*)c->dWide.p, msg, sizeof(msg)) != 0)
    return c->fail(PT_ERR_HIP, "pt_build_accel (two-level): %s", msg);
  if((rc = build_merged(c)) != PT_OK)
    return rc;
  c->hBlasRanges.clear();
  for(const PtBlasDesc& d : blas)
  {
    c->hBlasRanges.push_back(d.nodeBase);
    c->hBlasRanges.push_back(d.numWide);
  }
  c->nodeCapacity = uint32_t(std::max<uint64_t>(1, nodes));
  c->numBlas      = uint32_t(blas.size()) + (c->mergedTris ? 1u : 0u);
  c->numBvhNodes  = uint32_t(nodes);
  c->numWideNodes = c->mergedWide;  // (build_merged above already counted it into the old total: start over)
  for(const PtBlasDesc& d : blas)
    c->numWideNodes += d.numWide;
  if((rc = build_tlas(c)) != PT_OK)
    return rc;
  HIP_TRY(c, sync_all(c));
  c->msBuild   = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  c->haveAccel = true;
  c->warmPending = true;
  refresh_scene_ptrs(c);
  return PT_OK;
}
}  // namespace

// ---- stage timers ------------------------------------------------------------------------------------------
void pt_timers_begin(StageTimers* t, hipStream_t s, int stage)
{
  if(!t || !t->enabled)
    return;
  if(t->npend == t->cap)
  {
    size_t ncap = t->cap ? t->cap * 2 : 256;
    auto*  np   = (StageTimers::Pending*)realloc(t->pend, ncap * sizeof(StageTimers::Pending));
    if(!np)
      return;
    for(size_t i = t->cap; i < ncap; ++i)
    {
      (void)hipEventCreate(&np[i].a);
      (void)hipEventCreate(&np[i].b);
    }
    t->pend = np;
    t->cap  = ncap;
  }
  t->pend[t->npend].stage = stage;
  (void)hipEventRecord(t->pend[t->npend].a, s);
}
void pt_timers_end(StageTimers* t, hipStream_t s, int stage)
{
  if(!t || !t->enabled || t->npend >= t->cap)
    return;
  (void)hipEventRecord(t->pend[t->npend].b, s);
  if(stage == 1)
    t->launchesClosest++;
  if(stage == 5)
    t->launchesTail++;
  if(stage == 6)
    t->launchesFused++;
  t->npend++;
  if(t->npend == t->cap && t->cap >= 8192)
    pt_timers_collect(t);  // bound the number of live events
}
void pt_timers_collect(StageTimers* t)
{
  if(!t || !t->npend)
    return;
  (void)hipEventSynchronize(t->pend[t->npend - 1].b);
  for(size_t i = 0; i < t->npend; ++i)
  {
    float ms = 0.f;
    if(hipEventElapsedTime(&ms, t->pend[i].a, t->pend[i].b) == hipSuccess)
      t->ms[t->pend[i].stage] += ms;
  }
  t->npend = 0;
}

// PT_TUNE -> PtTuning, key by key (pt_internal.h)
void pt_parse_tuning(const char* tune, PtTuning& t, std::string& unknown)
{
  if(!tune)
    return;
  struct Key { const char* name; int PtTuning::*field; };
  static const Key keys[] = {{"stateMB", &PtTuning::stateMB}, {"stateGB", &PtTuning::stateGB}, {"packetClosest", &PtTuning::packetClosestBounces}, {"mergeSingles", &PtTuning::mergeSingles},
                             {"cnodes", &PtTuning::cnodes}, {"shadeTris", &PtTuning::shadeTris}, {"tail", &PtTuning::tailBelow}, {"warm", &PtTuning::warm}, {"texTile", &PtTuning::texTile},
                             {"texGroups", &PtTuning::texGroups}, {"regen", &PtTuning::regen}, {"packetTwo", &PtTuning::packetTwo}, {"blasWorkers", &PtTuning::blasWorkers},
                             {"batch", &PtTuning::batch}, {"inflight", &PtTuning::framesInFlight}, {"displaySlots", &PtTuning::displaySlots}, {"bands", &PtTuning::bands},
                             {"bandTiles", &PtTuning::bandTiles}, {"fuse", &PtTuning::fuse}};
  const std::string all(tune);
  size_t            at = 0;
  while(at <= all.size())
  {
    size_t end = all.find(',', at);
    if(end == std::string::npos)
      end = all.size();
    std::string tok = all.substr(at, end - at);
    at              = end + 1;
    while(!tok.empty() && (tok.front() == ' ' || tok.front() == '\t'))
      tok.erase(tok.begin());
    while(!tok.empty() && (tok.back() == ' ' || tok.back() == '\t'))
      tok.pop_back();
    if(tok.empty())
      continue;
    const size_t      eq  = tok.find('=');
    const std::string key = tok.substr(0, eq), val = eq == std::string::npos ? std::string() : tok.substr(eq + 1);
    bool              ok  = false;
    if(key == "build")
    {
      ok = true;
      if(val == "lbvh") t.sahBuild = 0;
      else if(val == "sah") t.sahBuild = 1;
      else if(val == "ploc") t.sahBuild = 2;
      else if(val == "sahdev") t.sahBuild = 3;
      else ok = false;
    }
    else if(key == "accel")
    {
      ok = val == "two" || val == "flat";
      if(ok)
        t.accelTwoLevel = val == "two" ? 1 : 0;
    }
    else
      for(const Key& k : keys)
        if(key == k.name)
        {
          char*      e = nullptr;
          const long v = std::strtol(val.c_str(), &e, 10);
          if(!val.empty() && e && *e == 0)
          {
            t.*(k.field) = int(v);
            ok           = true;
          }
          break;
        }
    if(!ok)
      unknown += (unknown.empty() ? "" : ",") + tok;
  }
  if(t.bandTiles < 1)
    t.bandTiles = 1;
}

// test hook (no GPU involved): parses `tune` as pt_create would and reports the knobs in the order of the keys below plus build / accel;
// `unknown` receives the tokens that name no knob.  Returns the number of values written.
extern "C" __attribute__((visibility("default"))) int pt_debug_parse_tuning(const char* tune, int* out, int maxOut, char* unknownOut, size_t unknownLen)
{
  PtTuning    t;
  std::string unknown;
  pt_parse_tuning(tune, t, unknown);
  const int v[] = {t.stateMB, t.stateGB, t.packetClosestBounces, t.mergeSingles, t.cnodes, t.shadeTris, t.tailBelow, t.warm, t.texTile, t.texGroups, t.regen, t.packetTwo,
                   t.blasWorkers, t.batch, t.framesInFlight, t.displaySlots, t.bands, t.bandTiles, t.fuse, t.sahBuild, t.accelTwoLevel};
  const int n   = int(sizeof(v) / sizeof(v[0]));
  for(int i = 0; i < n && i < maxOut; ++i)
    out[i] = v[i];
  if(unknownOut && unknownLen)
    snprintf(unknownOut, unknownLen, "%s", unknown.c_str());
  return n < maxOut ? n : maxOut;
}

extern "C" {

const char* pt_renderer_name(void) { return "HIP"; }

const char* pt_last_error(const pt_context* ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

int pt_create(int device_ordinal, pt_context** out_ctx)
{
  if(!out_ctx)
    return PT_ERR_INVALID;
  *out_ctx  = nullptr;
  // frames in flight need one hardware queue each; effective only if the HIP runtime is not initialised yet
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if(e != hipSuccess || count <= 0)
  {
    g_createError = std::string("no HIP device available: ") + hipGetErrorString(e) + " (libptmi has no CPU fallback)";
    return PT_ERR_NO_DEVICE;
  }
  if(device_ordinal < 0 || device_ordinal >= count)
  {
    g_createError = "device ordinal " + std::to_string(device_ordinal) + " out of range: " + std::to_string(count) + " HIP device(s) visible to this process";
    return PT_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if(hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess)
  {
    g_createError = "hipGetDeviceProperties failed";
    return PT_ERR_HIP;
  }
  if(std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
  {
    g_createError = std::string("device is ") + prop.gcnArchName + ", libptmi.so is built for gfx950 only";
    return PT_ERR_NO_DEVICE;
  }
  if(hipSetDevice(device_ordinal) != hipSuccess)
  {
    g_createError = "hipSetDevice failed";
    return PT_ERR_HIP;
  }
  // every context has its own knobs (round 6): parsed key by key, defaults otherwise; tokens that name no knob are reported once per process
  PtTuning    parsed;
  std::string unknown;
  pt_parse_tuning(getenv("PT_TUNE"), parsed, unknown);
  if(!unknown.empty())
  {
    static std::atomic<bool> warned{false};
    if(!warned.exchange(true))
      fprintf(stderr, "libptmi: PT_TUNE tokens that name no knob (ignored; removed knobs are constants now, see csrc/pt_internal.h PtTuning): %s\n", unknown.c_str());
  }
  pt_context* c = new pt_context();
  c->tune       = parsed;
  c->device     = device_ordinal;
  c->accelMode  = c->tune.accelTwoLevel ? PT_ACCEL_TWO_LEVEL : PT_ACCEL_FLAT;
  if(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
  {
    g_createError = "hipStreamCreate failed";
    delete c;
    return PT_ERR_HIP;
  }
  c->timers.stream = c->stream;
  c->inflight      = c->tune.framesInFlight < 1 ? 1 : (c->tune.framesInFlight > PT_MAX_INFLIGHT ? PT_MAX_INFLIGHT : c->tune.framesInFlight);
  c->inflightMax = c->inflight;
  c->displaySlotsMax = std::max(0, std::min(c->tune.displaySlots, PT_MAX_INFLIGHT - c->inflightMax));
  for(int i = 0; i < c->inflightMax + c->displaySlotsMax; ++i)
    if(hipStreamCreateWithFlags(&c->slots[i].stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->slots[i].accumDone, hipEventDisableTiming) != hipSuccess ||
       hipEventCreateWithFlags(&c->slots[i].countsDone, hipEventDisableTiming) != hipSuccess ||
       hipHostMalloc((void**)&c->slots[i].hCounts, sizeof(uint32_t) * CNT_STRIDE * (PT_MAX_DEPTH + 2)) != hipSuccess)
    {
      g_createError = "hipStreamCreate / hipEventCreate failed";
      delete c;
      return PT_ERR_HIP;
    }
  // defaults: sun & sky off, empty camera
  std::memset(&c->scene, 0, sizeof(c->scene));
  if(dev_alloc(c, c->dCounters, sizeof(Counters)) != PT_OK || hipMemset(c->dCounters.p, 0, sizeof(Counters)) != hipSuccess)
  {
    g_createError = c->err;
    delete c;
    return PT_ERR_HIP;
  }
  *out_ctx = c;
  return PT_OK;
}

int pt_destroy(pt_context* c)
{
  CTX_CHECK(c);
  (void)hipSetDevice(c->device);
  (void)sync_all(c);
  DevBuf* all[] = {&c->dMatLines, &c->dVertices, &c->dIndices, &c->dInstances, &c->dMaterials, &c->dLights, &c->dTexRecs, &c->dTexels, &c->dBvh, &c->dWide, &c->dTris, &c->dAlphaRecs, &c->dCNodes, &c->dCTlas, &c->dInstBlock, &c->dShadeTris, &c->dAlphaMats, &c->dAlphaMaps, &c->dPick, &c->dEnv,
                   &c->dTlas, &c->dTlasLeaves, &c->dInstTriBase, &c->dActive, &c->dInstNodeBase, &c->dInstPad, &c->dEnvAccel, &c->dFrame, &c->dSlotTile, &c->dCounters, &c->dRowMajor, &c->dRgba8,
                   &c->dMean, &c->dMips, &c->dGather, &c->dFullTiles, &c->dFullSlotTile, &c->dTileLocalIndex};
  for(DevBuf* b : all)
    dev_free(*b);
  for(auto& fs : c->slots)
  {
    for(DevBuf& b : fs.dState)
      dev_free(b);
    DevBuf* q[] = {&fs.dQueueA, &fs.dQueueB, &fs.dQueueS, &fs.dQueueX, &fs.dQueueX2, &fs.dQueueR, &fs.dCounts, &fs.dCountsDone};
    for(DevBuf* b : q)
      dev_free(*b);
    if(fs.accumDone)
      (void)hipEventDestroy(fs.accumDone);
    if(fs.countsDone)
      (void)hipEventDestroy(fs.countsDone);
    if(fs.hCounts)
      (void)hipHostFree(fs.hCounts);
    if(fs.stream)
      (void)hipStreamDestroy(fs.stream);
  }
  for(size_t i = 0; i < c->timers.cap; ++i)
  {
    (void)hipEventDestroy(c->timers.pend[i].a);
    (void)hipEventDestroy(c->timers.pend[i].b);
  }
  free(c->timers.pend);
  for(auto& ds : c->display)
  {
    if(ds.host)
      (void)hipHostFree(ds.host);
    if(ds.done)
      (void)hipEventDestroy(ds.done);
    if(ds.read)
      (void)hipEventDestroy(ds.read);
  }
  (void)hipStreamDestroy(c->stream);
  delete c;
  return PT_OK;
}

// Opacity maps (pt_device.h): for every non-opaque material whose base-colour texture takes the fast tap, classify each
// ALPHA_MAP_BLOCK^2 block of base texels + one texel of apron on every side (a bilinear tap based in the block blends
// texels of that window only).  A state is assigned only when every texel of the window decides the same way with a
// 1e-5 relative margin -- two orders above the fp32 filtering error -- so the map never changes a result.
static void build_opacity_maps(const pt_SceneDesc* d, std::vector<AlphaMat>& am, std::vector<uint32_t>& words)
{
  struct Key {
    int   tex, mode;
    float factor, cutoff;
    bool  operator<(const Key& o) const { return std::tie(tex, mode, factor, cutoff) < std::tie(o.tex, o.mode, o.factor, o.cutoff); }
  };
  std::map<Key, uint32_t> done;
  words.assign(1, 0u);  // never empty (word 0 is unused padding)
  for(size_t m = 0; m < am.size(); ++m)
  {
    AlphaMat& a = am[m];
    if(a.mode == PT_ALPHA_OPAQUE || a.tex < 0 || !(a.texWrap & ALPHA_FAST_TAP) || a.texW < ALPHA_MAP_BLOCK || a.texH < ALPHA_MAP_BLOCK)
      continue;
    if(!(a.factorA >= 0.0f && a.factorA <= 3.0e38f) || !(std::fabs(a.cutoff) <= 3.0e38f))
      continue;
    const Key key{a.tex, a.mode, a.factorA, a.cutoff};
    auto      it = done.find(key);
    if(it != done.end())
    {
      a.mapOffset = it->second;
      continue;
    }
    const int      W = a.texW, H = a.texH, bw = W >> ALPHA_MAP_SHIFT, bh = H >> ALPHA_MAP_SHIFT;
    const uint8_t* px = (const uint8_t*)d->textures[a.tex].rgba8;
    // separable min / max of the alpha byte over [b*B - 1, b*B + B] (wrapped)
    std::vector<uint8_t> rmin(size_t(bw) * H), rmax(size_t(bw) * H);
    for(int y = 0; y < H; ++y)
      for(int bx = 0; bx < bw; ++bx)
      {
        uint8_t lo = 255, hi = 0;
        for(int k = -1; k <= ALPHA_MAP_BLOCK; ++k)
        {
          const uint8_t v = px[(size_t(y) * W + ((bx * ALPHA_MAP_BLOCK + k) & (W - 1))) * 4 + 3];
          lo = v < lo ? v : lo;
          hi = v > hi ? v : hi;
        }
        rmin[size_t(y) * bw + bx] = lo;
        rmax[size_t(y) * bw + bx] = hi;
      }
    const uint32_t off = uint32_t(words.size());
    words.resize(words.size() + (size_t(bw) * bh + 15) / 16, 0u);
    const double f = a.factorA, cut = a.cutoff;
    for(int by = 0; by < bh; ++by)
      for(int bx = 0; bx < bw; ++bx)
      {
        uint8_t lo = 255, hi = 0;
        for(int k = -1; k <= ALPHA_MAP_BLOCK; ++k)
        {
          const int y = (by * ALPHA_MAP_BLOCK + k) & (H - 1);
          lo = rmin[size_t(y) * bw + bx] < lo ? rmin[size_t(y) * bw + bx] : lo;
          hi = rmax[size_t(y) * bw + bx] > hi ? rmax[size_t(y) * bw + bx] : hi;
        }
        const double vmin = f * lo / 255.0, vmax = f * hi / 255.0;
        uint32_t     st = ALPHA_ST_UNKNOWN;
        if(a.mode == PT_ALPHA_MASK)
        {
          if(vmin > cut + 1e-5 * std::fmax(std::fabs(cut), vmin))
            st = ALPHA_ST_ONE;
          else if((hi == 0 && cut >= 0.0) || vmax < cut - 1e-5 * std::fmax(std::fabs(cut), vmax))
            st = ALPHA_ST_ZERO;
        }
        else  // BLEND: opacity = factor x filtered alpha
        {
          if(hi == 0 || f == 0.0)
            st = ALPHA_ST_ZERO;
          else if(vmin >= 1.0 + 1e-5)
            st = ALPHA_ST_ONE;
        }
        const uint32_t bidx = uint32_t(by) * uint32_t(bw) + uint32_t(bx);
        words[off + (bidx >> 4)] |= st << ((bidx & 15u) * 2u);
      }
    a.mapOffset = off;
    done[key]   = off;
  }
}

// Everything pt_set_scene derives from a pt_SceneDesc on the HOST, before anything is uploaded: validation, the per-instance records
// (transforms, inverse, TLAS flags of src/accelstruct.cpp:144-149), the texture records and the compact alpha view of the materials with their
// opacity maps.  Shared with the test hook pt_debug_scene_records (CPU tests run the product's traversal on exactly these records).
struct SceneRecords {
  std::vector<InstanceRec> inst;
  uint64_t                 triTotal = 0;
  std::vector<float>       primBound;  // per prim-mesh: max |coordinate| of its vertices
  std::vector<TexRec>      texRecs;    // >= 1 (a 1x1 white default when the scene has no texture, src/scene.cpp:513-519)
  size_t                   texels = 0; // texels of the RGBA8 pool
  std::vector<AlphaMat>    alphaMats;
  std::vector<uint32_t>    alphaMaps;
  // interleaved texture groups (pt_device.h TexRec::tiled): the pool holds every texture in its plain form first, then the groups
  struct TexGroup {
    int      tex[4];  // texture ids in layer order (-1: unused layer)
    int      layers;
    uint32_t offset;  // first texel word of the group in the pool
  };
  std::vector<TexGroup> groups;
  std::vector<uint4>    matLines;  // PT_MAT_LINE_QUADS per material (pt_device.h mat_line_pack)
};
__attribute__((format(printf, 2, 3))) static int records_fail(std::string& err, const char* fmt, ...)
{
  char    buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  err = buf;
  return PT_ERR_INVALID;
}
// the texels of image `td` in the storage order its record says (row-major source -> row-major or block-linear, pt_device.h tex_index)
static void store_texture(uint32_t* dst, const TexRec& tr, const void* rgba8RowMajor)
{
  const uint32_t* src = static_cast<const uint32_t*>(rgba8RowMajor);
  if(!tr.tiled)
  {
    std::memcpy(dst, src, size_t(tr.w) * tr.h * 4);
    return;
  }
  for(int y = 0; y < tr.h; ++y)
    for(int x = 0; x < tr.w; x += PT_TEX_TILE_W)
      std::memcpy(dst + tex_index(tr.w, x, y, true), src + size_t(y) * tr.w + x, PT_TEX_TILE_W * 4);
}

// the texels of a group: texel (x, y) of layer l at tex_index(...) x layers + l
static void store_group(uint32_t* dst, const SceneRecords::TexGroup& g, const std::vector<TexRec>& texRecs, const pt_SceneDesc* d)
{
  for(int l = 0; l < g.layers; ++l)
  {
    const TexRec&   tr  = texRecs[size_t(g.tex[l])];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(d->textures[g.tex[l]].rgba8);
    for(int y = 0; y < tr.h; ++y)
      for(int x = 0; x < tr.w; ++x)
        dst[size_t(tex_index(tr.w, x, y, (tr.tiled & 1) != 0)) * size_t(g.layers) + size_t(l)] = src[size_t(y) * tr.w + x];
  }
}

static int build_scene_records(const pt_SceneDesc* d, SceneRecords& R, std::string& err, int texTile, int texGroups)
{
  if(!d || !d->vertices || !d->indices || !d->primMeshes || !d->nodes || !d->materials || d->numMaterials == 0)
    return records_fail(err, "pt_set_scene: null array or no material");
  if((d->numLights && !d->lights) || (d->numTextures && !d->textures))
    return records_fail(err, "pt_set_scene: count without array");
  // ---- validate + build the per-instance records
  R.inst.assign(d->numNodes, InstanceRec{});
  R.triTotal = 0;
  for(uint32_t n = 0; n < d->numNodes; ++n)
  {
    const pt_Node& nd = d->nodes[n];
    if(nd.primMesh < 0 || uint32_t(nd.primMesh) >= d->numPrimMeshes)
      return records_fail(err, "node %u: primMesh %d out of range", n, nd.primMesh);
    const pt_PrimMesh& pm = d->primMeshes[nd.primMesh];
    if(pm.materialIndex >= int(d->numMaterials))
      return records_fail(err, "primMesh %d: materialIndex %d out of range", nd.primMesh, pm.materialIndex);
    if(uint64_t(pm.vertexOffset) + pm.vertexCount > d->numVertices || uint64_t(pm.firstIndex) + pm.indexCount > d->numIndices || pm.indexCount % 3)
      return records_fail(err, "primMesh %d: vertex/index range out of bounds", nd.primMesh);
    const pt_GltfShadeMaterial& mat = d->materials[pm.materialIndex < 0 ? 0 : pm.materialIndex];
    InstanceRec&                I   = R.inst[n];
    // instance flags of the reference's TLAS (src/accelstruct.cpp:144-149)
    uint32_t flags = 0;
    if(mat.alphaMode == 0 || (mat.pbrBaseColorFactor[3] == 1.0f && mat.pbrBaseColorTexture == -1))
      flags |= TRI_OPAQUE;
    if(mat.doubleSided == 1)
      flags |= TRI_NOCULL;
    if(!set_instance_transform(I, nd.worldMatrix, flags))  // + TRI_FLIP for a mirroring matrix
      return records_fail(err, "node %u: singular world matrix", n);
    I.vertexOffset  = pm.vertexOffset;
    I.firstIndex    = pm.firstIndex;
    I.materialIndex = pm.materialIndex;
    I.primMesh      = nd.primMesh;
    I.triBase       = uint32_t(R.triTotal);
    I.triCount      = pm.indexCount / 3;
    I._pad  = 0;
    R.triTotal += I.triCount;
  }
  if(R.triTotal > TRI_INDEX_MASK)
    return records_fail(err, "scene has %llu triangles; the limit is %u", (unsigned long long)R.triTotal, TRI_INDEX_MASK);
  R.primBound.assign(d->numPrimMeshes, 0.f);
  for(uint32_t p = 0; p < d->numPrimMeshes; ++p)
  {
    const pt_PrimMesh& pm = d->primMeshes[p];
    for(uint32_t k = 0; k < pm.indexCount; ++k)
      if(d->indices[pm.firstIndex + k] >= pm.vertexCount)
        return records_fail(err, "primMesh %u: index %u >= vertexCount", p, d->indices[pm.firstIndex + k]);
    float b = 0.f;
    for(uint32_t v = 0; v < pm.vertexCount; ++v)
    {
      const float* q = d->vertices[pm.vertexOffset + v].position;
      for(int a = 0; a < 3; ++a)
        if(std::isfinite(q[a]))
          b = std::max(b, std::fabs(q[a]));
    }
    R.primBound[p] = b;
  }
  for(uint32_t m = 0; m < d->numMaterials; ++m)
  {
    const pt_GltfShadeMaterial& mt = d->materials[m];
    const int ids[] = {mt.pbrBaseColorTexture, mt.pbrMetallicRoughnessTexture, mt.emissiveTexture, mt.normalTexture, mt.transmissionTexture, mt.clearcoatTexture, mt.clearcoatRoughnessTexture};
    for(int id : ids)
      if(id >= int(d->numTextures))
        return records_fail(err, "material %u references texture %d of %u", m, id, d->numTextures);
  }
  // ---- texture records (one RGBA8 pool)
  R.texRecs.assign(d->numTextures ? d->numTextures : 1, TexRec{});
  R.texels = 0;
  for(uint32_t t = 0; t < d->numTextures; ++t)
  {
    const pt_TextureDesc& td = d->textures[t];
    if(!td.rgba8 || td.width <= 0 || td.height <= 0)
      return records_fail(err, "texture %u: empty image", t);
    if(td.width > 65535 || td.height > 65535)  // (pt_device.h tex_index multiplies row x stride in 24 bits)
      return records_fail(err, "texture %u: %d x %d exceeds 65535 texels a side (tex_desc_pack keeps a side in 16 bits)", t, td.width, td.height);
    R.texRecs[t].tiled  = (texTile && td.width % PT_TEX_TILE_W == 0 && td.height % PT_TEX_TILE_H == 0) ? 1 : 0;
    if(R.texRecs[t].tiled)
      R.texels = (R.texels + 31u) & ~size_t(31);  // a tile = one 128-byte line (the pool itself is 256-byte aligned)
    R.texRecs[t].offset = uint32_t(R.texels);
    R.texRecs[t].w      = td.width;
    R.texRecs[t].h      = td.height;
    R.texRecs[t].mag    = td.magFilter;
    R.texRecs[t].wrapS  = td.wrapS;
    R.texRecs[t].wrapT  = td.wrapT;
    R.texRecs[t].pot    = ((td.width & (td.width - 1)) == 0 ? 1 : 0) | ((td.height & (td.height - 1)) == 0 ? 2 : 0);
    R.texels += size_t(td.width) * td.height;
    if(R.texels > 0xffffffffull)
      return records_fail(err, "texture pool exceeds 2^32 texels");
  }
  if(d->numTextures == 0)
  {  // a 1x1 white default like src/scene.cpp:513-519
    R.texRecs[0] = TexRec{0, 1, 1, PT_FILTER_LINEAR, PT_WRAP_REPEAT, PT_WRAP_REPEAT, 3, 0};
    R.texels     = 1;
  }
  // ---- material lines, and the interleaved groups their descriptors point into.  The textures a material samples with one (u, v) -- normal, emissive,
  // metallic-roughness, base colour -- are ALSO stored texel by texel next to each other when they share size and sampler (the first present one sets the
  // shape): the 2 x 2 footprints of a shading's taps then share cache lines instead of pulling one or two 128-byte lines per texture for 16 bytes of texels
  // each (k_shade is the kernel next to the read-bandwidth ceiling).  Texel values and filter arithmetic are untouched; the plain copies stay for the any-hit
  // evaluation and the other texture roles.  PT_TUNE texGroups=0: descriptors point at the plain copies.
  R.matLines.assign(size_t(PT_MAT_LINE_QUADS) * std::max<size_t>(1, d->numMaterials), uint4{0u, 0u, 0u, 0u});
  for(uint32_t m = 0; m < d->numMaterials; ++m)
  {
    const pt_GltfShadeMaterial& mt = d->materials[m];
    const int ids[4] = {mt.normalTexture, mt.emissiveTexture, mt.pbrMetallicRoughnessTexture, mt.pbrBaseColorTexture};
    TexRec    rec[4];
    for(int k = 0; k < 4; ++k)
      rec[k] = R.texRecs[ids[k] > -1 ? size_t(ids[k]) : 0];
    SceneRecords::TexGroup g{{-1, -1, -1, -1}, 0, 0u};
    if(texGroups && d->numTextures)
      for(int k = 0; k < 4; ++k)
      {
        if(ids[k] < 0 || std::find(g.tex, g.tex + g.layers, ids[k]) != g.tex + g.layers)
          continue;
        const TexRec &a = R.texRecs[size_t(ids[k])], &b = R.texRecs[size_t(g.layers ? g.tex[0] : ids[k])];
        if(a.w == b.w && a.h == b.h && a.mag == b.mag && a.wrapS == b.wrapS && a.wrapT == b.wrapT)
          g.tex[g.layers++] = ids[k];
      }
    if(g.layers >= 2)
    {
      size_t at = R.groups.size();
      for(size_t q = 0; q < R.groups.size(); ++q)
        if(std::equal(g.tex, g.tex + 4, R.groups[q].tex))
          at = q;
      if(at == R.groups.size())
      {
        // the interleaved copy is an EXTRA on top of the plain copies (which serve the any-hit evaluation and the other texture roles): a group that
        // would take the pool past 2^32 texels is simply not made -- its material reads the plain copies, as with texGroups=0
        const TexRec& sh    = R.texRecs[size_t(g.tex[0])];
        const size_t  start = (R.texels + 31u) & ~size_t(31), after = start + size_t(sh.w) * sh.h * size_t(g.layers);
        if(after > 0xffffffffull)
        {
          mat_line_pack(mt, rec, &R.matLines[size_t(PT_MAT_LINE_QUADS) * m]);
          continue;
        }
        g.offset = uint32_t(start);
        R.texels = after;
        R.groups.push_back(g);
      }
      const SceneRecords::TexGroup& G = R.groups[at];
      for(int k = 0; k < 4; ++k)
      {
        const int* hit = ids[k] < 0 ? G.tex + G.layers : std::find(G.tex, G.tex + G.layers, ids[k]);
        if(hit == G.tex + G.layers)
          continue;  // absent, or of another shape: its plain copy
        rec[k].offset = G.offset;
        rec[k].tiled  = (rec[k].tiled & 1) | ((G.layers - 1) << 8) | (int(hit - G.tex) << 10);
      }
    }
    mat_line_pack(mt, rec, &R.matLines[size_t(PT_MAT_LINE_QUADS) * m]);
  }
  // ---- compact alpha view of every material (what the any-hit evaluation reads)
  R.alphaMats.assign(d->numMaterials, AlphaMat{});
  for(uint32_t m = 0; m < d->numMaterials; ++m)
  {
    const pt_GltfShadeMaterial& mt = d->materials[m];
    AlphaMat&                   a  = R.alphaMats[m];
    std::memset(&a, 0, sizeof(a));
    a.factorA = mt.pbrBaseColorFactor[3];
    a.cutoff  = mt.alphaCutoff;
    a.mode    = mt.alphaMode;
    a.tex     = mt.pbrBaseColorTexture;
    for(int k = 0; k < 8; ++k)
      a.m[k] = mt.uvTransform[k];
    a.mapOffset = ALPHA_NO_MAP;
    if(mt.pbrBaseColorTexture > -1)
    {
      const TexRec& tr = R.texRecs[mt.pbrBaseColorTexture];
      a.texOffset = tr.offset; a.texW = tr.w; a.texH = tr.h; a.texMag = tr.mag; a.texWrap = tr.wrapS | (tr.wrapT << 8) | (tr.pot << 16);
      if(tr.wrapS == PT_WRAP_REPEAT && tr.wrapT == PT_WRAP_REPEAT && tr.pot == 3)
        a.texWrap |= ALPHA_FAST_TAP;
      if(tr.tiled)
        a.texWrap |= ALPHA_TILED;
    }
  }
  build_opacity_maps(d, R.alphaMats, R.alphaMaps);
  return PT_OK;
}

int pt_set_scene(pt_context* c, const pt_SceneDesc* d)
{
  CTX_CHECK(c);
  SceneRecords R;
  {
    std::string msg;
    const int   vrc = build_scene_records(d, R, msg, c->tune.texTile, c->tune.texGroups);
    if(vrc != PT_OK)
      return c->fail(vrc, "%s", msg.c_str());
  }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));
  c->hPrimBound = R.primBound;

  // ---- uploads
  int rc;
  if((rc = upload(c, c->dVertices, d->vertices, sizeof(pt_VertexAttributes) * size_t(d->numVertices))) != PT_OK) return rc;
  if((rc = upload(c, c->dIndices, d->indices, 4 * size_t(d->numIndices))) != PT_OK) return rc;
  c->hInstances = R.inst;
  if((rc = upload_instances(c)) != PT_OK) return rc;
  if((rc = upload(c, c->dMaterials, d->materials, sizeof(pt_GltfShadeMaterial) * size_t(d->numMaterials))) != PT_OK) return rc;
  {
    pt_Light dummy{};  // "cannot be null" (src/scene.cpp:329-330); never read because nbLights == 0 then
    if((rc = upload(c, c->dLights, d->numLights ? d->lights : &dummy, sizeof(pt_Light) * size_t(d->numLights ? d->numLights : 1))) != PT_OK) return rc;
    c->numLights = d->numLights;
  }
  if((rc = dev_alloc(c, c->dTexels, R.texels * 4)) != PT_OK)
  {  // the interleaved groups double the pool: without the memory for them the scene loads as with texGroups=0
    if(R.groups.empty())
      return rc;
    (void)hipGetLastError();
    std::string msg;
    R = SceneRecords();
    if((rc = build_scene_records(d, R, msg, c->tune.texTile, 0)) != PT_OK)
      return c->fail(rc, "%s", msg.c_str());
    if((rc = dev_alloc(c, c->dTexels, R.texels * 4)) != PT_OK)
      return rc;
  }
  if(d->numTextures == 0)
  {
    uint32_t white = 0xffffffffu;
    HIP_TRY(c, hipMemcpy(c->dTexels.p, &white, 4, hipMemcpyHostToDevice));
  }
  {
    std::vector<uint32_t> staged;  // one image at a time in its storage order
    for(uint32_t t = 0; t < d->numTextures; ++t)
    {
      const TexRec& tr = R.texRecs[t];
      const void*   src = d->textures[t].rgba8;
      if(tr.tiled)
      {
        staged.resize(size_t(tr.w) * tr.h);
        store_texture(staged.data(), tr, src);
        src = staged.data();
      }
      HIP_TRY(c, hipMemcpy((uint32_t*)c->dTexels.p + tr.offset, src, size_t(tr.w) * tr.h * 4, hipMemcpyHostToDevice));
    }
  }
  {
    std::vector<uint32_t> staged;  // one interleaved group at a time
    for(const SceneRecords::TexGroup& g : R.groups)
    {
      const TexRec& sh = R.texRecs[size_t(g.tex[0])];
      staged.assign(size_t(sh.w) * sh.h * size_t(g.layers), 0u);
      store_group(staged.data(), g, R.texRecs, d);
      HIP_TRY(c, hipMemcpy((uint32_t*)c->dTexels.p + g.offset, staged.data(), staged.size() * 4, hipMemcpyHostToDevice));
    }
  }
  if((rc = upload(c, c->dTexRecs, R.texRecs.data(), sizeof(TexRec) * R.texRecs.size())) != PT_OK) return rc;
  if((rc = upload(c, c->dMatLines, R.matLines.data(), sizeof(uint4) * R.matLines.size())) != PT_OK) return rc;
  if((rc = upload(c, c->dAlphaMaps, R.alphaMaps.data(), 4 * R.alphaMaps.size())) != PT_OK) return rc;
  if((rc = upload(c, c->dAlphaMats, R.alphaMats.data(), sizeof(AlphaMat) * R.alphaMats.size())) != PT_OK) return rc;
  c->numInstances = d->numNodes;
  c->numTris      = uint32_t(R.triTotal);
  c->qRatioDepths = 0;  // queue-size feedback of the previous scene
  c->haveScene    = true;
  c->haveAccel    = false;
  refresh_scene_ptrs(c);
  return PT_OK;
}

int pt_build_accel(pt_context* c)
{
  CTX_CHECK(c);
  if(!c->haveScene)
    return c->fail(PT_ERR_STATE, "pt_build_accel before pt_set_scene");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));
  c->haveAccel = false;
  if(c->accelMode == PT_ACCEL_TWO_LEVEL)
    return build_two_level(c);
  int rc;
  c->numBlas = c->numTlasNodes = c->numActive = 0;
  c->hMerged.clear();
  c->mergedTris = c->mergedWide = 0;
  c->mergedOnly = false;
  c->hBlasRanges.clear();
  c->nodeCapacity = 0;
  dev_free(c->dCTlas);
  c->numBvhNodes = c->numTris > 1 ? c->numTris - 1 : 1;
  if((rc = dev_alloc(c, c->dTris, sizeof(TriRec) * size_t(c->numTris ? c->numTris : 1))) != PT_OK) return rc;
  if((rc = dev_alloc(c, c->dAlphaRecs, sizeof(AlphaRec) * size_t(c->numTris ? c->numTris : 1))) != PT_OK) return rc;
  if((rc = dev_alloc(c, c->dBvh, sizeof(BvhNode) * size_t(c->numBvhNodes))) != PT_OK) return rc;
  if((rc = dev_alloc(c, c->dWide, sizeof(WideNode) * size_t(c->numBvhNodes))) != PT_OK) return rc;
  auto t0 = std::chrono::steady_clock::now();
  char msg[256];
  // the builder's ~30 temporaries come out of one arena (one allocation and one free instead of thirty each: 3-5 ms of a 15 ms build); whatever
  // does not fit -- or everything, if the arena cannot be had -- is allocated singly
  PtScratch arena;
  {
    const size_t want = size_t(c->numTris) * 640 + (size_t(1) << 20);
    if(hipMalloc((void**)&arena.base, want) == hipSuccess)
      arena.cap = want;
    else
    {
      arena.base = nullptr;
      (void)hipGetLastError();
    }
  }
  const int brc = pt_accel_build(c->stream, c->tune, (const InstanceRec*)c->dInstances.p, c->numInstances, (const float4*)c->dVertices.p, (const uint32_t*)c->dIndices.p, c->numTris,
                                 (TriRec*)c->dTris.p, (AlphaRec*)c->dAlphaRecs.p, (BvhNode*)c->dBvh.p, (WideNode*)c->dWide.p, &c->numWideNodes, msg, sizeof(msg), nullptr, &arena);
  arena.release();
  if(arena.base)
    (void)hipFree(arena.base);
  if(brc != 0)
    return c->fail(PT_ERR_HIP, "pt_build_accel: %s", msg);
  build_cnodes(c, c->numWideNodes);
  build_shade_tris(c, c->numTris);
  HIP_TRY(c, sync_all(c));
  c->msBuild   = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // world bounds of the triangles = union of the root's two child boxes (origin cells of the ray-sort keys)
  for(int k = 0; k < 3; ++k)
  {
    c->scene.boundsMin[k]    = 0.f;
    c->scene.boundsInvExt[k] = 0.f;
  }
  if(c->numTris > 0)
  {
    BvhNode root;
    HIP_TRY(c, hipMemcpy(&root, c->dBvh.p, sizeof(root), hipMemcpyDeviceToHost));
    bounds_from_root(c, root, c->numTris > 1 && root.d.y != BVH_NONE);
  }
  c->haveAccel = true;
  c->warmPending = true;
  refresh_scene_ptrs(c);
  return PT_OK;
}

int pt_set_camera(pt_context* c, const pt_SceneCamera* cam)
{
  CTX_CHECK(c);
  if(!cam)
    return c->fail(PT_ERR_INVALID, "pt_set_camera: null");
  int rc = flush_pending(c);  // frames already handed over keep the camera they were given
  if(rc != PT_OK)
    return rc;
  c->scene.camera = *cam;
  c->haveCamera   = true;
  return PT_OK;
}

int pt_set_sunsky(pt_context* c, const pt_SunAndSky* ss)
{
  CTX_CHECK(c);
  if(!ss)
    return c->fail(PT_ERR_INVALID, "pt_set_sunsky: null");
  int rc = flush_pending(c);
  if(rc != PT_OK)
    return rc;
  c->scene.sunsky = *ss;
  return PT_OK;
}

int pt_set_env(pt_context* c, const float* rgba, int w, int h, float* out_integral, float* out_average)
{
  CTX_CHECK(c);
  if(!rgba || w <= 0 || h <= 0)
    return c->fail(PT_ERR_INVALID, "pt_set_env: bad image");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));
  std::vector<pt_EnvAccel> accel(size_t(w) * h);
  float                    integral = 1.f, average = 1.f;
  if(pt_build_env_accel(rgba, w, h, accel.data(), &integral, &average) != PT_OK)
    return c->fail(PT_ERR_INVALID, "pt_build_env_accel failed");
  int rc;
  if((rc = upload(c, c->dEnv, rgba, sizeof(float) * 4 * size_t(w) * h)) != PT_OK) return rc;
  if((rc = upload(c, c->dEnvAccel, accel.data(), sizeof(pt_EnvAccel) * accel.size())) != PT_OK) return rc;
  c->scene.envW = w;
  c->scene.envH = h;
  c->haveEnv    = true;
  refresh_scene_ptrs(c);
  if(out_integral) *out_integral = integral;
  if(out_average) *out_average = average;
  return PT_OK;
}

int pt_set_variant(pt_context* c, int variant)
{
  CTX_CHECK(c);
  if(variant != PT_VARIANT_RAYQUERY && variant != PT_VARIANT_RTX)
    return c->fail(PT_ERR_INVALID, "pt_set_variant: %d", variant);
  int rc = flush_pending(c);  // frames already handed over keep the variant they were given
  if(rc != PT_OK)
    return rc;
  c->variant = variant;
  return PT_OK;
}

int pt_use_any_hit(pt_context* c, int enable)
{
  CTX_CHECK(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));  // frames already handed over keep the mode they were given
  const bool on = enable != 0;
  if(on == c->anyHit)
    return PT_OK;
  c->anyHit = on;
  if(!c->haveScene)
    return PT_OK;
  int rc = upload_instances(c);
  if(rc != PT_OK)
    return rc;
  refresh_scene_ptrs(c);
  if(c->haveAccel)
  {  // the opaque / non-opaque classification is baked into the triangle records: rebuild, like useAnyHit re-creates the pipeline
    c->haveAccel = false;
    return pt_build_accel(c);
  }
  return PT_OK;
}


int pt_set_accel_mode(pt_context* c, int mode)
{
  CTX_CHECK(c);
  if(mode != PT_ACCEL_FLAT && mode != PT_ACCEL_TWO_LEVEL)
    return c->fail(PT_ERR_INVALID, "pt_set_accel_mode: %d", mode);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));  // frames already handed over keep the structure they were given
  if(mode == c->accelMode)
    return PT_OK;
  c->accelMode = mode;
  if(c->haveAccel)
  {
    c->haveAccel = false;
    return pt_build_accel(c);
  }
  refresh_scene_ptrs(c);
  return PT_OK;
}

int pt_update_instances(pt_context* c, const pt_Node* nodes, uint32_t numNodes)
{
  CTX_CHECK(c);
  if(!c->haveScene)
    return c->fail(PT_ERR_STATE, "pt_update_instances before pt_set_scene");
  if(!nodes || numNodes != c->hInstances.size())
    return c->fail(PT_ERR_INVALID, "pt_update_instances: %u nodes, the scene has %zu", numNodes, c->hInstances.size());
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));  // frames already handed over keep the transforms they were given
  std::vector<InstanceRec> inst = c->hInstances;
  for(uint32_t n = 0; n < numNodes; ++n)
  {
    if(nodes[n].primMesh != inst[n].primMesh)
      return c->fail(PT_ERR_INVALID, "pt_update_instances: node %u changes its primMesh (%d -> %d); only the world matrices may change", n, inst[n].primMesh, nodes[n].primMesh);
    if(!set_instance_transform(inst[n], nodes[n].worldMatrix, inst[n].flags))
      return c->fail(PT_ERR_INVALID, "node %u: singular world matrix", n);
  }
  const std::vector<InstanceRec> before = c->hInstances;
  c->hInstances = inst;
  int rc = upload_instances(c);
  if(rc != PT_OK)
    return rc;
  refresh_scene_ptrs(c);
  if(!c->haveAccel)
    return PT_OK;
  if(c->accelMode == PT_ACCEL_TWO_LEVEL)
  {  // refit: the object-space BLASes are untouched, only the instance boxes and the hierarchy over them are redone; the merged world-space
     // structure is rebuilt when one of its instances moved
    auto t0 = std::chrono::steady_clock::now();
    bool mergedMoved = false;
    for(uint32_t i : c->hMerged)
      mergedMoved = mergedMoved || std::memcmp(&before[i].objectToWorld, &c->hInstances[i].objectToWorld, sizeof(Affine)) != 0;
    if(mergedMoved && (rc = build_merged(c)) != PT_OK)
    {
      c->haveAccel = false;
      return rc;
    }
    if((rc = build_tlas(c)) != PT_OK)
    {
      c->haveAccel = false;
      return rc;
    }
    c->msBuild = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    refresh_scene_ptrs(c);
    return PT_OK;
  }
  c->haveAccel = false;  // flat structure: the world-space triangles are baked in
  return pt_build_accel(c);
}

int pt_set_shard(pt_context* c, int rank, int nranks)
{
  CTX_CHECK(c);
  if(nranks < 1 || rank < 0 || rank >= nranks)
    return c->fail(PT_ERR_INVALID, "pt_set_shard: rank %d of %d", rank, nranks);
  c->rank   = rank;
  c->nranks = nranks;
  c->width = c->height = 0;  // force a re-layout on the next pt_resize
  return PT_OK;
}

// Renderer::create is where the reference builds its pipelines (src/rayquery.cpp:63-92): everything a first frame would otherwise pay for happens
// here, untimed by definition.  Every frame slot's path state and queues are written once (first use of ~45 GB of fresh allocations), and one
// throw-away launch sequence runs on every slot's stream -- code objects of all stage kernels loaded, clocks up, scene / structure / textures
// pulled through the caches once, and the queue-size feedback (where k_tail takes over) seeded with the alive fractions of THIS scene
// instead of the 0.3-per-bounce guess.  Once per acceleration structure (pt_build_accel re-arms it).  Nothing the caller can observe changes: the accumulation image is cleared afterwards (pt_resize
// clears it anyway), the device counters are put back, statistics and frame numbering are untouched.  PT_TUNE warm=0 skips it.
static int warm_slots(pt_context* c)
{
  if(!c->tune.warm || !c->warmPending || !c->haveScene || !c->haveAccel || !c->haveCamera || !(c->haveEnv || c->scene.sunsky.in_use == 1) || c->numSlots == 0)
    return PT_OK;
  c->warmPending = false;  // once per scene: the de-scaling resizes of an interactive session (sample_example.cpp:410-413) must not stall on it
  if(c->scene.camera.nbLights < 0 || uint32_t(c->scene.camera.nbLights) > c->numLights)
    return PT_OK;  // pt_render_frame reports it
  Counters saved;
  HIP_TRY(c, hipMemcpy(&saved, c->dCounters.p, sizeof(Counters), hipMemcpyDeviceToHost));
  FrameParams fp{};
  fp.st.frame = 0; fp.st.maxDepth = 10; fp.st.maxSamples = 1; fp.st.fireflyClampThreshold = 1.0f; fp.st.hdrMultiplier = 1.0f;  // sample_example.hpp:162-174
  fp.st.debugging_mode = PT_DEBUG_NONE; fp.st.pbrMode = 0; fp.st.size[0] = c->width; fp.st.size[1] = c->height;
  fp.width = c->width; fp.height = c->height; fp.tilesX = c->tilesX; fp.tilesY = c->tilesY; fp.rank = c->rank; fp.nranks = c->nranks;
  fp.numLocalTiles = c->numLocalTiles; fp.numSlots = c->numSlots; fp.variant = c->variant; fp.sample = 0;
  fp.batch = uint32_t(std::min(c->batchMax, 8));
  StageTimers off;  // disabled: the warm-up never shows in the stage timings
  const int tailFrom = tail_from_depth(double(fp.batch) * double(c->numSlots), fp.st.maxDepth, c->tune.tailBelow, c->qRatio, 0);
  for(int k = 0; k < slot_total(c); ++k)
  {
    pt_context::FrameSlot& fs = slot_at(c, k);
    for(DevBuf& bf : fs.dState)
      (void)hipMemsetAsync(bf.p, 0, bf.bytes, fs.stream);
    DevBuf* q[] = {&fs.dQueueA, &fs.dQueueB, &fs.dQueueS, &fs.dQueueX, &fs.dQueueX2, &fs.dQueueR};
    for(DevBuf* bf : q)
      (void)hipMemsetAsync(bf->p, 0, bf->bytes, fs.stream);
  }
  for(int k = 0; k < slot_total(c); ++k)
  {
    FrameParams fk = fp;
    int         tk = tailFrom;
    if(k >= c->inflight)
    {  // a display slot holds one frame
      fk.batch = 1;
      tk       = tail_from_depth(double(c->numSlots), fp.st.maxDepth, c->tune.tailBelow, c->qRatio, 0);
    }
    pt_launch_frame(slot_at(c, k).stream, c->tune, c->scene, slot_at(c, k).rb, fk, &off, nullptr, nullptr, tk);
  }
  pt_context::FrameSlot& f0 = c->slots[0];
  const int nd = std::min(std::min(tailFrom + 1, int(fp.st.maxDepth)), PT_MAX_DEPTH);
  if(f0.hCounts)
    (void)hipMemcpyAsync(f0.hCounts, f0.rb.countsDone, sizeof(uint32_t) * CNT_STRIDE * size_t(nd), hipMemcpyDeviceToHost, f0.stream);
  hipError_t werr = hipSuccess;
  for(int k = 0; k < slot_total(c); ++k)
  {
    const hipError_t e = hipStreamSynchronize(slot_at(c, k).stream);
    werr               = werr == hipSuccess ? e : werr;
  }
  // whatever happened: the device counters and the accumulation image go back to what the caller left (best effort), and a failed warm-up is retried
  // by the next pt_resize instead of being silently skipped for the life of the scene
  const hipError_t r1 = hipMemcpy(c->dCounters.p, &saved, sizeof(Counters), hipMemcpyHostToDevice);
  const hipError_t r2 = hipMemset(c->dFrame.p, 0, c->dFrame.bytes);
  if(werr != hipSuccess || r1 != hipSuccess || r2 != hipSuccess)
  {
    c->warmPending = true;
    HIP_TRY(c, werr != hipSuccess ? werr : (r1 != hipSuccess ? r1 : r2));
  }
  if(f0.hCounts && c->qRatioDepths == 0)
  {
    const double paths = double(fp.batch) * double(c->numSlots);
    for(int d = 0; d < nd; ++d)
      c->qRatio[d] = double(f0.hCounts[size_t(d) * CNT_STRIDE + CNT_IN]) / paths;
    c->qRatioDepths = nd;
  }
  return PT_OK;
}

int pt_resize(pt_context* c, int width, int height)
{
  CTX_CHECK(c);
  if(width <= 0 || height <= 0)
    return c->fail(PT_ERR_INVALID, "pt_resize: %dx%d", width, height);
  if(width == c->width && height == c->height)
    return PT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));
  c->tilesX = (width + PT_TILE - 1) / PT_TILE;
  c->tilesY = (height + PT_TILE - 1) / PT_TILE;
  std::vector<uint32_t> local;
  std::vector<uint32_t> perRank(c->nranks, 0);
  uint64_t              localPixels = 0;
  for(int ty = 0; ty < c->tilesY; ++ty)
    for(int tx = 0; tx < c->tilesX; ++tx)
    {
      int r = (tx + ty) % c->nranks;
      perRank[r]++;
      if(r == c->rank)
      {
        local.push_back(uint32_t(ty * c->tilesX + tx));
        localPixels += uint64_t(std::min(PT_TILE, width - tx * PT_TILE)) * uint64_t(std::min(PT_TILE, height - ty * PT_TILE));
      }
    }
  c->localPixels = localPixels;
  c->numLocalTiles   = uint32_t(local.size());
  c->maxTilesPerRank = 0;
  for(uint32_t v : perRank)
    c->maxTilesPerRank = v > c->maxTilesPerRank ? v : c->maxTilesPerRank;
  c->numSlots = c->numLocalTiles * 1024u;

  int rc;
  // frames per batch: the tuning value, bounded so that one frame slot's path state stays below 2^26 paths (~11 GB):
  // 32 frames of a full 1080p image, 64 of an 8-GPU shard (measured best for both, profiles/r01_scaling_estimate.txt)
  c->batchMax = std::max(1, std::min(c->tune.batch, int((1u << 26) / (c->numSlots ? c->numSlots : 1u))));
  // In-flight path state: 9 float4 arrays + 9 index queues per path slot, times the batch, times the frame slots -- 38 GB for a 1080p image at
  // the defaults, sized for 288 GB of HBM.  It is a budget, not a requirement: PT_TUNE stateGB=<n> (or what hipMemGetInfo reports as free,
  // minus a reserve) caps it, and an allocation that still fails halves the batch / drops frame slots and retries, down to one frame on
  // one slot, before PT_ERR_OOM is reported.  Smaller batches only cost throughput, never results.
  const size_t perPath = 9 * sizeof(float4) + 6 * sizeof(uint32_t);
  c->inflight          = c->inflightMax;
  c->displaySlots      = c->displaySlotsMax;
  // (Round 6 tried a policy by shard size here -- two frame slots for a shard of at most 300 k pixels, what PT_TUNE inflight=2 showed on a 1/8 shard of a
  // 1080p image at 20 steps: +5 % there, but -9 % at the configuration's own 256 steps, where four full batches want four slots: 89 % -> 81 % predicted
  // efficiency at 8 GPUs (profiles/r06_shard_policy_experiment.txt).  Not adopted: the slot count stays what the context was created with.)
  {
    // what is free now PLUS what the frame slots already hold (those buffers are re-used or released below): a repeated pt_resize at the same
    // size must arrive at the same batch, not at half of it.  A failed query means "no cap" -- the retry loop below still shrinks on a failed allocation.
    size_t freeB = 0, totalB = 0, held = 0;
    const bool haveInfo = hipMemGetInfo(&freeB, &totalB) == hipSuccess;
    (void)hipGetLastError();
    for(int i = 0; i < PT_MAX_INFLIGHT; ++i)
    {
      pt_context::FrameSlot& fs = c->slots[i];
      for(DevBuf& bf : fs.dState) held += bf.bytes;
      const DevBuf* q[] = {&fs.dQueueA, &fs.dQueueB, &fs.dQueueS, &fs.dQueueX, &fs.dQueueX2, &fs.dQueueR};
      for(const DevBuf* bf : q) held += bf->bytes;
    }
    const double budget = c->tune.stateMB > 0 ? c->tune.stateMB * 1e6 : c->tune.stateGB > 0 ? c->tune.stateGB * 1e9 : (haveInfo ? (double(freeB) + double(held)) * 0.85 : 1e30);
    auto need = [&]() { return double(perPath) * double(c->numSlots ? c->numSlots : 1) * (double(c->batchMax) * c->inflight + c->displaySlots); };
    if(need() > budget)
      c->displaySlots = 0;  // the one-frame display slots go first
    while(c->batchMax > 1 && need() > budget)
      c->batchMax = (c->batchMax + 1) / 2;
    while(c->inflight > 1 && need() > budget)
      c->inflight = (c->inflight + 1) / 2;
    if((c->tune.stateMB > 0 || c->tune.stateGB > 0) && need() > budget)  // an explicit cap that one frame on one slot exceeds (a cap derived
      return c->fail(PT_ERR_OOM, "pt_resize: the path state of one %dx%d frame (%.0f bytes) exceeds the configured budget (%.0f bytes)", width, height, need(), budget);  // from the free memory is left to the allocations below)
  }
  for(;;)
  {
    bool         ok = true;
    for(int k = 0; k < slot_total(c) && ok; ++k)
    {
      pt_context::FrameSlot& fs = slot_at(c, k);
      const size_t           n  = size_t(c->numSlots ? c->numSlots : 1) * size_t(k < c->inflight ? c->batchMax : 1);
      for(DevBuf& bf : fs.dState)
        ok = ok && dev_alloc_quiet(bf, sizeof(float4) * n);
      DevBuf* q[] = {&fs.dQueueA, &fs.dQueueB, &fs.dQueueS, &fs.dQueueX, &fs.dQueueX2, &fs.dQueueR};
      for(DevBuf* bf : q)
        ok = ok && dev_alloc_quiet(*bf, 4 * n);
      ok = ok && dev_alloc_quiet(fs.dCounts, sizeof(uint32_t) * CNT_STRIDE * (PT_MAX_DEPTH + 2));
      ok = ok && dev_alloc_quiet(fs.dCountsDone, sizeof(uint32_t) * CNT_STRIDE * (PT_MAX_DEPTH + 2));
      if(ok)
      {  // a sample pass starts on a cleared counter block: cleared here once, then by every k_accumulate
        HIP_TRY(c, hipMemset(fs.dCounts.p, 0, fs.dCounts.bytes));
        HIP_TRY(c, hipMemset(fs.dCountsDone.p, 0, fs.dCountsDone.bytes));
      }
    }
    if(ok)
    {  // slots that dropped out (a smaller budget than at the last pt_resize) give their buffers back
      for(int i = 0; i < PT_MAX_INFLIGHT; ++i)
      {
        const bool used = i < c->inflight || (i >= c->inflightMax && i < c->inflightMax + c->displaySlots);
        if(used)
          continue;
        pt_context::FrameSlot& fs = c->slots[i];
        for(DevBuf& bf : fs.dState) dev_free(bf);
        DevBuf* q[] = {&fs.dQueueA, &fs.dQueueB, &fs.dQueueS, &fs.dQueueX, &fs.dQueueX2, &fs.dQueueR};
        for(DevBuf* bf : q) dev_free(*bf);
      }
      break;
    }
    (void)hipGetLastError();
    for(int i = 0; i < PT_MAX_INFLIGHT; ++i)
    {  // release everything before retrying smaller
      pt_context::FrameSlot& fs = c->slots[i];
      for(DevBuf& bf : fs.dState) dev_free(bf);
      DevBuf* q[] = {&fs.dQueueA, &fs.dQueueB, &fs.dQueueS, &fs.dQueueX, &fs.dQueueX2, &fs.dQueueR};
      for(DevBuf* bf : q) dev_free(*bf);
    }
    if(c->displaySlots > 0)
      c->displaySlots = 0;
    else if(c->batchMax > 1)
      c->batchMax = (c->batchMax + 1) / 2;
    else if(c->inflight > 1)
      c->inflight = (c->inflight + 1) / 2;
    else
      return c->fail(PT_ERR_OOM, "pt_resize: not enough device memory for the path state of one %dx%d frame (%zu bytes per path)", width, height, perPath);
  }
  if((rc = dev_alloc(c, c->dFrame, sizeof(float4) * size_t(c->maxTilesPerRank ? c->maxTilesPerRank : 1) * 1024u)) != PT_OK) return rc;
  if((rc = upload(c, c->dSlotTile, local.data(), 4 * local.size())) != PT_OK) return rc;
  HIP_TRY(c, hipMemset(c->dFrame.p, 0, c->dFrame.bytes));
  if((rc = dev_alloc(c, c->dRowMajor, sizeof(float4) * size_t(width) * height)) != PT_OK) return rc;
  HIP_TRY(c, hipMemset(c->dRowMajor.p, 0, sizeof(float4) * size_t(width) * height));
  if((rc = dev_alloc(c, c->dRgba8, 4 * size_t(width) * height)) != PT_OK) return rc;
  if((rc = dev_alloc(c, c->dMean, 3 * sizeof(double))) != PT_OK) return rc;

  c->width    = width;
  c->height   = height;
  c->haveFull = false;
  for(int k = 0; k < slot_total(c); ++k)
  {
    pt_context::FrameSlot& fs = slot_at(c, k);
    PathState&             ps = fs.rb.ps;
    ps.rayO.p = (float4*)fs.dState[0].p; ps.rayD.p = (float4*)fs.dState[1].p; ps.thr.p = (float4*)fs.dState[2].p; ps.rad.p = (float4*)fs.dState[3].p;
    ps.absorb.p = (float4*)fs.dState[4].p; ps.neeDir.p = (float4*)fs.dState[5].p; ps.neeRad.p = (float4*)fs.dState[6].p; ps.hit.p = (float4*)fs.dState[7].p;
    ps.sum.p = (float4*)fs.dState[8].p;
    fs.rb.queueA   = (uint32_t*)fs.dQueueA.p;
    fs.rb.queueB   = (uint32_t*)fs.dQueueB.p;
    fs.rb.queueS   = (uint32_t*)fs.dQueueS.p;
    fs.rb.queueX   = (uint32_t*)fs.dQueueX.p;
    fs.rb.queueX2  = (uint32_t*)fs.dQueueX2.p;
    fs.rb.queueR   = (uint32_t*)fs.dQueueR.p;
    fs.rb.counts   = (uint32_t*)fs.dCounts.p;
    fs.rb.countsDone = (uint32_t*)fs.dCountsDone.p;
    fs.rb.frame    = (float4*)c->dFrame.p;
    fs.rb.slotTile = (uint32_t*)c->dSlotTile.p;
    fs.rb.counters = (Counters*)c->dCounters.p;
  }
  return warm_slots(c);
}

int pt_render_frame(pt_context* c, const pt_RtxState* st)
{
  CTX_CHECK(c);
  if(!st)
    return c->fail(PT_ERR_INVALID, "pt_render_frame: null state");
  if(!c->haveScene || !c->haveAccel)
    return c->fail(PT_ERR_STATE, "pt_render_frame before pt_set_scene / pt_build_accel");
  if(!c->haveEnv && c->scene.sunsky.in_use != 1)
    return c->fail(PT_ERR_STATE, "pt_render_frame without an environment (pt_set_env) or sun & sky");
  if(c->width == 0)
    return c->fail(PT_ERR_STATE, "pt_render_frame before pt_resize");
  if(st->size[0] != c->width || st->size[1] != c->height)
    return c->fail(PT_ERR_INVALID, "RtxState.size %dx%d != pt_resize %dx%d", st->size[0], st->size[1], c->width, c->height);
  if(st->maxSamples < 1 || st->maxDepth < 0 || st->maxDepth > PT_MAX_DEPTH || st->frame < 0)
    return c->fail(PT_ERR_INVALID, "RtxState: maxSamples %d maxDepth %d (limit %d) frame %d", st->maxSamples, st->maxDepth, PT_MAX_DEPTH, st->frame);
  // the lights buffer holds numLights records (pt_set_scene); the shader indexes it with camera.nbLights (pathtrace.glsl:120-121), and the two
  // arrive through independent calls
  if(c->scene.camera.nbLights < 0 || uint32_t(c->scene.camera.nbLights) > c->numLights)
    return c->fail(PT_ERR_INVALID, "camera.nbLights = %d but the scene holds %u lights", c->scene.camera.nbLights, c->numLights);
  HIP_TRY(c, hipSetDevice(c->device));
  if(c->numSlots == 0)
    return PT_OK;
  pt_RtxState a = *st, b = c->pendState;
  a.frame = b.frame = 0;
  const bool joins = c->pendCount > 0 && c->pendCount < c->batchMax && std::memcmp(&a, &b, sizeof(a)) == 0 && st->frame == c->pendState.frame + c->pendCount;
  int rc;
  if(!joins && (rc = flush_pending(c)) != PT_OK)
    return rc;
  if(c->pendCount == 0)
    c->pendState = *st;
  c->pendCount++;
  c->haveFull = false;
  c->stats.samples += uint64_t(st->maxSamples) * c->localPixels;
  if(c->pendCount >= c->batchMax)
    return flush_pending(c);
  return PT_OK;
}

// Launches the pending batch of frames on the next frame slot's stream.
}  // extern "C"
int flush_pending(pt_context* c)
{
  if(c->pendCount == 0)
    return PT_OK;
  if(c->countsDirty)
  {  // The counter block of a frame slot is zero when a pass starts because the PREVIOUS pass's k_accumulate left it so (no fill kernel per sequence).  After a
     // HIP error a sequence may have stopped short of that: stale queue sizes would make the next pass append past its queues.  Clear every slot's block.
    (void)sync_all(c);
    (void)hipGetLastError();
    for(int i = 0; i < PT_MAX_INFLIGHT; ++i)
      if(c->slots[i].dCounts.p)
        HIP_TRY(c, hipMemset(c->slots[i].dCounts.p, 0, c->slots[i].dCounts.bytes));
    c->countsDirty = false;
  }
  FrameParams fp{};
  fp.st            = c->pendState;
  fp.width         = c->width;
  fp.height        = c->height;
  fp.tilesX        = c->tilesX;
  fp.tilesY        = c->tilesY;
  fp.rank          = c->rank;
  fp.nranks        = c->nranks;
  fp.numLocalTiles = c->numLocalTiles;
  fp.numSlots      = c->numSlots;
  fp.variant       = c->variant;
  fp.sample        = 0;
  // A batch is cut into as many pieces as there are idle frame slots (separate streams), so that a short run of frames -- or the first
  // batch of a long one -- has several launch sequences overlapping instead of one chain of dependent kernels.  In the steady state of a
  // long run every slot is busy and a full batch goes out as one sequence.
  int       parts = 1;
  const int total = c->pendCount;
  if(total >= 4)
  {
    int busy = 0;  // launch sequences still running (their accumulate has not completed)
    for(int i = 0; i < c->inflight; ++i)
      if(c->slots[i].launched && hipEventQuery(c->slots[i].accumDone) == hipErrorNotReady)
        ++busy;
      else
        c->slots[i].launched = false;
    (void)hipGetLastError();  // hipErrorNotReady is not an error
    const int freeSlots = c->inflight - busy;
    // a partial flush (the caller is waiting) is cut fine; a full batch only when the GPU is idle (the first batch of a run) -- later ones
    // find busy slots and go out whole, so the steady state of a long run works on full batches
    const int minPart = total < c->batchMax ? 2 : total;  // a full batch is never split (2-7 % slower at 96-256 frames, profiles/README.md)
    parts = std::max(1, std::min(freeSlots, total / minPart));
  }
  // A launch of ONE frame while nothing else runs (a display loop that waits for every image) is cut the other way: into bands of the frame's
  // tiles, one launch sequence per idle slot.  A band addresses its part of the tile list, of the accumulation image and of nothing else, so it
  // is an ordinary launch with shifted base pointers; the bands' late, thin bounces overlap each other's full ones (profiles/r04z_*).
  int bands = 1;
  if(total == 1 && c->tune.bands > 1 && !c->timers.enabled)
  {
    int idle = 0;
    for(int k = 0; k < slot_total(c); ++k)
    {
      pt_context::FrameSlot& fs = slot_at(c, k);
      if(fs.launched && hipEventQuery(fs.accumDone) == hipErrorNotReady)
        continue;
      fs.launched = false;
      ++idle;
    }
    (void)hipGetLastError();
    if(idle == slot_total(c))  // with frames in flight the slots are the pipeline: one sequence per frame
      bands = std::max(1, std::min(std::min(idle, c->tune.bands), int(c->numLocalTiles / uint32_t(c->tune.bandTiles))));
  }
  c->pendCount          = 0;
  c->renderedSinceCheck = true;
  // queue-size feedback: take the counters of the newest launch sequence that has finished
  for(int i = 0; i < PT_MAX_INFLIGHT; ++i)
  {
    pt_context::FrameSlot& fs = c->slots[i];
    if(fs.countsDone && fs.countsSeq > c->qRatioSeq && fs.countsPaths > 0 && hipEventQuery(fs.countsDone) == hipSuccess)
    {
      const int nd = std::min(fs.countsDepths, PT_MAX_DEPTH);
      for(int d = 0; d < nd; ++d)
        c->qRatio[d] = double(fs.hCounts[size_t(d) * CNT_STRIDE + CNT_IN]) / double(fs.countsPaths);
      c->qRatioDepths = nd;
      c->qRatioSeq    = fs.countsSeq;
    }
  }
  (void)hipGetLastError();  // hipErrorNotReady is not an error
  int done              = 0;
  // interleaved submission needs every piece to have exactly one accumulate step, and the stage timers record their events in launch order
  const bool                       interleave = (parts > 1 || bands > 1) && fp.st.maxSamples == 1 && !c->timers.enabled;
  std::vector<std::vector<PtStep>> plans;
  std::vector<pt_context::FrameSlot*> planSlot;
  std::vector<int>                 planTail;
  std::vector<uint32_t>            planPaths;
  plans.reserve(size_t(std::max(parts, bands)));
  for(int b = 0; b < bands && bands > 1; ++b)
  {
    const uint32_t t0 = uint32_t(uint64_t(c->numLocalTiles) * uint64_t(b) / uint64_t(bands)), t1 = uint32_t(uint64_t(c->numLocalTiles) * uint64_t(b + 1) / uint64_t(bands));
    FrameParams    fb = fp;
    fb.st.frame       = c->pendState.frame;
    fb.batch          = 1;
    fb.numLocalTiles  = t1 - t0;
    fb.numSlots       = (t1 - t0) * 1024u;
    const int tailFrom = tail_from_depth(double(fb.numSlots), fp.st.maxDepth, c->tune.tailBelow, c->qRatio, c->qRatioDepths);
    pt_context::FrameSlot& fs = slot_at(c, int(c->displayCounter++ % uint64_t(slot_total(c))));
    RenderBuffers  rbb = fs.rb;
    rbb.slotTile += t0;
    rbb.frame += size_t(t0) * 1024u;
    plans.emplace_back();
    planSlot.push_back(&fs);
    pt_plan_frame(plans.back(), fs.stream, c->tune, c->scene, rbb, fb, &c->timers, c->lastAccum, fs.accumDone, tailFrom);
    c->lastAccum = fs.accumDone;
    fs.launched  = true;
    if(!interleave)
    {
      for(PtStep& st : plans.back())
        st.fn();
      plans.back().clear();
    }
    planTail.push_back(tailFrom);
    planPaths.push_back(fb.numSlots);
  }
  for(int p = 0; p < parts && bands == 1; ++p)
  {
    // (equal pieces: sizes falling 4 : 3 : 2 : 1, meant to let the streams drift apart so that trace and shade stages overlap, measured 5 % slower,
    // eight pieces on eight slots 20 % slower, profiles/r04d_*)
    const int n = (total - done) / (parts - p);
    fp.st.frame = c->pendState.frame + done;
    fp.batch    = uint32_t(n);
    done += n;
    // where k_tail takes over: the first bounce whose queue is expected to hold <= tailBelow paths.  Expectation = this launch's paths x the
    // alive fraction observed at that bounce; bounces beyond the observed ones continue the last observed shrink factor; before anything was
    // observed a shrink of 0.3 per bounce is assumed.  A wrong guess costs time, never results.
    const int tailFrom = tail_from_depth(double(n) * double(c->numSlots), fp.st.maxDepth, c->tune.tailBelow, c->qRatio, c->qRatioDepths);
    // a launch of ONE frame (the display loop flushes per frame) rotates over the batch slots and the display slots, a batch over the batch slots
    pt_context::FrameSlot& fs = (total == 1 && c->displaySlots > 0) ? slot_at(c, int(c->displayCounter++ % uint64_t(slot_total(c)))) : c->slots[c->frameCounter++ % uint64_t(c->inflight)];
    plans.emplace_back();
    planSlot.push_back(&fs);
    pt_plan_frame(plans.back(), fs.stream, c->tune, c->scene, fs.rb, fp, &c->timers, c->lastAccum, fs.accumDone, tailFrom);
    c->lastAccum = fs.accumDone;
    fs.launched  = true;
    if(!interleave)
    {  // one sequence after the other
      for(PtStep& st : plans.back())
        st.fn();
      plans.back().clear();
    }
    planTail.push_back(tailFrom);
    planPaths.push_back(uint32_t(n) * c->numSlots);
  }
  if(interleave)
  {  // stage by stage in turn over the pieces: every stream gets its first kernels at once, the host stays ahead of all of them; a piece's
     // accumulate step (it waits on the previous piece's event) is never issued before the previous piece's
    std::vector<size_t> at(plans.size(), 0);
    std::vector<char>   accumIssued(plans.size(), 0);
    for(bool more = true; more;)
    {
      more = false;
      for(size_t q = 0; q < plans.size(); ++q)
      {
        if(at[q] >= plans[q].size())
          continue;
        PtStep& st = plans[q][at[q]];
        if(st.accum && q > 0 && !accumIssued[q - 1])
        {
          more = true;
          continue;
        }
        st.fn();
        if(st.accum)
          accumIssued[q] = 1;
        ++at[q];
        more = more || at[q] < plans[q].size();
      }
    }
  }
  for(size_t q = 0; q < planSlot.size(); ++q)
  {
    pt_context::FrameSlot& fs       = *planSlot[q];
    const int              tailFrom = planTail[q];
    const uint32_t         n        = planPaths[q];
    if(fs.hCounts && fp.st.debugging_mode != PT_DEBUG_HEATMAP)
    {
      fs.countsDepths = std::min(std::min(tailFrom + 1, int(fp.st.maxDepth)), PT_MAX_DEPTH);  // the bounce k_tail starts at still has its input count
      fs.countsPaths  = n;
      if(hipMemcpyAsync(fs.hCounts, fs.rb.countsDone, sizeof(uint32_t) * CNT_STRIDE * size_t(fs.countsDepths), hipMemcpyDeviceToHost, fs.stream) == hipSuccess &&
         hipEventRecord(fs.countsDone, fs.stream) == hipSuccess)
        fs.countsSeq = ++c->launchSeq;
      else
        fs.countsSeq = 0;
    }
  }
  HIP_TRY(c, hipGetLastError());
  return PT_OK;
}
extern "C" {

int pt_synchronize(pt_context* c)
{
  CTX_CHECK(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));
  return check_traversal(c);
}

static int untile_to_rowmajor(pt_context* c)
{
  int rc = flush_pending(c);
  if(rc != PT_OK)
    return rc;
  // frames run on their own streams; the last accumulate (itself ordered after all earlier ones) gates the readback
  if(c->lastAccum)
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->lastAccum, 0));
  if(c->haveFull)
    pt_launch_untile(c->stream, (const float4*)c->dFullTiles.p, (const uint32_t*)c->dFullSlotTile.p, uint32_t(c->tilesX) * c->tilesY, c->tilesX, c->width, c->height,
                     (float4*)c->dRowMajor.p);
  else
    pt_launch_untile(c->stream, (const float4*)c->dFrame.p, (const uint32_t*)c->dSlotTile.p, c->numLocalTiles, c->tilesX, c->width, c->height, (float4*)c->dRowMajor.p);
  HIP_TRY(c, hipGetLastError());
  return PT_OK;
}

int pt_read_accum(pt_context* c, float* out)
{
  CTX_CHECK(c);
  if(!out)
    return c->fail(PT_ERR_INVALID, "pt_read_accum: null");
  if(c->width == 0)
    return c->fail(PT_ERR_STATE, "pt_read_accum before pt_resize");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = untile_to_rowmajor(c);
  if(rc != PT_OK)
    return rc;
  HIP_TRY(c, hipMemcpyAsync(out, c->dRowMajor.p, sizeof(float4) * size_t(c->width) * c->height, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, sync_all(c));
  return check_traversal(c);
}

int pt_write_accum(pt_context* c, const float* in)
{
  CTX_CHECK(c);
  if(!in)
    return c->fail(PT_ERR_INVALID, "pt_write_accum: null");
  if(c->width == 0)
    return c->fail(PT_ERR_STATE, "pt_write_accum before pt_resize");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));  // launches what is pending and waits: the restored image replaces everything rendered so far
  HIP_TRY(c, hipMemcpyAsync(c->dRowMajor.p, in, sizeof(float4) * size_t(c->width) * c->height, hipMemcpyHostToDevice, c->stream));
  pt_launch_retile(c->stream, (const float4*)c->dRowMajor.p, (const uint32_t*)c->dSlotTile.p, c->numLocalTiles, c->tilesX, c->width, c->height, (float4*)c->dFrame.p);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, sync_all(c));
  c->haveFull = false;
  return PT_OK;
}

int pt_pick(pt_context* c, float pick_x, float pick_y, const float* view_inverse, const float* proj_inverse, pt_PickResult* out)
{
  CTX_CHECK(c);
  if(!view_inverse || !proj_inverse || !out)
    return c->fail(PT_ERR_INVALID, "pt_pick: null");
  if(!c->haveScene || !c->haveAccel)
    return c->fail(PT_ERR_STATE, "pt_pick before pt_set_scene / pt_build_accel");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if((rc = dev_alloc(c, c->dPick, sizeof(pt_PickResult))) != PT_OK)
    return rc;
  pt_launch_pick(c->stream, c->scene, pick_x, pick_y, view_inverse, proj_inverse, (pt_PickResult*)c->dPick.p, (Counters*)c->dCounters.p);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, c->dPick.p, sizeof(pt_PickResult), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PT_OK;
}

// ---- on-box calibration of the two rooflines bench.py prices against (no reference counterpart) ------------------------------------------
// VALU issue: every lane runs 8 independent v_fmac_f32 chains (the form with the highest measured issue rate, tools/valu_peak.hip; inline asm: the
// compiler can neither pack two of them into v_pk_fma_f32 nor drop them), 8 waves per SIMD on every CU; the result is wave-instructions per second over the whole chip.
__global__ void __launch_bounds__(256) k_calib_valu(int iters, float* out)
{
  float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 0.999f, c = 0.001f;
  for(int i = 0; i < iters; ++i)
  {
    asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                 "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(m), "v"(c));
  }
  float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if(s == 12345.678f)
    out[0] = s;
}
// HBM streaming: float4 copy (read + write) and float4 read-only reduction over buffers far larger than the 256 MB Infinity Cache
__global__ void __launch_bounds__(256) k_calib_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
  for(size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
    dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_calib_read(const float4* __restrict__ src, size_t n, float* out)
{
  float acc = 0.f;
  for(size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
  {
    float4 v = src[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
  if(acc == 12345.678f)
    out[0] = acc;
}
int pt_measure_peaks(pt_context* c, pt_Peaks* out)
{
  CTX_CHECK(c);
  if(!out)
    return c->fail(PT_ERR_INVALID, "pt_measure_peaks: null");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));
  hipDeviceProp_t prop;
  HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
  const int    cus = prop.multiProcessorCount;
  const size_t n   = size_t(1) << 26;  // 2^26 float4 = 1 GiB per buffer
  float4 *     a = nullptr, *b = nullptr;
  float*       sink = nullptr;
  hipEvent_t   e0 = nullptr, e1 = nullptr;
  auto         done = [&](int rc) {
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
    if(e0) (void)hipEventDestroy(e0);
    if(e1) (void)hipEventDestroy(e1);
    return rc;
  };
  if(hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess)
    return done(c->fail(PT_ERR_OOM, "pt_measure_peaks: out of device memory"));
  if(hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipMemsetAsync(a, 0, n * 16, c->stream) != hipSuccess)
    return done(c->fail(PT_ERR_HIP, "pt_measure_peaks: setup failed"));
  auto timed = [&](auto&& launch, int reps) -> double {
    launch();  // warm-up
    (void)hipEventRecord(e0, c->stream);
    for(int i = 0; i < reps; ++i)
      launch();
    (void)hipEventRecord(e1, c->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return double(ms) * 1e-3 / reps;
  };
  const int      iters = 4096;
  const unsigned blocks = unsigned(cus) * 8u;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  double         t = timed([&] { k_calib_valu<<<blocks, 256, 0, c->stream>>>(iters, sink); }, 5);
  out->valuWaveInstrPerSec = double(blocks) * 4.0 * double(iters) * 8.0 / t;
  t = timed([&] { k_calib_copy<<<unsigned(cus) * 16u, 256, 0, c->stream>>>(a, b, n); }, 5);
  out->hbmCopyBytesPerSec = 2.0 * double(n) * 16.0 / t;
  t = timed([&] { k_calib_read<<<unsigned(cus) * 16u, 256, 0, c->stream>>>(a, n, sink); }, 5);
  out->hbmReadBytesPerSec = double(n) * 16.0 / t;
  out->computeUnits = cus;
  out->clockMHz     = prop.clockRate / 1000;
  if(hipGetLastError() != hipSuccess)
    return done(c->fail(PT_ERR_HIP, "pt_measure_peaks: kernel failed"));
  return done(PT_OK);
}

// The fp32 transcendental contract evaluated on the device (include/pt_fpmath.h); tests hold it bit for bit to the host evaluation.
__global__ void k_fpmath(int fn, uint64_t n, const float* a, const float* b, float* out)
{
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  float x = a[i], y = b ? b[i] : 0.0f, r;
  switch(fn)
  {
    case PT_FN_SIN: r = pt_sin(x); break;
    case PT_FN_COS: r = pt_cos(x); break;
    case PT_FN_TAN: r = pt_tan(x); break;
    case PT_FN_ASIN: r = pt_asin(x); break;
    case PT_FN_ACOS: r = pt_acos(x); break;
    case PT_FN_ATAN2: r = pt_atan2(x, y); break;
    case PT_FN_EXP: r = pt_exp(x); break;
    case PT_FN_LOG: r = pt_log(x); break;
    default: r = pt_pow(x, y); break;
  }
  out[i] = r;
}
int pt_fpmath_eval(pt_context* c, int fn, uint64_t n, const float* a, const float* b, float* out)
{
  CTX_CHECK(c);
  if(fn < PT_FN_SIN || fn > PT_FN_POW || !a || !out || ((fn == PT_FN_ATAN2 || fn == PT_FN_POW) && !b))
    return c->fail(PT_ERR_INVALID, "pt_fpmath_eval: bad arguments");
  if(n == 0)
    return PT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  float *dA = nullptr, *dB = nullptr, *dO = nullptr;
  int    rc = PT_OK;
  auto   done = [&](int r) {
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dO);
    return r;
  };
  if(hipMalloc(&dA, n * 4) != hipSuccess || hipMalloc(&dO, n * 4) != hipSuccess || (b && hipMalloc(&dB, n * 4) != hipSuccess))
    return done(c->fail(PT_ERR_OOM, "pt_fpmath_eval: out of device memory"));
  if(hipMemcpy(dA, a, n * 4, hipMemcpyHostToDevice) != hipSuccess || (b && hipMemcpy(dB, b, n * 4, hipMemcpyHostToDevice) != hipSuccess))
    return done(c->fail(PT_ERR_HIP, "pt_fpmath_eval: upload failed"));
  k_fpmath<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>(fn, n, dA, dB, dO);
  if(hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, dO, n * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return done(c->fail(PT_ERR_HIP, "pt_fpmath_eval: kernel failed"));
  return done(rc);
}

}  // extern "C"
// The display pass enqueued on the context's stream, ending with the copy of the RGBA8 image to `out` (host memory; the caller synchronises).
// readDone: recorded once the accumulation image has been read, and made the event the next frame's accumulate step waits for -- frames
// rendered after this call may then overlap the rest of the pass.
static int enqueue_tonemap(pt_context* c, const pt_Tonemapper* tm, int dispW, int dispH, uint8_t* out, hipEvent_t readDone)
{
  if(!tm || !out)
    return c->fail(PT_ERR_INVALID, "pt_tonemap: null");
  if(c->width == 0)
    return c->fail(PT_ERR_STATE, "pt_tonemap before pt_resize");
  if(dispW < c->width || dispH < c->height || dispW > 32768 || dispH > 32768)
    return c->fail(PT_ERR_INVALID, "pt_tonemap_zoom: the viewport must be at least as large as the accumulation image");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = untile_to_rowmajor(c);
  if(rc != PT_OK)
    return rc;
  if(readDone)
  {
    HIP_TRY(c, hipEventRecord(readDone, c->stream));
    c->lastAccum = readDone;
  }
  // level 0: the accumulation image itself, or a viewport-sized image with it in the corner; levels 1.. only with auto-exposure
  // (src/sample_example.cpp:423-427 generates the chain only then)
  MipView mv{};
  const bool padded = dispW != c->width || dispH != c->height;
  size_t     texels = padded ? size_t(dispW) * dispH : 0, offset = texels;
  int        lw = dispW, lh = dispH, levels = 1;
  if(tm->autoExposure & 1)
    for(int m = dispW > dispH ? dispW : dispH; m > 1; m >>= 1)
    {
      lw = lw > 1 ? lw / 2 : 1;
      lh = lh > 1 ? lh / 2 : 1;
      texels += size_t(lw) * lh;
      levels++;
    }
  if(texels && (rc = dev_alloc(c, c->dMips, sizeof(float4) * texels)) != PT_OK)
    return rc;
  if((rc = dev_alloc(c, c->dRgba8, 4 * size_t(dispW) * dispH)) != PT_OK)
    return rc;
  float4* pool = (float4*)c->dMips.p;
  if(padded)
    pt_launch_pad_corner(c->stream, (const float4*)c->dRowMajor.p, c->width, c->height, pool, dispW, dispH);
  mv.level[0] = padded ? pool : (const float4*)c->dRowMajor.p;
  mv.w[0] = dispW; mv.h[0] = dispH; mv.n = levels;
  for(int i = 1; i < levels; ++i)
  {
    mv.w[i]     = mv.w[i - 1] > 1 ? mv.w[i - 1] / 2 : 1;
    mv.h[i]     = mv.h[i - 1] > 1 ? mv.h[i - 1] / 2 : 1;
    mv.level[i] = pool + offset;
    pt_launch_blit_linear(c->stream, mv.level[i - 1], mv.w[i - 1], mv.h[i - 1], pool + offset, mv.w[i], mv.h[i]);
    offset += size_t(mv.w[i]) * mv.h[i];
  }
  pt_launch_tonemap(c->stream, mv, *tm, (uint32_t*)c->dRgba8.p);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, c->dRgba8.p, 4 * size_t(dispW) * dispH, hipMemcpyDeviceToHost, c->stream));
  return PT_OK;
}
extern "C" {
int pt_tonemap_zoom(pt_context* c, const pt_Tonemapper* tm, int dispW, int dispH, uint8_t* out)
{
  CTX_CHECK(c);
  int rc = enqueue_tonemap(c, tm, dispW, dispH, out, nullptr);
  if(rc != PT_OK)
    return rc;
  HIP_TRY(c, sync_all(c));
  return check_traversal(c);
}
int pt_tonemap(pt_context* c, const pt_Tonemapper* tm, uint8_t* out)
{
  CTX_CHECK(c);
  return pt_tonemap_zoom(c, tm, c->width, c->height, out);
}

int pt_tonemap_begin(pt_context* c, const pt_Tonemapper* tm, int dispW, int dispH)
{
  CTX_CHECK(c);
  if(c->displayTail - c->displayHead >= PT_DISPLAY_RING)
    return c->fail(PT_ERR_STATE, "pt_tonemap_begin: %d images are waiting for pt_tonemap_end", PT_DISPLAY_RING);
  if(dispW <= 0 || dispH <= 0 || dispW > 32768 || dispH > 32768)
    return c->fail(PT_ERR_INVALID, "pt_tonemap_begin: viewport %dx%d", dispW, dispH);
  HIP_TRY(c, hipSetDevice(c->device));
  pt_context::DisplaySlot& ds    = c->display[c->displayTail % PT_DISPLAY_RING];
  const size_t             bytes = 4 * size_t(dispW) * dispH;
  if(ds.bytes < bytes)
  {
    if(ds.host)
      (void)hipHostFree(ds.host);
    ds.host  = nullptr;
    ds.bytes = 0;
    HIP_TRY(c, hipHostMalloc((void**)&ds.host, bytes, hipHostMallocDefault));
    ds.bytes = bytes;
  }
  if(!ds.done)
    HIP_TRY(c, hipEventCreateWithFlags(&ds.done, hipEventDisableTiming));
  if(!ds.read)
    HIP_TRY(c, hipEventCreateWithFlags(&ds.read, hipEventDisableTiming));
  int rc = enqueue_tonemap(c, tm, dispW, dispH, ds.host, ds.read);
  if(rc != PT_OK)
    return rc;
  HIP_TRY(c, hipEventRecord(ds.done, c->stream));
  ds.used = bytes;
  c->displayTail++;
  return PT_OK;
}
int pt_tonemap_pending(pt_context* c)
{
  return c ? int(c->displayTail - c->displayHead) : 0;
}
int pt_tonemap_end(pt_context* c, uint8_t* out)
{
  CTX_CHECK(c);
  if(!out)
    return c->fail(PT_ERR_INVALID, "pt_tonemap_end: null");
  if(c->displayHead == c->displayTail)
    return c->fail(PT_ERR_STATE, "pt_tonemap_end without a pending pt_tonemap_begin");
  HIP_TRY(c, hipSetDevice(c->device));
  pt_context::DisplaySlot& ds = c->display[c->displayHead % PT_DISPLAY_RING];
  HIP_TRY(c, hipEventSynchronize(ds.done));
  std::memcpy(out, ds.host, ds.used);
  c->displayHead++;
  return PT_OK;
}

int pt_local_shard(pt_context* c, void** device_ptr, size_t* bytes, int* num_local_tiles, int* max_tiles_per_rank)
{
  CTX_CHECK(c);
  if(c->width == 0)
    return c->fail(PT_ERR_STATE, "pt_local_shard before pt_resize");
  int rc = flush_pending(c);
  if(rc != PT_OK)
    return rc;
  if(device_ptr) *device_ptr = c->dFrame.p;
  if(bytes) *bytes = sizeof(float4) * size_t(c->maxTilesPerRank) * 1024u;
  if(num_local_tiles) *num_local_tiles = int(c->numLocalTiles);
  if(max_tiles_per_rank) *max_tiles_per_rank = int(c->maxTilesPerRank);
  return PT_OK;
}

int pt_scatter_shards(pt_context* c, const void* gathered_dev, int nranks)
{
  CTX_CHECK(c);
  if(!gathered_dev || nranks != c->nranks)
    return c->fail(PT_ERR_INVALID, "pt_scatter_shards: nranks %d != %d", nranks, c->nranks);
  if(c->width == 0)
    return c->fail(PT_ERR_STATE, "pt_scatter_shards before pt_resize");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t        nt = uint32_t(c->tilesX) * c->tilesY;
  std::vector<uint32_t> localIndex(nt), identity(nt), next(nranks, 0);
  for(uint32_t gt = 0; gt < nt; ++gt)
  {
    int r          = (int(gt % c->tilesX) + int(gt / c->tilesX)) % nranks;
    localIndex[gt] = next[r]++;
    identity[gt]   = gt;
  }
  int rc;
  if((rc = dev_alloc(c, c->dFullTiles, sizeof(float4) * size_t(nt) * 1024u)) != PT_OK) return rc;
  if((rc = upload(c, c->dTileLocalIndex, localIndex.data(), 4 * size_t(nt))) != PT_OK) return rc;
  if((rc = upload(c, c->dFullSlotTile, identity.data(), 4 * size_t(nt))) != PT_OK) return rc;
  pt_launch_scatter_tiles(c->stream, (const float4*)gathered_dev, nranks, int(c->maxTilesPerRank), c->tilesX, c->tilesY, (const uint32_t*)c->dTileLocalIndex.p,
                          (float4*)c->dFullTiles.p);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, sync_all(c));
  c->haveFull = true;
  return PT_OK;
}

// internal hooks of pt_comm.cpp (hidden: not part of the ABI)
__attribute__((visibility("hidden"))) int pt_comm_internal_shard(pt_context* c, void** shard, size_t* bytes, int* rank, int* nranks, int root, void** gatherBuf, hipStream_t* stream, int* device)
{
  CTX_CHECK(c);
  if(c->width == 0)
    return c->fail(PT_ERR_STATE, "pt_gather_shards before pt_resize");
  if(root < 0 || root >= c->nranks)
    return c->fail(PT_ERR_INVALID, "pt_gather_shards: root %d of %d ranks", root, c->nranks);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));  // the shard is complete; the gather is enqueued on the context's stream
  *shard  = c->dFrame.p;
  *bytes  = sizeof(float4) * size_t(c->maxTilesPerRank) * 1024u;
  *rank   = c->rank;
  *nranks = c->nranks;
  *stream = c->stream;
  *device = c->device;
  *gatherBuf       = nullptr;
  c->gatherEnqueued = false;
  if(c->rank == root)
  {  // only the root holds the nranks x shard buffer (133 MB for a 4K image on 8 GPUs)
    int rc = dev_alloc(c, c->dGather, *bytes * size_t(c->nranks));
    if(rc != PT_OK)
      return rc;
    *gatherBuf        = c->dGather.p;
    c->gatherEnqueued = true;
  }
  return PT_OK;
}
__attribute__((visibility("hidden"))) void pt_comm_internal_fail(pt_context* c, int code, const char* msg)
{
  c->gatherEnqueued = false;
  c->fail(code, "%s", msg);
}

int pt_gather_finish(pt_context* c)
{
  CTX_CHECK(c);
  if(!c->gatherEnqueued || !c->dGather.p)
    return c->fail(PT_ERR_STATE, "pt_gather_finish without a pt_gather_shards that this context enqueued as the root");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->gatherEnqueued = false;
  return pt_scatter_shards(c, c->dGather.p, c->nranks);
}

int pt_set_profiling(pt_context* c, int enable)
{
  CTX_CHECK(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));
  pt_timers_collect(&c->timers);
  c->timers.enabled = enable != 0;
  return PT_OK;
}

int pt_get_stats(pt_context* c, pt_Stats* out)
{
  CTX_CHECK(c);
  if(!out)
    return c->fail(PT_ERR_INVALID, "pt_get_stats: null");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));
  pt_timers_collect(&c->timers);
  Counters k{};
  HIP_TRY(c, hipMemcpy(&k, c->dCounters.p, sizeof(k), hipMemcpyDeviceToHost));
  if(k.stackOverflow)
    return c->fail(PT_ERR_STATE, "BVH traversal stack overflowed %u times (results are invalid)", k.stackOverflow);
  pt_Stats s = c->stats;
  s.closestRays = k.closestRays; s.shadowRays = k.shadowRays; s.shadedHits = k.shadedHits; s.misses = k.misses; s.alphaTests = k.alphaTests;
  s.neeLookups = k.neeLookups; s.nodesVisited = k.nodesVisited; s.trisTested = k.trisTested;
  s.msGenerate = c->timers.ms[0]; s.msTraceClosest = c->timers.ms[1]; s.msShade = c->timers.ms[2]; s.msTraceShadow = c->timers.ms[3];
  s.msAccumulate = c->timers.ms[4];
  s.msTail       = c->timers.ms[5];
  s.launchesTraceClosest = c->timers.launchesClosest;
  s.launchesTail         = c->timers.launchesTail;
  s.msTraceFused         = c->timers.ms[6];
  s.launchesTraceFused   = c->timers.launchesFused;
  s.numMergedTriangles   = c->mergedTris;
  s.tailClosestRays = k.tailClosestRays; s.tailShadowRays = k.tailShadowRays; s.tailShadedHits = k.tailShadedHits; s.tailMisses = k.tailMisses;
  s.tailAlphaTests = k.tailAlphaTests;
  s.numTriangles = c->numTris;
  s.numBvhNodes  = PT_BVH_WIDTH == 2 ? c->numBvhNodes : c->numWideNodes;
  s.msBuildAccel = c->msBuild;
  s.numBlas      = c->numBlas;
  s.numTlasNodes = c->numTlasNodes;
  s.batchFrames    = uint32_t(c->batchMax);
  s.framesInFlight = uint32_t(c->inflight);
  s.bytesAccel   = c->dBvh.bytes + c->dWide.bytes + c->dTris.bytes + c->dAlphaRecs.bytes + c->dTlas.bytes + c->dTlasLeaves.bytes + c->dInstTriBase.bytes + c->dCNodes.bytes + c->dCTlas.bytes + c->dShadeTris.bytes;
  uint64_t bytes = 0;
  const DevBuf* sb[] = {&c->dVertices, &c->dIndices, &c->dInstances, &c->dMaterials, &c->dLights, &c->dTexRecs, &c->dTexels, &c->dBvh, &c->dWide, &c->dTris, &c->dAlphaRecs, &c->dAlphaMats, &c->dAlphaMaps, &c->dEnv, &c->dEnvAccel,
                        &c->dTlas, &c->dTlasLeaves, &c->dInstTriBase};
  for(const DevBuf* b : sb)
    bytes += b->bytes;
  s.bytesScene = bytes;
  *out         = s;
  return PT_OK;
}

int pt_reset_stats(pt_context* c)
{
  CTX_CHECK(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_all(c));
  pt_timers_collect(&c->timers);
  for(double& m : c->timers.ms)
    m = 0;
  c->timers.launchesClosest = 0;
  c->timers.launchesTail    = 0;
  c->timers.launchesFused   = 0;
  c->stats                  = pt_Stats{};
  HIP_TRY(c, hipMemset(c->dCounters.p, 0, sizeof(Counters)));
  return PT_OK;
}

}  // extern "C"

// Test hook (not part of the ABI; tests/cpp/trace_host.cpp): the host-side records pt_set_scene derives from a scene description, copied into
// caller arrays (no GPU involved).  Call with null outputs to get the counts: counts[0] instances, [1] materials, [2] opacity-map words,
// [3] texels of the RGBA8 pool, [4] world triangles.  instOut: InstanceRec[counts[0]] (128 B each); padOut: 2 floats per instance
// (TlasLeaf::padC0 / padC1 of the two-level walk); alphaMatsOut: AlphaMat[counts[1]] (80 B each); texelsOut: the pool in upload order;
// texRecsOut: TexRec[max(1, numTextures)] (32 B each).
extern "C" __attribute__((visibility("default"))) int pt_debug_scene_records(const pt_SceneDesc* d, unsigned long long* counts5, void* instOut, float* padOut, void* alphaMatsOut,
                                                                            uint32_t* alphaMapsOut, uint32_t* texelsOut, void* texRecsOut, char* err, size_t errLen)
{
  SceneRecords R;
  std::string  msg, unknown;
  PtTuning     tune;  // the knobs a context created now would get
  pt_parse_tuning(getenv("PT_TUNE"), tune, unknown);
  const int    rc = build_scene_records(d, R, msg, tune.texTile, tune.texGroups);
  if(rc != PT_OK)
  {
    if(err && errLen)
      snprintf(err, errLen, "%s", msg.c_str());
    return rc;
  }
  if(counts5)
  {
    counts5[0] = R.inst.size(); counts5[1] = R.alphaMats.size(); counts5[2] = R.alphaMaps.size(); counts5[3] = R.texels; counts5[4] = R.triTotal;
  }
  if(instOut)
    std::memcpy(instOut, R.inst.data(), sizeof(InstanceRec) * R.inst.size());
  if(padOut)
    for(size_t i = 0; i < R.inst.size(); ++i)
      two_level_pad(R.inst[i], R.primBound[R.inst[i].primMesh], padOut[2 * i], padOut[2 * i + 1]);
  if(alphaMatsOut)
    std::memcpy(alphaMatsOut, R.alphaMats.data(), sizeof(AlphaMat) * R.alphaMats.size());
  if(alphaMapsOut)
    std::memcpy(alphaMapsOut, R.alphaMaps.data(), 4 * R.alphaMaps.size());
  if(texRecsOut)
    std::memcpy(texRecsOut, R.texRecs.data(), sizeof(TexRec) * R.texRecs.size());
  if(texelsOut)
  {
    if(d->numTextures == 0)
      texelsOut[0] = 0xffffffffu;
    for(uint32_t t = 0; t < d->numTextures; ++t)
      store_texture(texelsOut + R.texRecs[t].offset, R.texRecs[t], d->textures[t].rgba8);
    for(const SceneRecords::TexGroup& g : R.groups)
      store_group(texelsOut + g.offset, g, R.texRecs, d);
  }
  return PT_OK;
}
// ... and the material lines (PT_MAT_LINE_QUADS x 16 bytes per material) whose descriptors point into that pool
extern "C" __attribute__((visibility("default"))) int pt_debug_mat_lines(const pt_SceneDesc* d, void* linesOut, char* err, size_t errLen)
{
  SceneRecords R;
  std::string  msg, unknown;
  PtTuning     tune;  // the knobs a context created now would get
  pt_parse_tuning(getenv("PT_TUNE"), tune, unknown);
  const int    rc = build_scene_records(d, R, msg, tune.texTile, tune.texGroups);
  if(rc != PT_OK)
  {
    if(err && errLen)
      snprintf(err, errLen, "%s", msg.c_str());
    return rc;
  }
  if(linesOut)
    std::memcpy(linesOut, R.matLines.data(), sizeof(uint4) * R.matLines.size());
  return PT_OK;
}

// Test hook (not part of the ABI): the launch-policy decision of flush_pending on plain numbers
extern "C" __attribute__((visibility("default"))) int pt_debug_tail_from(double paths, int maxDepth, int tailBelow, const double* ratio, int numObserved)
{
  return tail_from_depth(paths, maxDepth, tailBelow, ratio, numObserved);
}

// Test hook (not part of the ABI; CPU tests hold the bound to a float32 emulation of the ray transform): the instance record pt_set_scene
// derives from a node's world matrix and the object-space padding of the two-level walk for a mesh whose |coordinates| are <= Bo.
// out: objectToWorld rows (12), worldToObject rows (12), padC0, padC1, flags
extern "C" __attribute__((visibility("default"))) int pt_debug_two_level_pad(const float* worldMatrix16, float Bo, float* out27)
{
  InstanceRec I{};
  if(!worldMatrix16 || !out27 || !set_instance_transform(I, worldMatrix16, 0u))
    return PT_ERR_INVALID;
  const float4 rows[6] = {I.objectToWorld.r0, I.objectToWorld.r1, I.objectToWorld.r2, I.worldToObject.r0, I.worldToObject.r1, I.worldToObject.r2};
  for(int r = 0; r < 6; ++r)
  {
    out27[4 * r] = rows[r].x; out27[4 * r + 1] = rows[r].y; out27[4 * r + 2] = rows[r].z; out27[4 * r + 3] = rows[r].w;
  }
  two_level_pad(I, Bo, out27[24], out27[25]);
  out27[26] = float(I.flags);
  return PT_OK;
}

