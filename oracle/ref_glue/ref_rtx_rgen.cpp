// TEST INFRASTRUCTURE -- part of the oracle/_ref recipe (see README.md). Not linked into the product.
//
// The reference's SECOND renderer flavour, RtxPipeline (src/rtx_pipeline.cpp): shaders/pathtrace.rgen with traceray_rtx.glsl and the shared
// pathtrace.glsl, compiled from the lexically rewritten sources, dispatched like vkCmdTraceRaysKHR (one raygen invocation per pixel), with
// traceRayEXT emulated on the trace contract: hit groups {pathtrace.rchit, pathtrace.rahit (absent after useAnyHit(false),
// src/rtx_pipeline.cpp:186-195)}, miss stages {pathtrace.rmiss, pathtraceShadow.rmiss}.
#include <omp.h>
#include <cstring>
#include <vector>
#include "../../include/pt_types.h"
#include "ref_driver.h"

namespace glslc {
void ref_rtx_rchit(void* payload, size_t bytes, vec2 attribs);
bool ref_rtx_rahit(void* payload, size_t bytes, vec2 attribs);
void ref_rtx_rahit_bind(const pt_SceneDesc* d);
void ref_rtx_miss0(void* payload, size_t bytes);
void ref_rtx_miss1(void* payload, size_t bytes);

namespace rgen {
#include "pathtrace.rgen"  // the generated file in the scratch directory (reference: shaders/pathtrace.rgen)

static std::vector<InstanceData> s_geoInfo;
static std::vector<sampler2D>    s_textures;
static bool                      s_anyHit = true;

static void* payload_ptr(int loc, size_t& bytes)
{
  if(loc == 0)
  {
    bytes = sizeof(prd);
    return &prd;
  }
  bytes = sizeof(shadow_payload);
  return &shadow_payload;
}

// vkCmdTraceRaysKHR's traversal for one ray
static void trace(uint flags, uint missIndex, vec3 origin, vec3 dir, float tmax, int payloadLoc)
{
  size_t bytes   = 0;
  void*  payload = payload_ptr(payloadLoc, bytes);
  float  tPrev = 0.0f;
  uint32_t wPrev = 0xffffffffu;
  bool   committed = false;
  RqHit  hit;
  for(;;)
  {
    RqHit h;
    if(!g_hooks.query(g_hooks.user, origin.d, dir.d, committed ? hit.t : tmax, tPrev, wPrev, 0, &h.t, &h.u, &h.v, &h.w))
      break;
    tPrev = h.t;
    wPrev = h.w;
    int node, prim, custom, opaque;
    g_hooks.tri_info(g_hooks.user, h.w, &node, &prim, &custom, &opaque);
    bool accept = true;
    if(!opaque && s_anyHit)
    {
      float a[12], b[12];
      g_hooks.instance(g_hooks.user, node, a, b);
      gl_HitTEXT = h.t; gl_PrimitiveID = prim; gl_InstanceID = node; gl_InstanceCustomIndexEXT = custom;
      for(int i = 0; i < 4; ++i)
      {
        gl_ObjectToWorldEXT.c[i] = vec3(a[3 * i], a[3 * i + 1], a[3 * i + 2]);
        gl_WorldToObjectEXT.c[i] = vec3(b[3 * i], b[3 * i + 1], b[3 * i + 2]);
      }
      accept = ref_rtx_rahit(payload, bytes, vec2(h.u, h.v));
    }
    if(accept)
    {
      committed = true;
      hit       = h;
      if(flags & gl_RayFlagsTerminateOnFirstHitEXT)
        break;
    }
  }
  if(committed)
  {
    if(!(flags & gl_RayFlagsSkipClosestHitShaderEXT))
    {
      int node, prim, custom, opaque;
      g_hooks.tri_info(g_hooks.user, hit.w, &node, &prim, &custom, &opaque);
      float a[12], b[12];
      g_hooks.instance(g_hooks.user, node, a, b);
      gl_HitTEXT = hit.t; gl_PrimitiveID = prim; gl_InstanceID = node; gl_InstanceCustomIndexEXT = custom;
      for(int i = 0; i < 4; ++i)
      {
        gl_ObjectToWorldEXT.c[i] = vec3(a[3 * i], a[3 * i + 1], a[3 * i + 2]);
        gl_WorldToObjectEXT.c[i] = vec3(b[3 * i], b[3 * i + 1], b[3 * i + 2]);
      }
      ref_rtx_rchit(payload, bytes, vec2(hit.u, hit.v));
    }
  }
  else if(missIndex == 0)
    ref_rtx_miss0(payload, bytes);
  else
    ref_rtx_miss1(payload, bytes);
}
}  // namespace rgen
}  // namespace glslc

using namespace glslc;
using namespace glslc::rgen;

extern "C" {

int ref_rtx_bind(const pt_SceneDesc* d, const pt_EnvAccel* envAccel, int envW, int envH, const RefHooks* hooks)
{
  g_hooks    = *hooks;
  g_rtxTrace = &rgen::trace;
  s_geoInfo.resize(d->numPrimMeshes);
  for(uint32_t i = 0; i < d->numPrimMeshes; ++i)
  {
    s_geoInfo[i].vertexAddress = (uint64_t)(uintptr_t)(d->vertices + d->primMeshes[i].vertexOffset);
    s_geoInfo[i].indexAddress  = (uint64_t)(uintptr_t)(d->indices + d->primMeshes[i].firstIndex);
    s_geoInfo[i].materialIndex = d->primMeshes[i].materialIndex;
  }
  geoInfo   = s_geoInfo.data();
  materials = reinterpret_cast<const GltfShadeMaterial*>(d->materials);
  lights    = reinterpret_cast<const Light*>(d->lights);
  s_textures.resize(d->numTextures);
  for(uint32_t i = 0; i < d->numTextures; ++i)
  {
    s_textures[i].kind = 0;
    s_textures[i].id   = (int)i;
    s_textures[i].w    = d->textures[i].width;
    s_textures[i].h    = d->textures[i].height;
  }
  texturesMap             = s_textures.data();
  environmentTexture.kind = 1;
  environmentTexture.w    = envW;
  environmentTexture.h    = envH;
  envSamplingData         = reinterpret_cast<const EnvAccel*>(envAccel);
  ref_rtx_rahit_bind(d);
  return 0;
}
int ref_rtx_set_camera(const pt_SceneCamera* c)
{
  std::memcpy(&sceneCamera, c, sizeof(SceneCamera));
  return 0;
}
int ref_rtx_set_sunsky(const pt_SunAndSky* s)
{
  std::memcpy(&_sunAndSky, s, sizeof(SunAndSky));
  return 0;
}
// RtxPipeline::useAnyHit (src/rtx_pipeline.cpp:269-276): 0 = hit groups without the any-hit stage
int ref_rtx_use_any_hit(int enable)
{
  s_anyHit = enable != 0;
  return 0;
}
// vkCmdTraceRaysKHR(size.width, size.height, 1) of pathtrace.rgen (src/rtx_pipeline.cpp:253-267), or over the listed pixels only
int ref_rtx_render_frame(const pt_RtxState* st, float* accum, const uint32_t* pixel_ids, uint64_t n, int threads)
{
  std::memcpy(&rtxState, st, sizeof(RtxState));
  resultImage.px = accum;
  resultImage.w  = st->size[0];
  resultImage.h  = st->size[1];
  const int     W     = st->size[0], H = st->size[1];
  const int64_t total = pixel_ids ? (int64_t)n : (int64_t)W * H;
  if(threads <= 0)
    threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
  for(int64_t i = 0; i < total; ++i)
  {
    uint32_t id      = pixel_ids ? pixel_ids[i] : (uint32_t)i;
    gl_LaunchIDEXT.x = id % (uint32_t)W; gl_LaunchIDEXT.y = id / (uint32_t)W; gl_LaunchIDEXT.z = 0u;
    gl_LaunchIDEXT.xy = uvec2(gl_LaunchIDEXT.x, gl_LaunchIDEXT.y);
    gl_LaunchSizeEXT.x = (uint32_t)W; gl_LaunchSizeEXT.y = (uint32_t)H; gl_LaunchSizeEXT.z = 1u;
    gl_LaunchSizeEXT.xy = uvec2((uint32_t)W, (uint32_t)H);
    g_clock          = 0;
    shader_main();
  }
  return 0;
}
}
