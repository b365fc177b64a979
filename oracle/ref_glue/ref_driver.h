// TEST INFRASTRUCTURE -- part of the oracle/_ref recipe (see oracle/ref_glue/README.md). Not linked into the product.
//
// The part of the reference's execution environment that is NOT in /root/reference: what the Vulkan driver and the
// GLSL extensions provide to shaders/pathtrace.comp -- opaque types (sampler2D, image2D, accelerationStructureEXT,
// rayQueryEXT), texture / image access, the ray-query built-ins and a few constants.  The implementation-defined
// behaviour (triangle candidates and their order, the object<->world matrices of an instance, bilinear filtering)
// comes in through RefHooks; tests bind them to the oracle's BVH-independent trace contract (oracle/orc_scene.h T1-T6)
// and its Appendix-F sampler, so that what is compared is the reference's SHADER CODE against the oracle's restatement.
#pragma once
#include <cstdint>
#include "glsl_compat.h"

extern "C" {
struct RefHooks {
  void* user;
  // smallest candidate key (t, w) > (tPrev, wPrev) with 0 < t < tmax; want = 0 every triangle, 1 non-opaque only, 2 opaque only
  // (any one).  Returns 0 when there is none.  Face culling (instance flags + gl_RayFlagsCullBackFacingTrianglesEXT) is the hook's.
  int (*query)(void* user, const float* o, const float* d, float tmax, float tPrev, uint32_t wPrev, int want, float* t, float* u, float* v, uint32_t* w);
  // world triangle w -> TLAS instance (node), primitive index inside its BLAS, instance custom index, VK_GEOMETRY_INSTANCE_FORCE_OPAQUE
  void (*tri_info)(void* user, uint32_t w, int* node, int* prim, int* customIndex, int* opaque);
  // column-major 4x3 matrices of a TLAS instance (rayQueryGetIntersectionObjectToWorldEXT / WorldToObjectEXT)
  void (*instance)(void* user, int node, float* o2w12, float* w2o12);
  // combined image samplers of descriptor set 2 (scene textures, LOD 0) and of the environment map
  void (*sample_texture)(void* user, int id, float u, float v, float* rgba);
  void (*sample_env)(void* user, float u, float v, float* rgb);
};
}

namespace glslc {

extern RefHooks g_hooks;

// ---- images ---------------------------------------------------------------------------------------------------
struct RefMip {
  int          w, h;
  const float* px;  // RGBA32F
};
struct sampler2D {
  int           kind = 0;  // 0: scene texture `id` (hook), 1: environment (hook), 2: RGBA32F image with a mip chain, NEAREST / NEAREST / REPEAT
  int           id = 0;
  int           w = 0, h = 0;
  const RefMip* mips = nullptr;
  int           numMips = 0;
};
struct image2D {
  float* px = nullptr;  // RGBA32F, row-major
  int    w = 0, h = 0;
};
inline vec4 fetch_nearest(const sampler2D& s, vec2 uv, int level)
{
  // Vulkan: VK_FILTER_NEAREST, VK_SAMPLER_MIPMAP_MODE_NEAREST, VK_SAMPLER_ADDRESS_MODE_REPEAT (render_output.cpp:98-100: a zeroed
  // VkSamplerCreateInfo with maxLod = FLT_MAX)
  if(level < 0) level = 0;
  if(level > s.numMips - 1) level = s.numMips - 1;
  const RefMip& m = s.mips[level];
  int           i = (int)::floorf(uv.x * float(m.w)), j = (int)::floorf(uv.y * float(m.h));
  i %= m.w; if(i < 0) i += m.w;
  j %= m.h; if(j < 0) j += m.h;
  const float* p = m.px + (size_t(j) * m.w + i) * 4;
  return vec4(p[0], p[1], p[2], p[3]);
}
inline vec4 texture(const sampler2D& s, vec2 uv, float bias = 0.0f)
{
  if(s.kind == 0)
  {
    float r[4];
    g_hooks.sample_texture(g_hooks.user, s.id, uv.x, uv.y, r);
    return vec4(r[0], r[1], r[2], r[3]);
  }
  if(s.kind == 1)
  {
    float r[3];
    g_hooks.sample_env(g_hooks.user, uv.x, uv.y, r);
    return vec4(r[0], r[1], r[2], 1.0f);
  }
  // full-screen pass at 1:1 (or magnified): implicit LOD 0, so the level is the bias, rounded (mipmapMode NEAREST)
  return fetch_nearest(s, uv, (int)::floorf(bias + 0.5f));
}
inline vec4 textureLod(const sampler2D& s, vec2 uv, float lod)
{
  if(s.kind == 2)
    return fetch_nearest(s, uv, (int)::floorf(lod + 0.5f));
  return texture(s, uv);
}
inline ivec2 textureSize(const sampler2D& s, int) { return ivec2(s.w, s.h); }
inline vec4  imageLoad(const image2D& im, ivec2 p)
{
  const float* q = im.px + (size_t(p.y) * im.w + p.x) * 4;
  return vec4(q[0], q[1], q[2], q[3]);
}
inline void imageStore(const image2D& im, ivec2 p, vec4 v)
{
  float* q = im.px + (size_t(p.y) * im.w + p.x) * 4;
  q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
}
template <class T> inline T nonuniformEXT(T x) { return x; }

// ---- built-in variables -----------------------------------------------------------------------------------------
struct GlobalInvocationID {
  uint  x, y, z;
  uvec2 xy;
};
extern thread_local GlobalInvocationID gl_GlobalInvocationID;
extern thread_local vec4               gl_FragCoord;
extern thread_local uint64_t           g_clock;  // clockRealtimeEXT(): a deterministic tick count (one tick per ray-query candidate / commit)
inline uint64_t                        clockRealtimeEXT() { return g_clock; }

// ---- GL_EXT_ray_query -------------------------------------------------------------------------------------------
struct accelerationStructureEXT {
  int dummy = 0;
};
const uint gl_RayFlagsNoneEXT                       = 0u;
const uint gl_RayFlagsOpaqueEXT                     = 1u;
const uint gl_RayFlagsNoOpaqueEXT                   = 2u;
const uint gl_RayFlagsTerminateOnFirstHitEXT        = 4u;
const uint gl_RayFlagsSkipClosestHitShaderEXT       = 8u;
const uint gl_RayFlagsCullBackFacingTrianglesEXT    = 16u;
const uint gl_RayQueryCommittedIntersectionNoneEXT     = 0u;
const uint gl_RayQueryCommittedIntersectionTriangleEXT = 1u;
const uint gl_RayQueryCandidateIntersectionTriangleEXT = 0u;

struct RqHit {
  float    t = 0, u = 0, v = 0;
  uint32_t w = 0;
};
struct rayQueryEXT {
  vec3     o, d;
  float    tmax = 0;
  uint     flags = 0;
  bool     done = false, committed = false;
  float    tPrev = 0;
  uint32_t wPrev = 0xffffffffu;
  RqHit    cand, hit;
};
inline void rayQueryInitializeEXT(rayQueryEXT& q, const accelerationStructureEXT&, uint flags, uint /*cullMask*/, vec3 o, float /*tmin = 0*/, vec3 d, float tmax)
{
  q       = rayQueryEXT();
  q.o     = o;
  q.d     = d;
  q.tmax  = tmax;
  q.flags = flags;
}
inline bool rq_query(rayQueryEXT& q, float tmax, int want, RqHit& out)
{
  return g_hooks.query(g_hooks.user, q.o.d, q.d.d, tmax, q.tPrev, q.wPrev, want, &out.t, &out.u, &out.v, &out.w) != 0;
}
// Candidates are produced in the order of the trace contract (oracle/orc_scene.h T4-T6): by key (t, w); opaque candidates commit
// without the shader; with gl_RayFlagsTerminateOnFirstHitEXT the first commit ends the query.
inline bool rayQueryProceedEXT(rayQueryEXT& q)
{
  if(q.done)
    return false;
  for(;;)
  {
    RqHit h;
    if(!rq_query(q, q.committed ? q.hit.t : q.tmax, 0, h))
    {
      q.done = true;
      return false;
    }
    g_clock++;
    q.tPrev = h.t;
    q.wPrev = h.w;
    int node, prim, custom, opaque;
    g_hooks.tri_info(g_hooks.user, h.w, &node, &prim, &custom, &opaque);
    if(opaque)
    {
      q.hit = h; q.committed = true;
      if(q.flags & gl_RayFlagsTerminateOnFirstHitEXT) { q.done = true; return false; }
      continue;
    }
    q.cand = h;
    return true;
  }
}
inline void rayQueryConfirmIntersectionEXT(rayQueryEXT& q)
{
  q.hit = q.cand; q.committed = true;
  g_clock++;
  if(q.flags & gl_RayFlagsTerminateOnFirstHitEXT)
    q.done = true;
}
inline uint rayQueryGetIntersectionTypeEXT(const rayQueryEXT& q, bool committed)
{
  if(committed)
    return q.committed ? gl_RayQueryCommittedIntersectionTriangleEXT : gl_RayQueryCommittedIntersectionNoneEXT;
  return gl_RayQueryCandidateIntersectionTriangleEXT;
}
inline const RqHit& rq_sel(const rayQueryEXT& q, bool committed) { return committed ? q.hit : q.cand; }
inline float rayQueryGetIntersectionTEXT(const rayQueryEXT& q, bool c) { return rq_sel(q, c).t; }
inline vec2  rayQueryGetIntersectionBarycentricsEXT(const rayQueryEXT& q, bool c) { return vec2(rq_sel(q, c).u, rq_sel(q, c).v); }
inline int   rq_info(const rayQueryEXT& q, bool c, int which)
{
  int node, prim, custom, opaque;
  g_hooks.tri_info(g_hooks.user, rq_sel(q, c).w, &node, &prim, &custom, &opaque);
  return which == 0 ? node : (which == 1 ? prim : custom);
}
inline int    rayQueryGetIntersectionInstanceIdEXT(const rayQueryEXT& q, bool c) { return rq_info(q, c, 0); }
inline int    rayQueryGetIntersectionPrimitiveIndexEXT(const rayQueryEXT& q, bool c) { return rq_info(q, c, 1); }
inline int    rayQueryGetIntersectionInstanceCustomIndexEXT(const rayQueryEXT& q, bool c) { return rq_info(q, c, 2); }
inline mat4x3 rq_matrix(const rayQueryEXT& q, bool c, int which)
{
  float a[12], b[12];
  g_hooks.instance(g_hooks.user, rq_info(q, c, 0), a, b);
  const float* m = which ? b : a;
  mat4x3       r;
  for(int i = 0; i < 4; ++i)
    r.c[i] = vec3(m[3 * i], m[3 * i + 1], m[3 * i + 2]);
  return r;
}
inline mat4x3 rayQueryGetIntersectionObjectToWorldEXT(const rayQueryEXT& q, bool c) { return rq_matrix(q, c, 0); }
inline mat4x3 rayQueryGetIntersectionWorldToObjectEXT(const rayQueryEXT& q, bool c) { return rq_matrix(q, c, 1); }


// ---- GL_EXT_ray_tracing (the RtxPipeline flavour: pathtrace.rgen / .rchit / .rahit / .rmiss / pathtraceShadow.rmiss) ---------------
// Built-in variables of the stages, and traceRayEXT as a call into the pipeline emulation of ref_rtx_rgen.cpp: candidates in the order of the
// trace contract; an instance with FORCE_OPAQUE commits without the any-hit stage, any other candidate runs pathtrace.rahit on the payload
// the trace call named (that is what makes the shadow ray's alpha tests draw from a COPY of the path's seed: pathtrace.rahit declares the
// payload of location 0 and is handed the one of location 1, traceray_rtx.glsl:54-55); the closest-hit stage unless
// gl_RayFlagsSkipClosestHitShaderEXT; the miss stage `missIndex` when nothing was committed.
extern thread_local GlobalInvocationID gl_LaunchIDEXT, gl_LaunchSizeEXT;  // uvec3 with its .xy swizzle
extern thread_local float  gl_HitTEXT;
extern thread_local int    gl_PrimitiveID, gl_InstanceID, gl_InstanceCustomIndexEXT;
extern thread_local mat4x3 gl_ObjectToWorldEXT, gl_WorldToObjectEXT;
extern thread_local bool   gl_IgnoreIntersection;
typedef void (*RtxTraceFn)(uint flags, uint missIndex, vec3 origin, vec3 dir, float tmax, int payload);
extern RtxTraceFn g_rtxTrace;
inline void traceRayEXT(const accelerationStructureEXT&, uint flags, uint /*cullMask*/, uint, uint, uint missIndex, vec3 origin, float /*tmin = 0*/, vec3 dir, float tmax, int payload)
{
  g_rtxTrace(flags, missIndex, origin, dir, tmax, payload);
}

}  // namespace glslc
