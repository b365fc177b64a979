// TEST INFRASTRUCTURE -- part of the oracle/_ref recipe (see oracle/ref_glue/README.md). Not linked into the product.
//
// Compiles the reference's ray-query compute shader -- shaders/pathtrace.comp and everything it includes, rewritten
// lexically by glsl2cpp.py into the scratch directory given with -I -- as C++, and dispatches it like vkCmdDispatch does:
// one invocation per pixel with gl_GlobalInvocationID set, descriptor sets / push constants filled from plain pointers.
#include <omp.h>
#include <cstring>
#include <vector>
#include "../../include/pt_types.h"
#include "ref_driver.h"

namespace glslc {
RefHooks                               g_hooks{};
thread_local GlobalInvocationID        gl_GlobalInvocationID;
thread_local vec4                      gl_FragCoord;
thread_local uint64_t                  g_clock = 0;
namespace refcomp {
#include "pathtrace.comp"  // the generated file in the scratch directory (reference: shaders/pathtrace.comp)

// the structures the reference shares between host and device have the layout the C ABI assumes (host_device.h vs pt_types.h)
static_assert(sizeof(RtxState) == sizeof(pt_RtxState) && sizeof(SceneCamera) == sizeof(pt_SceneCamera), "host_device.h layout");
static_assert(sizeof(VertexAttributes) == sizeof(pt_VertexAttributes) && sizeof(GltfShadeMaterial) == sizeof(pt_GltfShadeMaterial), "host_device.h layout");
static_assert(sizeof(Light) == sizeof(pt_Light) && sizeof(EnvAccel) == sizeof(pt_EnvAccel) && sizeof(SunAndSky) == sizeof(pt_SunAndSky), "host_device.h layout");
static_assert(sizeof(InstanceData) == 24 && sizeof(uvec3) == 12, "host_device.h layout");

static std::vector<InstanceData> s_geoInfo;
static std::vector<sampler2D>    s_textures;
}  // namespace refcomp
}  // namespace glslc

using namespace glslc;
using namespace glslc::refcomp;

extern "C" {

// Descriptor sets 0, 2, 3 (shaders/layouts.glsl:37-50).  Every pointer is borrowed; the caller keeps the arrays alive.
int ref_bind(const pt_SceneDesc* d, const pt_EnvAccel* envAccel, int envW, int envH, const RefHooks* hooks)
{
  g_hooks = *hooks;
  s_geoInfo.resize(d->numPrimMeshes);
  for(uint32_t i = 0; i < d->numPrimMeshes; ++i)
  {
    // src/scene.cpp:161-176: one InstanceData per prim-mesh with the addresses of ITS vertex / index buffer
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
  return 0;
}
int ref_set_camera(const pt_SceneCamera* c)
{
  std::memcpy(&sceneCamera, c, sizeof(SceneCamera));
  return 0;
}
int ref_set_sunsky(const pt_SunAndSky* s)
{
  std::memcpy(&_sunAndSky, s, sizeof(SunAndSky));
  return 0;
}

// vkCmdPushConstants + vkCmdDispatch of pathtrace.comp (src/rayquery.cpp:97-109) over the image, or over the listed pixels only.
// heat != NULL receives the per-pixel clockRealtimeEXT() delta the heat-map debug mode colours.
int ref_render_frame(const pt_RtxState* st, float* accum, const uint32_t* pixel_ids, uint64_t n, int threads)
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
    uint32_t id = pixel_ids ? pixel_ids[i] : (uint32_t)i;
    gl_GlobalInvocationID.x  = id % (uint32_t)W;
    gl_GlobalInvocationID.y  = id / (uint32_t)W;
    gl_GlobalInvocationID.z  = 0;
    gl_GlobalInvocationID.xy = uvec2(gl_GlobalInvocationID.x, gl_GlobalInvocationID.y);
    g_clock                  = 0;
    shader_main();
  }
  return 0;
}

// The same dispatch for `nframes` consecutive frames (rtxState.frame = first_frame ...) inside ONE thread team: the CPU-baseline leg of bench.py
// times this, and a fork / join of a few hundred threads per frame would be most of a small frame's time.  Frames stay ordered (the running
// mean of pathtrace.comp:122-133 folds them in frame order per pixel): the push constant changes between two work-sharing loops, behind a barrier.
int ref_render_frames(const pt_RtxState* st, int first_frame, int nframes, float* accum, const uint32_t* pixel_ids, uint64_t n, int threads)
{
  std::memcpy(&rtxState, st, sizeof(RtxState));
  resultImage.px = accum;
  resultImage.w  = st->size[0];
  resultImage.h  = st->size[1];
  const int     W     = st->size[0], H = st->size[1];
  const int64_t total = pixel_ids ? (int64_t)n : (int64_t)W * H;
  if(threads <= 0)
    threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
  for(int f = first_frame; f < first_frame + nframes; ++f)
  {
#pragma omp single
    rtxState.frame = f;  // (implicit barrier: every thread sees the new push constant)
#pragma omp for schedule(dynamic, 16)
    for(int64_t i = 0; i < total; ++i)
    {
      uint32_t id = pixel_ids ? pixel_ids[i] : (uint32_t)i;
      gl_GlobalInvocationID.x  = id % (uint32_t)W;
      gl_GlobalInvocationID.y  = id / (uint32_t)W;
      gl_GlobalInvocationID.z  = 0;
      gl_GlobalInvocationID.xy = uvec2(gl_GlobalInvocationID.x, gl_GlobalInvocationID.y);
      g_clock                  = 0;
      shader_main();
    }
  }
  return 0;
}

// ---- function-level known answers straight from the reference's GLSL ------------------------------------------------------
uint32_t ref_tea(uint32_t a, uint32_t b) { return tea(a, b); }                        // random.glsl:34-48
void     ref_pcg_stream(uint32_t seed, uint32_t n, uint32_t* words, float* floats, uint32_t* state)
{
  for(uint32_t i = 0; i < n; ++i)
  {
    uint32_t s1 = seed, s2 = seed;
    uint32_t w = pcg(s1);      // random.glsl:59-65
    float    f = rand(s2);     // random.glsl:98-102
    if(words) words[i] = w;
    if(floats) floats[i] = f;
    seed = s1;
  }
  if(state) *state = seed;
}
void ref_pcg3d(uint32_t* v)                                                           // random.glsl:81-92
{
  uvec3 r = pcg3d(uvec3(v[0], v[1], v[2]));
  v[0] = r.x; v[1] = r.y; v[2] = r.z;
}
uint32_t ref_compress_unit_vec(const float* v) { return compress_unit_vec(vec3(v[0], v[1], v[2])); }   // compress.glsl:111-139 (device flavour)
void     ref_decompress_unit_vec(uint32_t p, float* o)                                                  // compress.glsl:149-180
{
  vec3 r = decompress_unit_vec(p);
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void ref_offset_ray(const float* p, const float* nrm, float* o)                                        // common.glsl:96-113
{
  vec3 r = OffsetRay(vec3(p[0], p[1], p[2]), vec3(nrm[0], nrm[1], nrm[2]));
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void ref_spherical_uv(const float* d, float* o)                                                        // common.glsl:67-74
{
  vec2 r = GetSphericalUv(vec3(d[0], d[1], d[2]));
  o[0] = r.x; o[1] = r.y;
}
void ref_coordinate_system(const float* nrm, float* t, float* b)                                       // common.glsl:80-91
{
  vec3 T, B;
  CreateCoordinateSystem(vec3(nrm[0], nrm[1], nrm[2]), T, B);
  t[0] = T.x; t[1] = T.y; t[2] = T.z; b[0] = B.x; b[1] = B.y; b[2] = B.z;
}
void ref_temperature(float x, float* o)                                                                // common.glsl:39-62
{
  vec3 c = temperature(x);
  o[0] = c.x; o[1] = c.y; o[2] = c.z;
}
void ref_sun_and_sky(const pt_SunAndSky* ss, const float* dir, float* o)                               // sun_and_sky.glsl:453-603
{
  SunAndSky s;
  std::memcpy(&s, ss, sizeof(s));
  vec3 c = sun_and_sky(s, vec3(dir[0], dir[1], dir[2]));
  o[0] = c.x; o[1] = c.y; o[2] = c.z;
}
float ref_range_attenuation(float range, float dist) { return getRangeAttenuation(range, dist); }      // punctual.glsl:28-36
float ref_spot_attenuation(const float* p2l, const float* dir, float outerCos, float innerCos)         // punctual.glsl:39-51
{
  return getSpotAttenuation(vec3(p2l[0], p2l[1], p2l[2]), vec3(dir[0], dir[1], dir[2]), outerCos, innerCos);
}

// One BSDF evaluation / sample on a synthetic shading state (pbr_disney.glsl:414-602, pbr_gltf.glsl:365-556).  `m` = 24 floats:
// albedo3 specular emission3(unused) anisotropy metallic roughness subsurface specularTint sheen sheenTint3 clearcoat clearcoatRoughness
// transmission ior ax ay f0_3 -> see the assignments below; frame = (N, T, B), eta, thinwalled.
static void fill_state(State& st, const float* m, const float* N, const float* T, const float* B, float eta, int thin)
{
  st.depth = 0; st.eta = eta;
  st.position = vec3(0); st.normal = vec3(N[0], N[1], N[2]); st.ffnormal = st.normal;
  st.tangent = vec3(T[0], T[1], T[2]); st.bitangent = vec3(B[0], B[1], B[2]); st.texCoord = vec2(0);
  st.isEmitter = false; st.specularBounce = false; st.isSubsurface = false; st.matID = 0;
  Material& a = st.mat;
  a.albedo = vec3(m[0], m[1], m[2]); a.specular = m[3]; a.emission = vec3(0); a.anisotropy = m[4]; a.metallic = m[5]; a.roughness = m[6];
  a.subsurface = m[7]; a.specularTint = m[8]; a.sheen = m[9]; a.sheenTint = vec3(m[10], m[11], m[12]); a.clearcoat = m[13];
  a.clearcoatRoughness = m[14]; a.transmission = m[15]; a.ior = m[16]; a.attenuationColor = vec3(1); a.attenuationDistance = 1;
  a.ax = m[17]; a.ay = m[18]; a.f0 = vec3(m[19], m[20], m[21]); a.alpha = 1; a.unlit = false; a.thinwalled = thin != 0;
}
void ref_bsdf_eval(int pbrMode, const float* m, const float* N, const float* T, const float* B, float eta, int thin, const float* V, const float* L, float* f, float* pdf)
{
  State st;
  fill_state(st, m, N, T, B, eta, thin);
  rtxState.pbrMode = pbrMode;
  float p = 0.0f;
  vec3  r = Eval(st, vec3(V[0], V[1], V[2]), st.ffnormal, vec3(L[0], L[1], L[2]), p);
  f[0] = r.x; f[1] = r.y; f[2] = r.z; *pdf = p;
}
void ref_bsdf_sample(int pbrMode, const float* m, const float* N, const float* T, const float* B, float eta, int thin, const float* V, uint32_t* seed, float* L, float* f,
                     float* pdf)
{
  State st;
  fill_state(st, m, N, T, B, eta, thin);
  rtxState.pbrMode = pbrMode;
  float p = 0.0f;
  vec3  l(0);
  vec3  r = Sample(st, vec3(V[0], V[1], V[2]), st.ffnormal, l, p, *seed);
  L[0] = l.x; L[1] = l.y; L[2] = l.z; f[0] = r.x; f[1] = r.y; f[2] = r.z; *pdf = p;
}

}  // extern "C"
