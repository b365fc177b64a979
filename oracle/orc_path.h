// TEST INFRASTRUCTURE -- part of the CPU oracle (see oracle/README.md). Not linked into the product.
//
// Restatement of the per-pixel path tracer of the reference's compute / ray-query flavour:
//   shaders/pathtrace.comp:87-134   entry, seed, accumulate          -> render_pixel()
//   shaders/pathtrace.glsl          samplePixel / PathTrace / DirectLight / DebugInfo
//   shaders/traceray_rq.glsl        HitTest / ClosestHit / AnyHit (on the trace contract of orc_scene.h)
//   shaders/shade_state.glsl        GetShadeState
//   shaders/compress.glsl:142-180   decompress_unit_vec
//   shaders/gltf_material.glsl      GetMaterialsAndTextures / GetMetallicRoughness
//   shaders/punctual.glsl:28-51     range / spot attenuation
//   shaders/env_sampling.glsl       Environment_sample / EnvSample
//   shaders/common.glsl             GetSphericalUv / CreateCoordinateSystem / OffsetRay
#pragma once
#include <cstdio>
#include <cstdlib>
#include "orc_bsdf.h"
#include "orc_scene.h"
#include "orc_sky.h"

namespace orc {

struct Ray {
  vec3 origin, direction;
};

// shaders/globals.glsl:53-63
struct PtPayload {
  uint32_t seed = 0;
  float    hitT = 0;
  int      primitiveID = 0, instanceID = 0, instanceCustomIndex = 0;
  vec2     baryCoord;
  mat4x3   objectToWorld, worldToObject;
};

// ---- shaders/common.glsl ----------------------------------------------------------------------
// :67-74
inline vec2 GetSphericalUv(vec3 v)
{
  float gamma = masin(-v.y);
  float theta = matan2(v.z, v.x);
  return vec2(theta * M_1_OVER_PI * 0.5f + 0.5f, gamma * M_1_OVER_PI + 0.5f);
}
// :80-92 (identical body: shade_state.glsl:34-39 CreateTangent)
// ---- shaders/common.glsl:39-62 (heat-map palette) ------------------------------------------------
inline float fade(float low, float high, float value)
{
  float mid   = (low + high) * 0.5f;
  float range = (high - low) * 0.5f;
  float x     = 1.0f - gclamp(std::fabs(mid - value) / range, 0.0f, 1.0f);
  return gsmoothstep(0.0f, 1.0f, x);
}
inline vec3 temperature(float intensity)
{
  const vec3 blue(0.0f, 0.0f, 1.0f), cyan(0.0f, 1.0f, 1.0f), green(0.0f, 1.0f, 0.0f), yellow(1.0f, 1.0f, 0.0f), red(1.0f, 0.0f, 0.0f);
  return (((fade(-0.25f, 0.25f, intensity) * blue + fade(0.0f, 0.5f, intensity) * cyan) + fade(0.25f, 0.75f, intensity) * green) + fade(0.5f, 1.0f, intensity) * yellow)
         + gsmoothstep(0.75f, 1.0f, intensity) * red;
}

inline void CreateCoordinateSystem(vec3 N, vec3& Nt, vec3& Nb)
{
  Nt = normalize((std::fabs(N.z) > 0.99999f) ? vec3(-N.x * N.y, 1.0f - N.y * N.y, -N.y * N.z) : vec3(-N.x * N.z, -N.y * N.z, 1.0f - N.z * N.z));
  Nb = cross(Nt, N);
}
// :98-113
inline vec3 OffsetRay(vec3 p, vec3 n)
{
  const float intScale   = 256.0f;
  const float floatScale = 1.0f / 65536.0f;
  const float origin     = 1.0f / 32.0f;
  int32_t     of_i[3]    = {int32_t(intScale * n.x), int32_t(intScale * n.y), int32_t(intScale * n.z)};
  vec3        p_i(intBitsToFloat(floatBitsToInt(p.x) + ((p.x < 0) ? -of_i[0] : of_i[0])),
                  intBitsToFloat(floatBitsToInt(p.y) + ((p.y < 0) ? -of_i[1] : of_i[1])),
                  intBitsToFloat(floatBitsToInt(p.z) + ((p.z < 0) ? -of_i[2] : of_i[2])));
  return vec3(std::fabs(p.x) < origin ? p.x + floatScale * n.x : p_i.x,  //
              std::fabs(p.y) < origin ? p.y + floatScale * n.y : p_i.y,  //
              std::fabs(p.z) < origin ? p.z + floatScale * n.z : p_i.z);
}

// ---- shaders/compress.glsl:142-180 -----------------------------------------------------------
inline float short_to_floatm11(int v)
{
  return (v >= 0) ? (uintBitsToFloat(0x3F800000u | (uint32_t(v) << 8)) - 1.0f) : (uintBitsToFloat((0x80000000u | 0x3F800000u) | (uint32_t(-v) << 8)) + 1.0f);
}
inline vec3 decompress_unit_vec(uint32_t packed)
{
  if(packed != ~0u)
  {
    int       x     = int(packed & 0xFFFFu) - 32767;
    int       y     = int(packed >> 16) - 32767;
    const int maskx = x >> 31;
    const int masky = y >> 31;
    const int tmp0  = 32767 + maskx + masky;
    const int ymask = y ^ masky;
    const int tmp1  = tmp0 - (x ^ maskx);
    const int z     = tmp1 - ymask;
    float     zf;
    if(z < 0)
    {
      x  = (tmp0 - ymask) ^ maskx;
      y  = tmp1 ^ masky;
      zf = uintBitsToFloat((0x80000000u | 0x3F800000u) | (uint32_t(-z) << 8)) + 1.0f;
    }
    else
    {
      zf = uintBitsToFloat(0x3F800000u | (uint32_t(z) << 8)) - 1.0f;
    }
    return normalize(vec3(short_to_floatm11(x), short_to_floatm11(y), zf));
  }
  return vec3(3.402823466e+38f);
}
inline vec4 unpackUnorm4x8(uint32_t p)
{
  return vec4(float(p & 0xffu) / 255.0f, float((p >> 8) & 0xffu) / 255.0f, float((p >> 16) & 0xffu) / 255.0f, float(p >> 24) / 255.0f);
}

// ---- shaders/shade_state.glsl:43-53 ------------------------------------------------------------
struct ShadeState {
  vec3     normal, geom_normal, position;
  vec2     text_coords;
  vec3     tangent_u, tangent_v, color;
  uint32_t matIndex = 0;
};

struct VisibilityContribution {  // shaders/pathtrace.glsl:87-93
  vec3  radiance, lightDir;
  float lightDist = 0;
  bool  visible   = false;
};

// One of these per worker thread: the reference's global `prd` plus counters.
struct Tracer {
  const Scene&      sc;
  const pt_RtxState st;
  PtPayload         prd;
  Stats             stats;

  int               variant = 0;  // 0: ray-query / compute path, 1: RT-pipeline path (see below)

  Tracer(const Scene& s, const pt_RtxState& r, int variant_ = 0) : sc(s), st(r), variant(variant_) {}

  static vec2 uv_of(const pt_VertexAttributes& a) { return vec2(a.texcoord[0], a.texcoord[1]); }

  // ---- shaders/traceray_rq.glsl:32-102.  (instanceCustomIndex, primitiveID, bary) identify the candidate.
  bool HitTest(uint32_t w, float bu, float bv)
  {
    const WorldTri&             tr    = sc.tris[w];
    const pt_PrimMesh&          pinfo = sc.primMeshes[sc.nodes[tr.node].primMesh];
    const uint32_t              matIndex = (uint32_t)std::max(0, pinfo.materialIndex);
    const pt_GltfShadeMaterial& mat      = sc.materials[matIndex];
    stats.alphaTests++;

    float baseColorAlpha = mat.pbrBaseColorFactor[3];
    if(mat.pbrBaseColorTexture > -1)
    {
      const uint32_t*            tri   = &sc.indices[pinfo.firstIndex + 3 * tr.prim];
      const pt_VertexAttributes& attr0 = sc.vertices[pinfo.vertexOffset + tri[0]];
      const pt_VertexAttributes& attr1 = sc.vertices[pinfo.vertexOffset + tri[1]];
      const pt_VertexAttributes& attr2 = sc.vertices[pinfo.vertexOffset + tri[2]];
      const vec3                 bary(1.0f - bu - bv, bu, bv);
      // note: the handedness bit of v is NOT cleared here (Appendix C-8)
      vec2 texcoord0 = uv_of(attr0) * bary.x + uv_of(attr1) * bary.y + uv_of(attr2) * bary.z;
      vec4 t         = vec4(texcoord0.x, texcoord0.y, 1, 1) * mat4(mat.uvTransform);
      texcoord0      = vec2(t.x, t.y);
      baseColorAlpha *= sc.sample_texture(mat.pbrBaseColorTexture, texcoord0, &stats).w;
    }
    float opacity;
    if(mat.alphaMode == PT_ALPHA_MASK)
      opacity = baseColorAlpha > mat.alphaCutoff ? 1.0f : 0.0f;
    else
      opacity = baseColorAlpha;
    if(rnd(prd.seed) > opacity)
      return false;
    return true;
  }

  // ---- shaders/traceray_rq.glsl:108-147 on trace contract T5
  void ClosestHit(const Ray& r)
  {
    prd.hitT = INFINITY_RT;
    stats.closestRays++;
    float    tPrev = 0.0f;
    uint32_t wPrev = ~0u;
    bool     first = true;
    for(;;)
    {
      Candidate c = sc.query(r.origin, r.direction, INFINITY_RT, tPrev, wPrev, 0, first ? &stats : nullptr);
      first       = false;
      if(!c.found)
        return;
      const WorldTri& tr = sc.tris[c.w];
      if((tr.flags & TRI_OPAQUE) || HitTest(c.w, c.u, c.v))
      {
        prd.hitT                = c.t;
        prd.primitiveID         = (int)tr.prim;
        prd.instanceID          = (int)tr.node;
        prd.instanceCustomIndex = sc.nodes[tr.node].primMesh;
        prd.baryCoord           = vec2(c.u, c.v);
        prd.objectToWorld       = sc.objectToWorld[tr.node];
        prd.worldToObject       = sc.worldToObject[tr.node];
        return;
      }
      tPrev = c.t;
      wPrev = c.w;
    }
  }

  // RT-pipeline variant (shaders/traceray_rtx.glsl:52-73): the shadow payload carries a COPY of the path's seed
  // (`shadow_payload.seed = prd.seed`, :54-55) and pathtrace.rahit draws from the payload it is handed (:44,112), so the
  // stochastic alpha tests of a shadow ray do not advance prd.seed.
  bool AnyHit(const Ray& r, float maxDist)
  {
    const uint32_t saved = prd.seed;
    const bool     hit   = AnyHitRq(r, maxDist);
    if(variant == 1)
      prd.seed = saved;
    return hit;
  }

  // ---- shaders/traceray_rq.glsl:153-185 on trace contract T6
  // shaders/traceray_rq.glsl:153-185 on trace contract T6: candidates strictly in key order inside (0, maxDist); an opaque candidate commits
  // without a draw, a non-opaque one runs HitTest; the first commit ends the ray (gl_RayFlagsTerminateOnFirstHitEXT)
  bool AnyHitRq(const Ray& r, float maxDist)
  {
    stats.shadowRays++;
    const uint64_t n0 = stats.nodesVisited, t0 = stats.trisTested;
    float          tPrev = 0.0f;
    uint32_t       wPrev = ~0u;
    bool           first = true, found = false;
    for(;;)
    {
      Candidate c = sc.query(r.origin, r.direction, maxDist, tPrev, wPrev, 0, first ? &stats : nullptr);
      first       = false;
      if(!c.found)
        break;
      if((sc.tris[c.w].flags & TRI_OPAQUE) || HitTest(c.w, c.u, c.v))
      {
        found = true;
        break;
      }
      tPrev = c.t;
      wPrev = c.w;
    }
    stats.nodesShadow += stats.nodesVisited - n0;
    stats.trisShadow += stats.trisTested - t0;
    return found;
  }

  // ---- shaders/shade_state.glsl:63-145
  ShadeState GetShadeState(const PtPayload& hstate)
  {
    ShadeState         sstate;
    const pt_PrimMesh& geo  = sc.primMeshes[hstate.instanceCustomIndex];
    const uint32_t*    tri  = &sc.indices[geo.firstIndex + 3 * (uint32_t)hstate.primitiveID];
    const vec3         bary(1.0f - hstate.baryCoord.x - hstate.baryCoord.y, hstate.baryCoord.x, hstate.baryCoord.y);

    const pt_VertexAttributes& attr0 = sc.vertices[geo.vertexOffset + tri[0]];
    const pt_VertexAttributes& attr1 = sc.vertices[geo.vertexOffset + tri[1]];
    const pt_VertexAttributes& attr2 = sc.vertices[geo.vertexOffset + tri[2]];

    const uint32_t matIndex = (uint32_t)std::max(0, geo.materialIndex);

    const vec3 pos0(attr0.position[0], attr0.position[1], attr0.position[2]);
    const vec3 pos1(attr1.position[0], attr1.position[1], attr1.position[2]);
    const vec3 pos2(attr2.position[0], attr2.position[1], attr2.position[2]);
    const vec3 position       = pos0 * bary.x + pos1 * bary.y + pos2 * bary.z;
    const vec3 world_position = mul_point(hstate.objectToWorld, position);

    vec3 nrm0         = decompress_unit_vec(attr0.normal);
    vec3 nrm1         = decompress_unit_vec(attr1.normal);
    vec3 nrm2         = decompress_unit_vec(attr2.normal);
    vec3 normal       = normalize(nrm0 * bary.x + nrm1 * bary.y + nrm2 * bary.z);
    vec3 world_normal = normalize(mul_rowvec(normal, hstate.worldToObject));
    vec3 geom_normal  = normalize(cross(pos1 - pos0, pos2 - pos0));
    vec3 wgeom_normal = normalize(mul_rowvec(geom_normal, hstate.worldToObject));

    float h0 = (floatBitsToInt(attr0.texcoord[1]) & 1) == 1 ? 1.0f : -1.0f;
    // h1, h2 are computed by the reference but only vertex 0's handedness is used (:114)

    vec3 tng0           = decompress_unit_vec(attr0.tangent);
    vec3 tng1           = decompress_unit_vec(attr1.tangent);
    vec3 tng2           = decompress_unit_vec(attr2.tangent);
    vec3 tangent        = tng0 * bary.x + tng1 * bary.y + tng2 * bary.z;
    tangent             = normalize(tangent);
    vec3 world_tangent  = normalize(mul_dir(hstate.objectToWorld, tangent));
    world_tangent       = normalize(world_tangent - world_normal * dot(world_tangent, world_normal));
    vec3 world_binormal = cross(world_normal, world_tangent) * h0;

    auto decode_texture = [](const pt_VertexAttributes& a) { return vec2(a.texcoord[0], uintBitsToFloat(floatBitsToUint(a.texcoord[1]) & ~1u)); };
    const vec2 texcoord0 = decode_texture(attr0) * bary.x + decode_texture(attr1) * bary.y + decode_texture(attr2) * bary.z;

    const vec4 color = unpackUnorm4x8(attr0.color) * bary.x + unpackUnorm4x8(attr1.color) * bary.y + unpackUnorm4x8(attr2.color) * bary.z;

    sstate.normal      = world_normal;
    sstate.geom_normal = wgeom_normal;
    sstate.position    = world_position;
    sstate.text_coords = texcoord0;
    sstate.tangent_u   = world_tangent;
    sstate.tangent_v   = world_binormal;
    sstate.color       = color.xyz();
    sstate.matIndex    = matIndex;
    if(dot(sstate.normal, sstate.geom_normal) <= 0)
      sstate.normal *= -1.0f;
    return sstate;
  }

  // ---- shaders/gltf_material.glsl:37-46 (SRGB_FAST_APPROXIMATION)
  static vec4 SRGBtoLINEAR(vec4 srgbIn)
  {
    vec3 lin = gpow(srgbIn.xyz(), vec3(2.2f));
    return vec4(lin, srgbIn.w);
  }
  vec4 tex(int id, vec2 uv) { return sc.sample_texture(id, uv, &stats); }

  // ---- shaders/gltf_material.glsl:52-93
  void GetMetallicRoughness(State& state, const pt_GltfShadeMaterial& material)
  {
    float dielectricSpecular = (material.ior - 1) / (material.ior + 1);
    dielectricSpecular *= dielectricSpecular;

    float perceptualRoughness = material.pbrRoughnessFactor;
    float metallic            = material.pbrMetallicFactor;
    if(material.pbrMetallicRoughnessTexture > -1)
    {
      vec4 mrSample       = tex(material.pbrMetallicRoughnessTexture, state.texCoord);
      perceptualRoughness = mrSample.y * perceptualRoughness;
      metallic            = mrSample.z * metallic;
    }
    vec4 baseColor(material.pbrBaseColorFactor[0], material.pbrBaseColorFactor[1], material.pbrBaseColorFactor[2], material.pbrBaseColorFactor[3]);
    if(material.pbrBaseColorTexture > -1)
      baseColor = baseColor * SRGBtoLINEAR(tex(material.pbrBaseColorTexture, state.texCoord));

    vec3 f0 = gmix(vec3(dielectricSpecular), baseColor.xyz(), metallic);

    state.mat.albedo    = baseColor.xyz();
    state.mat.metallic  = metallic;
    state.mat.roughness = perceptualRoughness;
    state.mat.f0        = f0;
    state.mat.alpha     = baseColor.w;
  }

  // ---- shaders/gltf_material.glsl:104-193
  void GetMaterialsAndTextures(State& state, const Ray& r)
  {
    const pt_GltfShadeMaterial& material = sc.materials[state.matID];

    state.mat.specular     = 0.5f;
    state.mat.subsurface   = 0;
    state.mat.specularTint = 1;
    state.mat.sheen        = 0;
    state.mat.sheenTint    = vec3(0);

    vec4 tc        = vec4(state.texCoord.x, state.texCoord.y, 1, 1) * mat4(material.uvTransform);
    state.texCoord = vec2(tc.x, tc.y);
    vec3 T = state.tangent, B = state.bitangent, Nn = state.normal;  // mat3 TBN columns

    if(material.normalTexture > -1)
    {
      vec3 normalVector = tex(material.normalTexture, state.texCoord).xyz();
      normalVector      = normalize(normalVector * 2.0f - 1.0f);
      normalVector      = normalVector * vec3(material.normalTextureScale, material.normalTextureScale, 1.0f);
      state.normal      = normalize(mul_mat3(T, B, Nn, normalVector));
      state.ffnormal    = dot(state.normal, r.direction) <= 0.0f ? state.normal : -state.normal;
      CreateCoordinateSystem(state.ffnormal, state.tangent, state.bitangent);
    }

    state.mat.emission = vec3(material.emissiveFactor[0], material.emissiveFactor[1], material.emissiveFactor[2]);
    if(material.emissiveTexture > -1)
      state.mat.emission *= SRGBtoLINEAR(tex(material.emissiveTexture, state.texCoord)).xyz();

    GetMetallicRoughness(state, material);

    state.mat.roughness = gmax(state.mat.roughness, 0.001f);

    state.mat.transmission = material.transmissionFactor;
    if(material.transmissionTexture > -1)
      state.mat.transmission *= tex(material.transmissionTexture, state.texCoord).x;

    state.mat.ior = material.ior;
    state.eta     = dot(state.normal, state.ffnormal) > 0.0f ? (1.0f / state.mat.ior) : state.mat.ior;

    state.mat.unlit = (material.unlit == 1);

    state.mat.anisotropy = material.anisotropy;
    float aspect         = std::sqrt(1.0f - material.anisotropy * 0.9f);
    state.mat.ax         = gmax(0.001f, state.mat.roughness / aspect);
    state.mat.ay         = gmax(0.001f, state.mat.roughness * aspect);

    if(material.anisotropy > 0)
    {
      // TBN is the matrix captured BEFORE normal mapping (:116)
      state.tangent   = normalize(mul_mat3(T, B, Nn, vec3(material.anisotropyDirection[0], material.anisotropyDirection[1], material.anisotropyDirection[2])));
      state.bitangent = normalize(cross(state.normal, state.tangent));
    }

    state.mat.attenuationColor    = vec3(material.attenuationColor[0], material.attenuationColor[1], material.attenuationColor[2]);
    state.mat.attenuationDistance = material.attenuationDistance;
    state.mat.thinwalled          = material.thicknessFactor == 0;

    state.mat.clearcoat          = material.clearcoatFactor;
    state.mat.clearcoatRoughness = material.clearcoatRoughness;
    if(material.clearcoatTexture > -1)
      state.mat.clearcoat *= tex(material.clearcoatTexture, state.texCoord).x;
    if(material.clearcoatRoughnessTexture > -1)
      state.mat.clearcoatRoughness *= tex(material.clearcoatRoughnessTexture, state.texCoord).y;
    state.mat.clearcoatRoughness = gmax(state.mat.clearcoatRoughness, 0.001f);

    vec4 sheen          = unpackUnorm4x8(material.sheen);
    state.mat.sheenTint = sheen.xyz();
    state.mat.sheen     = sheen.w;
  }

  // ---- shaders/punctual.glsl:28-51
  static float getRangeAttenuation(float range, float distance)
  {
    if(range <= 0.0f)
      return 1.0f;
    return gmax(gmin(1.0f - mpow(distance / range, 4.0f), 1.0f), 0.0f) / mpow(distance, 2.0f);
  }
  static float getSpotAttenuation(vec3 pointToLight, vec3 spotDirection, float outerConeCos, float innerConeCos)
  {
    float actualCos = dot(normalize(spotDirection), normalize(-pointToLight));
    if(actualCos > outerConeCos)
    {
      if(actualCos < innerConeCos)
        return gsmoothstep(outerConeCos, innerConeCos, actualCos);
      return 1.0f;
    }
    return 0.0f;
  }

  // ---- shaders/env_sampling.glsl:38-99
  vec3 Environment_sample(vec3 randVal, vec3& to_light, float& pdf)
  {
    vec3           xi     = randVal;
    const uint32_t width  = (uint32_t)sc.envW;
    const uint32_t height = (uint32_t)sc.envH;
    const uint32_t size   = width * height;
    const uint32_t idx    = std::min(uint32_t(xi.x * float(size)), size - 1);
    stats.neeLookups++;

    const pt_EnvAccel& sample_data = sc.envAccel[idx];
    uint32_t           env_idx;
    if(xi.y < sample_data.q)
    {
      env_idx = idx;
      xi.y /= sample_data.q;
      pdf = sample_data.pdf;
    }
    else
    {
      env_idx = sample_data.alias;
      xi.y    = (xi.y - sample_data.q) / (1.0f - sample_data.q);
      pdf     = sample_data.aliasPdf;
    }
    const uint32_t px = env_idx % width;
    uint32_t       py = env_idx / width;

    const float u       = (float(px) + xi.y) / float(width);  // float(px + xi.y): uint + float promotes to float
    const float phi     = u * (2.0f * M_PI_F) - M_PI_F;
    float       sin_phi = msin(phi);
    float       cos_phi = mcos(phi);

    const float step_theta = M_PI_F / float(height);
    const float theta0     = float(py) * step_theta;
    const float cos_theta  = mcos(theta0) * (1.0f - xi.z) + mcos(theta0 + step_theta) * xi.z;
    const float theta      = macos(cos_theta);
    const float sin_theta  = msin(theta);
    const float v          = theta * M_1_OVER_PI;

    to_light = vec3(cos_phi * sin_theta, cos_theta, sin_phi * sin_theta);
    return sc.sample_env(vec2(u, v));
  }

  // ---- shaders/env_sampling.glsl:105-135
  vec4 EnvSample(vec3& radiance)
  {
    vec3  lightDir;
    float pdf;
    if(sc.sunsky.in_use == 1)
    {
      float sun_radius = (0.00465f * 10.0f) * sc.sunsky.sun_disk_scale;
      vec3  sd(sc.sunsky.sun_direction[0], sc.sunsky.sun_direction[1], sc.sunsky.sun_direction[2]);
      vec3  T, B;
      CreateCoordinateSystem(sd, T, B);
      vec3 dir;
      dir.x    = rnd(prd.seed) * sun_radius;
      dir.y    = rnd(prd.seed) * sun_radius;
      dir.z    = std::sqrt(gmax(0.0f, 1.0f - dir.x * dir.x - dir.y * dir.y));
      lightDir = normalize(T * dir.x + B * dir.y + sd * dir.z);
      radiance = sky::sun_and_sky(sc.sunsky, lightDir);
      pdf      = 0.5f;
    }
    else
    {
      float a = rnd(prd.seed);
      float b = rnd(prd.seed);
      float c = rnd(prd.seed);
      radiance = Environment_sample(vec3(a, b, c), lightDir, pdf);
    }
    radiance *= st.hdrMultiplier;
    return vec4(lightDir, pdf);
  }

  vec3 Eval(const State& state, vec3 V, vec3 N, vec3 L, float& pdf)  // pathtrace.glsl:40-46
  {
    return st.pbrMode == 0 ? DisneyEval(state, V, N, L, pdf) : PbrEval(state, V, N, L, pdf);
  }
  vec3 Sample(const State& state, vec3 V, vec3 N, vec3& L, float& pdf, uint32_t& seed)  // pathtrace.glsl:50-56
  {
    return st.pbrMode == 0 ? DisneySample(state, V, N, L, pdf, seed) : PbrSample(state, V, N, L, pdf, seed);
  }

  // ---- shaders/pathtrace.glsl:61-83
  vec3 DebugInfo(const State& state)
  {
    switch(st.debugging_mode)
    {
      case PT_DEBUG_METALLIC: return vec3(state.mat.metallic);
      case PT_DEBUG_NORMAL: return (state.normal + vec3(1)) * .5f;
      case PT_DEBUG_BASECOLOR: return state.mat.albedo;
      case PT_DEBUG_EMISSIVE: return state.mat.emission;
      case PT_DEBUG_ALPHA: return vec3(state.mat.alpha);
      case PT_DEBUG_ROUGHNESS: return vec3(state.mat.roughness);
      case PT_DEBUG_TEXCOORD: return vec3(state.texCoord.x, state.texCoord.y, 0);
      case PT_DEBUG_TANGENT: return (state.tangent + vec3(1)) * .5f;
    }
    return vec3(1000, 0, 0);
  }

  // ---- shaders/pathtrace.glsl:97-188
  VisibilityContribution DirectLight(const Ray& r, const State& state)
  {
    vec3  Li(0);
    float lightPdf;
    vec3  lightContrib;
    vec3  lightDir;
    float lightDist = 1e32f;
    bool  isLight   = false;

    VisibilityContribution contrib;
    contrib.radiance = vec3(0);
    contrib.visible  = false;

    float p_select_light = st.hdrMultiplier > 0.0f ? 0.5f : 1.0f;

    if(sc.camera.nbLights != 0 && rnd(prd.seed) <= p_select_light)
    {
      isLight = true;
      int light_index = int(gmin(rnd(prd.seed) * float(sc.camera.nbLights), float(sc.camera.nbLights)));
      // the reference indexes lights[nbLights] when the draw rounds up to nbLights (never happens: rand < 1 and
      // nbLights <= 2^23); clamp for memory safety only
      light_index           = std::min(light_index, (int)sc.lights.size() - 1);
      const pt_Light& light = sc.lights[light_index];
      vec3  ldir(light.direction[0], light.direction[1], light.direction[2]);
      vec3  pointToLight     = -ldir;
      float rangeAttenuation = 1.0f;
      float spotAttenuation  = 1.0f;
      if(light.type != PT_LIGHT_DIRECTIONAL)
        pointToLight = vec3(light.position[0], light.position[1], light.position[2]) - state.position;
      lightDist = length(pointToLight);
      if(light.type != PT_LIGHT_DIRECTIONAL)
        rangeAttenuation = getRangeAttenuation(light.range, lightDist);
      if(light.type == PT_LIGHT_SPOT)
        spotAttenuation = getSpotAttenuation(pointToLight, ldir, light.outerConeCos, light.innerConeCos);
      vec3 intensity = vec3(light.color[0], light.color[1], light.color[2]) * (rangeAttenuation * spotAttenuation * light.intensity);
      lightContrib   = intensity;
      lightDir       = normalize(pointToLight);
      lightPdf       = 1.0f;
    }
    else
    {
      vec4 dirPdf = EnvSample(lightContrib);
      lightDir    = dirPdf.xyz();
      lightPdf    = dirPdf.w;
    }

    if(state.isSubsurface || dot(lightDir, state.ffnormal) > 0.0f)
    {
      float bsdfPdf   = 0.0f;
      vec3  f         = Eval(state, -r.direction, state.ffnormal, lightDir, bsdfPdf);
      float misWeight = isLight ? 1.0f : gmax(0.0f, powerHeuristic(lightPdf, bsdfPdf));
      Li += f * misWeight * std::fabs(dot(lightDir, state.ffnormal)) * lightContrib / lightPdf;

      contrib.visible   = true;
      contrib.lightDir  = lightDir;
      contrib.lightDist = lightDist;
      contrib.radiance  = Li;
    }
    return contrib;
  }

  vec3 env_lookup(vec3 dir)  // pathtrace.glsl:218-225
  {
    stats.misses++;
    if(sc.sunsky.in_use == 1)
      return sky::sun_and_sky(sc.sunsky, dir);
    return sc.sample_env(GetSphericalUv(dir));
  }

  // ---- shaders/pathtrace.glsl:193-343
  vec3 PathTrace(Ray r)
  {
    vec3 radiance(0.0f), throughput(1.0f), absorption(0.0f);

    static const bool traceOn = std::getenv("ORC_TRACE") != nullptr;  // diagnostics: the rays and hits of every path rendered (use with a pixel list)
    for(int depth = 0; depth < st.maxDepth; depth++)
    {
      ClosestHit(r);
      if(traceOn)
        std::fprintf(stderr, "ORC_TRACE frame %d depth %d o %.9g %.9g %.9g d %.9g %.9g %.9g t %.9g inst %d prim %d uv %.9g %.9g\n", st.frame, depth, r.origin.x, r.origin.y, r.origin.z,
                     r.direction.x, r.direction.y, r.direction.z, prd.hitT, prd.hitT == INFINITY_RT ? -1 : prd.instanceID, prd.hitT == INFINITY_RT ? -1 : prd.primitiveID, prd.baryCoord.x,
                     prd.baryCoord.y);

      if(prd.hitT == INFINITY_RT)
      {
        if(st.debugging_mode != PT_DEBUG_NONE)
        {
          if(depth != st.maxDepth - 1)
            return vec3(0);
          if(st.debugging_mode == PT_DEBUG_RADIANCE)
            return radiance;
          else if(st.debugging_mode == PT_DEBUG_WEIGHT)
            return throughput;
          else if(st.debugging_mode == PT_DEBUG_RAYDIR)
            return (r.direction + vec3(1)) * 0.5f;
        }
        vec3 env = env_lookup(r.direction);
        return radiance + (env * st.hdrMultiplier * throughput);
      }

      ShadeState sstate = GetShadeState(prd);
      stats.shadedHits++;

      State state;
      state.position       = sstate.position;
      state.normal         = sstate.normal;
      state.tangent        = sstate.tangent_u;
      state.bitangent      = sstate.tangent_v;
      state.texCoord       = sstate.text_coords;
      state.matID          = sstate.matIndex;
      state.isEmitter      = false;
      state.specularBounce = false;
      state.isSubsurface   = false;
      state.ffnormal       = dot(state.normal, r.direction) <= 0.0f ? state.normal : -state.normal;

      GetMaterialsAndTextures(state, r);
      state.mat.albedo *= sstate.color;

      if(st.debugging_mode != PT_DEBUG_NONE && st.debugging_mode < PT_DEBUG_RADIANCE)
        return DebugInfo(state);

      if(state.mat.unlit)
        return radiance + state.mat.albedo * throughput;

      if(dot(state.normal, state.ffnormal) > 0.0f)
        absorption = vec3(0.0f);

      radiance += state.mat.emission * throughput;
      throughput *= gexp(-absorption * prd.hitT);

      VisibilityContribution vcontrib = DirectLight(r, state);
      vcontrib.radiance *= throughput;

      vec3  bsdfL;
      float bsdfPdf = 0.0f;
      vec3  bsdfF   = Sample(state, -r.direction, state.ffnormal, bsdfL, bsdfPdf, prd.seed);

      if(dot(state.ffnormal, bsdfL) < 0.0f)
        absorption = -glog(state.mat.attenuationColor) / vec3(state.mat.attenuationDistance);

      if(bsdfPdf > 0.0f)
        throughput *= bsdfF * std::fabs(dot(state.ffnormal, bsdfL)) / bsdfPdf;
      else
        break;

      if(st.debugging_mode != PT_DEBUG_NONE && (depth == st.maxDepth - 1))
      {
        if(st.debugging_mode == PT_DEBUG_RADIANCE)
          return vcontrib.radiance;
        else if(st.debugging_mode == PT_DEBUG_WEIGHT)
          return throughput;
        else if(st.debugging_mode == PT_DEBUG_RAYDIR)
          return (bsdfL + vec3(1)) * 0.5f;
      }

      // RR (RR_DEPTH 0)
      float rrPcont = gmin(gmax(throughput.x, gmax(throughput.y, throughput.z)) * state.eta * state.eta + 0.001f, 0.95f);

      r.direction = bsdfL;
      r.origin    = OffsetRay(sstate.position, dot(bsdfL, state.ffnormal) > 0 ? state.ffnormal : -state.ffnormal);

      if(vcontrib.visible == true)
      {
        Ray  shadowRay{r.origin, vcontrib.lightDir};
        bool inShadow = AnyHit(shadowRay, vcontrib.lightDist);
        if(traceOn)
          std::fprintf(stderr, "ORC_TRACE   shadow o %.9g %.9g %.9g d %.9g %.9g %.9g dist %.9g inShadow %d contrib %.9g %.9g %.9g\n", shadowRay.origin.x, shadowRay.origin.y, shadowRay.origin.z,
                       shadowRay.direction.x, shadowRay.direction.y, shadowRay.direction.z, vcontrib.lightDist, int(inShadow), vcontrib.radiance.x, vcontrib.radiance.y, vcontrib.radiance.z);
        if(!inShadow)
          radiance += vcontrib.radiance;
      }

      if(rnd(prd.seed) >= rrPcont)
        break;
      throughput /= rrPcont;
    }
    return radiance;
  }

  // ---- shaders/pathtrace.glsl:348-387
  vec3 samplePixel(int px, int py, int sizeX, int sizeY)
  {
    vec2 subpixel_jitter;
    if(st.frame == 0)
      subpixel_jitter = vec2(0.5f, 0.5f);
    else
    {
      float jx        = rnd(prd.seed);
      float jy        = rnd(prd.seed);
      subpixel_jitter = vec2(jx, jy);
    }
    const vec2 pixelCenter = vec2(float(px), float(py)) + subpixel_jitter;
    const vec2 inUV        = pixelCenter / vec2(float(sizeX), float(sizeY));
    vec2       d           = inUV * 2.0f - vec2(1.0f);

    mat4 viewInverse(sc.camera.viewInverse), projInverse(sc.camera.projInverse);
    vec4 origin    = viewInverse * vec4(0, 0, 0, 1);
    vec4 target    = projInverse * vec4(d.x, d.y, 1, 1);
    vec4 direction = viewInverse * vec4(normalize(target.xyz()), 0);

    vec3  focalPoint        = direction.xyz() * sc.camera.focalDist;
    float cam_r1            = rnd(prd.seed) * M_TWO_PI;
    float cam_r2            = rnd(prd.seed) * sc.camera.aperture;
    vec4  cam_right         = viewInverse * vec4(1, 0, 0, 0);
    vec4  cam_up            = viewInverse * vec4(0, 1, 0, 0);
    vec3  randomAperturePos = (cam_right.xyz() * mcos(cam_r1) + cam_up.xyz() * msin(cam_r1)) * std::sqrt(cam_r2);
    vec3  finalRayDir       = normalize(focalPoint - randomAperturePos);

    Ray ray{origin.xyz() + randomAperturePos, finalRayDir};

    vec3 radiance = PathTrace(ray);

    float lum = dot(radiance, vec3(0.212671f, 0.715160f, 0.072169f));
    if(lum > st.fireflyClampThreshold)
      radiance *= st.fireflyClampThreshold / lum;
    stats.samples++;
    return radiance;
  }

  // ---- shaders/pathtrace.comp:87-134 (heat-map debug mode is not reproduced: it is a wall-clock read)
  void render_pixel(int px, int py, float* rgba)
  {
    // pathtrace.comp:97 seeds with frame * maxSamples, pathtrace.rgen:72 (initRandom) with the frame alone
    prd.seed = tea(uint32_t(st.size[0]) * uint32_t(py) + uint32_t(px), uint32_t(variant == 1 ? st.frame : st.frame * st.maxSamples));
    // pathtrace.comp:89 `start = clockRealtimeEXT()`: real time has no CPU restatement; the oracle's clock ticks once per BVH node visited and
    // once per triangle tested (deterministic, and proportional to where the time goes)
    const uint64_t start = stats.nodesVisited + stats.trisTested;
    vec3 pixelColor(0);
    for(int smpl = 0; smpl < st.maxSamples; ++smpl)
      pixelColor += samplePixel(px, py, st.size[0], st.size[1]);
    pixelColor /= float(st.maxSamples);
    if(st.debugging_mode == 12)  // eHeatmap, pathtrace.comp:108-119
    {
      const uint64_t end  = stats.nodesVisited + stats.trisTested;
      const float    low  = float(st.minHeatmap), high = float(st.maxHeatmap);
      const float    val  = gclamp((float(end - start) - low) / (high - low), 0.0f, 1.0f);
      pixelColor          = temperature(val);
    }

    if(st.frame > 0)
    {
      vec3 old_color(rgba[0], rgba[1], rgba[2]);
      vec3 new_result = gmix(old_color, pixelColor, 1.0f / float(st.frame + 1));
      rgba[0] = new_result.x; rgba[1] = new_result.y; rgba[2] = new_result.z; rgba[3] = 1.f;
    }
    else
    {
      rgba[0] = pixelColor.x; rgba[1] = pixelColor.y; rgba[2] = pixelColor.z; rgba[3] = 1.f;
    }
  }
};

}  // namespace orc
