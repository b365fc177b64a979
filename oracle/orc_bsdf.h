// TEST INFRASTRUCTURE -- part of the CPU oracle (see oracle/README.md). Not linked into the product.
//
// Restatement of shaders/random.glsl, shaders/globals.glsl (Material/State), shaders/pbr_disney.glsl
// and shaders/pbr_gltf.glsl.  Function names follow the reference so that the file:line citations
// can be checked one to one.  Known quirks of the reference are kept.
#pragma once
#include "glsl_math.h"

namespace orc {

// shaders/globals.glsl:27-43.  PI/TWO_PI are macros there (float literals in GLSL), M_* are consts.
static const float PI          = 3.14159265358979323f;
static const float TWO_PI      = 6.28318530717958648f;
static const float INFINITY_RT = 1e32f;
static const float M_PI_F      = 3.14159265358979323846f;
static const float M_TWO_PI    = 6.28318530717958648f;
static const float M_1_OVER_PI = 0.318309886183790671538f;

// ---- shaders/random.glsl -------------------------------------------------------------------
// :34-48
inline uint32_t tea(uint32_t val0, uint32_t val1)
{
  uint32_t v0 = val0, v1 = val1, s0 = 0;
  for(uint32_t n = 0; n < 16; n++)
  {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}
// :59-65
inline uint32_t pcg(uint32_t& state)
{
  uint32_t prev = state * 747796405u + 2891336453u;
  uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
  state         = prev;
  return (word >> 22u) ^ word;
}
// :82-93
inline void pcg3d(uint32_t v[3])
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
}
// :98-102
inline float rnd(uint32_t& seed)
{
  uint32_t r = pcg(seed);
  return uintBitsToFloat(0x3f800000u | (r >> 9)) - 1.0f;
}

// ---- shaders/globals.glsl:67-123 -----------------------------------------------------------
struct Material {
  vec3  albedo;
  float specular = 0;
  vec3  emission;
  float anisotropy = 0, metallic = 0, roughness = 0, subsurface = 0, specularTint = 0, sheen = 0;
  vec3  sheenTint;
  float clearcoat = 0, clearcoatRoughness = 0, transmission = 0, ior = 0;
  vec3  attenuationColor;
  float attenuationDistance = 0;
  float ax = 0, ay = 0;
  vec3  f0;
  float alpha = 0;
  bool  unlit = false, thinwalled = false;
};
struct State {
  int      depth = 0;
  float    eta   = 0;
  vec3     position, normal, ffnormal, tangent, bitangent;
  vec2     texCoord;
  bool     isEmitter = false, specularBounce = false, isSubsurface = false;
  uint32_t matID = 0;
  Material mat;
};

// ================================ shaders/pbr_disney.glsl ===================================
// :68-81 (r2 is ignored by the reference: Appendix C-5)
inline vec3 ImportanceSampleGTR1(float rgh, float r1, float /*r2*/)
{
  float a        = gmax(0.001f, rgh);
  float a2       = a * a;
  float phi      = r1 * TWO_PI;
  float cosTheta = std::sqrt((1.0f - mpow(a2, 1.0f - r1)) / (1.0f - a2));
  float sinTheta = gclamp(std::sqrt(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
  float sinPhi   = msin(phi);
  float cosPhi   = mcos(phi);
  return vec3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
// :85-94
inline vec3 ImportanceSampleGTR2_aniso(float ax, float ay, float r1, float r2)
{
  float phi      = r1 * TWO_PI;
  float sinPhi   = ay * msin(phi);
  float cosPhi   = ax * mcos(phi);
  float tanTheta = std::sqrt(r2 / (1 - r2));
  return vec3(tanTheta * cosPhi, tanTheta * sinPhi, 1.0f);
}
// :98-110
inline vec3 ImportanceSampleGTR2(float rgh, float r1, float r2)
{
  float a        = gmax(0.001f, rgh);
  float phi      = r1 * TWO_PI;
  float cosTheta = std::sqrt((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
  float sinTheta = gclamp(std::sqrt(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
  float sinPhi   = msin(phi);
  float cosPhi   = mcos(phi);
  return vec3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
// :114-119
inline float SchlickFresnel(float u)
{
  float m  = gclamp(1.0f - u, 0.0f, 1.0f);
  float m2 = m * m;
  return m2 * m2 * m;
}
// :123-137
inline float DielectricFresnel(float cos_theta_i, float eta)
{
  float sinThetaTSq = eta * eta * (1.0f - cos_theta_i * cos_theta_i);
  if(sinThetaTSq > 1.0f)
    return 1.0f;
  float cos_theta_t = std::sqrt(gmax(1.0f - sinThetaTSq, 0.0f));
  float rs          = (eta * cos_theta_t - cos_theta_i) / (eta * cos_theta_t + cos_theta_i);
  float rp          = (eta * cos_theta_i - cos_theta_t) / (eta * cos_theta_i + cos_theta_t);
  return 0.5f * (rs * rs + rp * rp);
}
// :141-148
inline float GTR1(float NdotH, float a)
{
  if(a >= 1.0f)
    return M_1_OVER_PI;
  float a2 = a * a;
  float t  = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
  return (a2 - 1.0f) / (PI * mlog(a2) * t);
}
// :152-157
inline float GTR2(float NdotH, float a)
{
  float a2 = a * a;
  float t  = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
  return a2 / (PI * t * t);
}
// :161-167
inline float GTR2_aniso(float NdotH, float HdotX, float HdotY, float ax, float ay)
{
  float a = HdotX / ax;
  float b = HdotY / ay;
  float c = a * a + b * b + NdotH * NdotH;
  return 1.0f / (PI * ax * ay * c * c);
}
// :171-176
inline float SmithG_GGX(float NdotV, float alphaG)
{
  float a = alphaG * alphaG;
  float b = NdotV * NdotV;
  return 1.0f / (NdotV + std::sqrt(a + b - a * b));
}
// :180-186
inline float SmithG_GGX_aniso(float NdotV, float VdotX, float VdotY, float ax, float ay)
{
  float a = VdotX * ax;
  float b = VdotY * ay;
  float c = NdotV;
  return 1.0f / (NdotV + std::sqrt(a * a + b * b + c * c));
}
// :190-200
inline vec3 CosineSampleHemisphere(float r1, float r2)
{
  vec3  dir;
  float r   = std::sqrt(r1);
  float phi = TWO_PI * r2;
  dir.x     = r * mcos(phi);
  dir.y     = r * msin(phi);
  dir.z     = std::sqrt(gmax(0.0f, 1.0f - dir.x * dir.x - dir.y * dir.y));
  return dir;
}
// :204-210
inline vec3 UniformSampleHemisphere(float r1, float r2)
{
  float r   = std::sqrt(gmax(0.0f, 1.0f - r1 * r1));
  float phi = TWO_PI * r2;
  return vec3(r * mcos(phi), r * msin(phi), r1);
}
// :225-230
inline float powerHeuristic(float a, float b)
{
  float t = a * a;
  return t / (b * b + t);
}
// :320-333
inline vec3 EvalDielectricReflection(const State& state, vec3 V, vec3 N, vec3 L, vec3 H, float& pdf)
{
  if(dot(N, L) < 0.0f)
    return vec3(0.0f);
  float F = DielectricFresnel(dot(V, H), state.eta);
  float D = GTR2(dot(N, H), state.mat.roughness);
  pdf     = D * dot(N, H) * F / (4.0f * dot(V, H));
  float G = SmithG_GGX(std::fabs(dot(N, L)), state.mat.roughness) * SmithG_GGX(dot(N, V), state.mat.roughness);
  return state.mat.albedo * F * D * G;
}
// :337-349
inline vec3 EvalDielectricRefraction(const State& state, vec3 V, vec3 N, vec3 L, vec3 H, float& pdf)
{
  float F         = DielectricFresnel(std::fabs(dot(V, H)), state.eta);
  float D         = GTR2(dot(N, H), state.mat.roughness);
  float denomSqrt = dot(L, H) * state.eta + dot(V, H);
  pdf             = D * dot(N, H) * (1.0f - F) * std::fabs(dot(L, H)) / (denomSqrt * denomSqrt);
  float G = SmithG_GGX(std::fabs(dot(N, L)), state.mat.roughness) * SmithG_GGX(dot(N, V), state.mat.roughness);
  return state.mat.albedo * (1.0f - F) * D * G * std::fabs(dot(V, H)) * std::fabs(dot(L, H)) * 4.0f * state.eta * state.eta / (denomSqrt * denomSqrt);
}
// :353-367
inline vec3 EvalSpecular(const State& state, vec3 Cspec0, vec3 V, vec3 N, vec3 L, vec3 H, float& pdf)
{
  if(dot(N, L) < 0.0f)
    return vec3(0.0f);
  float D  = GTR2_aniso(dot(N, H), dot(H, state.tangent), dot(H, state.bitangent), state.mat.ax, state.mat.ay);
  pdf      = D * dot(N, H) / (4.0f * dot(V, H));
  float FH = SchlickFresnel(dot(L, H));
  vec3  F  = gmix(Cspec0, vec3(1.0f), FH);
  float G  = SmithG_GGX_aniso(dot(N, L), dot(L, state.tangent), dot(L, state.bitangent), state.mat.ax, state.mat.ay);
  G *= SmithG_GGX_aniso(dot(N, V), dot(V, state.tangent), dot(V, state.bitangent), state.mat.ax, state.mat.ay);
  return F * D * G;
}
// :371-383 (Smith alpha hard-coded to 0.25: Appendix C-5)
inline vec3 EvalClearcoat(const State& state, vec3 V, vec3 N, vec3 L, vec3 H, float& pdf)
{
  if(dot(N, L) < 0.0f)
    return vec3(0.0f);
  float D  = GTR1(dot(N, H), state.mat.clearcoatRoughness);
  pdf      = D * dot(N, H) / (4.0f * dot(V, H));
  float FH = SchlickFresnel(dot(L, H));
  float F  = gmix(0.04f, 1.0f, FH);
  float G  = SmithG_GGX(dot(N, L), 0.25f) * SmithG_GGX(dot(N, V), 0.25f);
  return vec3(0.25f * state.mat.clearcoat * F * D * G);
}
// :387-401
inline vec3 EvalDiffuse(const State& state, vec3 Csheen, vec3 V, vec3 N, vec3 L, vec3 H, float& pdf)
{
  if(dot(N, L) < 0.0f)
    return vec3(0.0f);
  pdf          = dot(N, L) * (1.0f / PI);
  float FL     = SchlickFresnel(dot(N, L));
  float FV     = SchlickFresnel(dot(N, V));
  float FH     = SchlickFresnel(dot(L, H));
  float Fd90   = 0.5f + 2.0f * dot(L, H) * dot(L, H) * state.mat.roughness;
  float Fd     = gmix(1.0f, Fd90, FL) * gmix(1.0f, Fd90, FV);
  vec3  Fsheen = Csheen * (FH * state.mat.sheen);
  return ((1.0f / PI) * Fd * (1.0f - state.mat.subsurface) * state.mat.albedo + Fsheen) * (1.0f - state.mat.metallic);
}
// :405-413
inline vec3 EvalSubsurface(const State& state, vec3 V, vec3 N, vec3 L, float& pdf)
{
  pdf      = (1.0f / TWO_PI);
  float FL = SchlickFresnel(std::fabs(dot(N, L)));
  float FV = SchlickFresnel(dot(N, V));
  float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
  return gsqrt(state.mat.albedo) * state.mat.subsurface * (1.0f / PI) * Fd * (1.0f - state.mat.metallic) * (1.0f - state.mat.transmission);
}

// :417-521.  `state` is a by-value copy: pathtrace.glsl:50 takes `in State`, so the writes to
// state.eta / state.isSubsurface never reach the integrator.
inline vec3 DisneySample(State state, vec3 V, vec3 N, vec3& L, float& pdf, uint32_t& seed)
{
  state.isSubsurface = false;
  pdf                = 0.0f;
  vec3 f(0.0f);

  float r1 = rnd(seed);
  float r2 = rnd(seed);

  float diffuseRatio = 0.5f * (1.0f - state.mat.metallic);
  float transWeight  = (1.0f - state.mat.metallic) * state.mat.transmission;

  vec3  Cdlin  = state.mat.albedo;
  float Cdlum  = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;
  vec3  Ctint  = Cdlum > 0.0f ? Cdlin / Cdlum : vec3(1.0f);
  vec3  Cspec0 = gmix(gmix(vec3(1.0f), Ctint, state.mat.specularTint) * (state.mat.specular * 0.08f), Cdlin, state.mat.metallic);
  vec3  Csheen = state.mat.sheenTint;

  if(rnd(seed) < transWeight)
  {
    vec3 H = ImportanceSampleGTR2(state.mat.roughness, r1, r2);
    H      = state.tangent * H.x + state.bitangent * H.y + N * H.z;

    vec3  R = reflect(-V, H);
    float F = DielectricFresnel(std::fabs(dot(R, H)), state.eta);

    if(state.mat.thinwalled)
    {
      if(dot(state.ffnormal, state.normal) < 0.0f)
        F = 0;
      state.eta = 1.001f;
    }

    if(rnd(seed) < F)
    {
      L = normalize(R);
      f = EvalDielectricReflection(state, V, N, L, H, pdf);
    }
    else
    {
      L = normalize(refract(-V, H, state.eta));
      f = EvalDielectricRefraction(state, V, N, L, H, pdf);
    }
    f *= transWeight;
    pdf *= transWeight;
  }
  else
  {
    if(rnd(seed) < diffuseRatio)
    {
      if(rnd(seed) < state.mat.subsurface)
      {
        L = UniformSampleHemisphere(r1, r2);
        L = state.tangent * L.x + state.bitangent * L.y - N * L.z;
        f = EvalSubsurface(state, V, N, L, pdf);
        pdf *= state.mat.subsurface * diffuseRatio;
        state.isSubsurface = true;
      }
      else
      {
        L      = CosineSampleHemisphere(r1, r2);
        L      = state.tangent * L.x + state.bitangent * L.y + N * L.z;
        vec3 H = normalize(L + V);
        f      = EvalDiffuse(state, Csheen, V, N, L, H, pdf);
        pdf *= (1.0f - state.mat.subsurface) * diffuseRatio;
      }
    }
    else
    {
      float primarySpecRatio = 1.0f / (1.0f + state.mat.clearcoat);
      if(rnd(seed) < primarySpecRatio)
      {
        vec3 H = ImportanceSampleGTR2_aniso(state.mat.ax, state.mat.ay, r1, r2);
        H      = state.tangent * H.x + state.bitangent * H.y + N * H.z;
        L      = normalize(reflect(-V, H));
        f      = EvalSpecular(state, Cspec0, V, N, L, H, pdf);
        pdf *= primarySpecRatio * (1.0f - diffuseRatio);
      }
      else
      {
        vec3 H = ImportanceSampleGTR1(state.mat.clearcoatRoughness, r1, r2);
        H      = state.tangent * H.x + state.bitangent * H.y + N * H.z;
        L      = normalize(reflect(-V, H));
        f      = EvalClearcoat(state, V, N, L, H, pdf);
        pdf *= (1.0f - primarySpecRatio) * (1.0f - diffuseRatio);
      }
    }
    f *= (1.0f - transWeight);
    pdf *= (1.0f - transWeight);
  }
  return f;
}

// :525-599
inline vec3 DisneyEval(const State& state, vec3 V, vec3 N, vec3 L, float& pdf)
{
  vec3 H;
  if(dot(N, L) < 0.0f)
    H = normalize(L * (1.0f / state.eta) + V);
  else
    H = normalize(L + V);
  if(dot(N, H) < 0.0f)
    H = -H;

  float diffuseRatio     = 0.5f * (1.0f - state.mat.metallic);
  float primarySpecRatio = 1.0f / (1.0f + state.mat.clearcoat);
  float transWeight      = (1.0f - state.mat.metallic) * state.mat.transmission;

  vec3  brdf(0.0f), bsdf(0.0f);
  float brdfPdf = 0.0f, bsdfPdf = 0.0f;

  if(transWeight > 0.0f)
  {
    if(dot(N, L) < 0.0f)
      bsdf = EvalDielectricRefraction(state, V, N, L, H, bsdfPdf);
    else
      bsdf = EvalDielectricReflection(state, V, N, L, H, bsdfPdf);
  }

  float m_pdf = 0.0f;  // uninitialised in the reference; only read after being written on the paths taken
  if(transWeight < 1.0f)
  {
    if(dot(N, L) < 0.0f)
    {
      if(state.mat.subsurface > 0.0f)
      {
        brdf    = EvalSubsurface(state, V, N, L, m_pdf);
        brdfPdf = m_pdf * state.mat.subsurface * diffuseRatio;
      }
    }
    else
    {
      vec3  Cdlin  = state.mat.albedo;
      float Cdlum  = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;
      vec3  Ctint  = Cdlum > 0.0f ? Cdlin / Cdlum : vec3(1.0f);
      vec3  Cspec0 = gmix(gmix(vec3(1.0f), Ctint, state.mat.specularTint) * (state.mat.specular * 0.08f), Cdlin, state.mat.metallic);
      vec3  Csheen = state.mat.sheenTint;

      brdf += EvalDiffuse(state, Csheen, V, N, L, H, m_pdf);
      brdfPdf += m_pdf * (1.0f - state.mat.subsurface) * diffuseRatio;

      brdf += EvalSpecular(state, Cspec0, V, N, L, H, m_pdf);
      brdfPdf += m_pdf * primarySpecRatio * (1.0f - diffuseRatio);

      brdf += EvalClearcoat(state, V, N, L, H, m_pdf);
      brdfPdf += m_pdf * (1.0f - primarySpecRatio) * (1.0f - diffuseRatio);
    }
  }
  pdf = gmix(brdfPdf, bsdfPdf, transWeight);
  return gmix(brdf, bsdf, transWeight);
}

// ================================= shaders/pbr_gltf.glsl ====================================
// :38-41
inline vec3 F_Schlick(vec3 f0, vec3 f90, float VdotH) { return f0 + (f90 - f0) * mpow(gclamp(1.0f - VdotH, 0.0f, 1.0f), 5.0f); }
// :43-46
inline float F_Schlick(float f0, float f90, float VdotH) { return f0 + (f90 - f0) * mpow(gclamp(1.0f - VdotH, 0.0f, 1.0f), 5.0f); }
// :54-67
inline float V_GGX(float NdotL, float NdotV, float alphaRoughness)
{
  float a2   = alphaRoughness * alphaRoughness;
  float GGXV = NdotL * std::sqrt(NdotV * NdotV * (1.0f - a2) + a2);
  float GGXL = NdotV * std::sqrt(NdotL * NdotL * (1.0f - a2) + a2);
  float GGX  = GGXV + GGXL;
  if(GGX > 0.0f)
    return 0.5f / GGX;
  return 0.0f;
}
// :71-77
inline float V_GGX_anisotropic(float NdotL, float NdotV, float BdotV, float TdotV, float TdotL, float BdotL, float /*anisotropy*/, float at, float ab)
{
  float GGXV = NdotL * length(vec3(at * TdotV, ab * BdotV, NdotV));
  float GGXL = NdotV * length(vec3(at * TdotL, ab * BdotL, NdotL));
  float v    = 0.5f / (GGXV + GGXL);
  return gclamp(v, 0.0f, 1.0f);
}
// :99-104
inline float D_GGX(float NdotH, float alphaRoughness)
{
  float a2 = alphaRoughness * alphaRoughness;
  float f  = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
  return a2 / (M_PI_F * f * f);
}
// :109-115
inline float D_GGX_anisotropic(float NdotH, float TdotH, float BdotH, float /*anisotropy*/, float at, float ab)
{
  float a2 = at * ab;
  vec3  f  = vec3(ab * TdotH, at * BdotH, a2 * NdotH);
  float w2 = a2 / dot(f, f);
  return a2 * w2 * w2 / M_PI_F;
}
// :134-140
inline vec3 BRDF_lambertian(vec3 /*f0*/, vec3 /*f90*/, vec3 diffuseColor, float /*VdotH*/, float metallic)
{
  return (diffuseColor / M_PI_F) * (1.0f - metallic);
}
// :143-150
inline vec3 BRDF_specularGGX(vec3 f0, vec3 f90, float alphaRoughness, float VdotH, float NdotL, float NdotV, float NdotH)
{
  vec3  F = F_Schlick(f0, f90, VdotH);
  float V = V_GGX(NdotL, NdotV, alphaRoughness);
  float D = D_GGX(NdotH, gmax(0.001f, alphaRoughness));
  return F * V * D;
}
// :153-178
inline vec3 BRDF_specularAnisotropicGGX(vec3 f0, vec3 f90, float alphaRoughness, float VdotH, float NdotL, float NdotV, float NdotH, float BdotV,
                                        float TdotV, float TdotL, float BdotL, float TdotH, float BdotH, float anisotropy)
{
  float at = gmax(alphaRoughness * (1.0f + anisotropy), 0.00001f);
  float ab = gmax(alphaRoughness * (1.0f - anisotropy), 0.00001f);
  vec3  F  = F_Schlick(f0, f90, VdotH);
  float V  = V_GGX_anisotropic(NdotL, NdotV, BdotV, TdotV, TdotL, BdotL, anisotropy, at, ab);
  float D  = D_GGX_anisotropic(NdotH, TdotH, BdotH, anisotropy, at, ab);
  return F * V * D;
}
// :191-202
inline vec3 GgxSampling(float specularAlpha, float r1, float r2)
{
  float phi      = r1 * 2.0f * M_PI_F;
  float cosTheta = std::sqrt((1.0f - r2) / (1.0f + (specularAlpha * specularAlpha - 1.0f) * r2));
  float sinTheta = gclamp(std::sqrt(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
  float sinPhi   = msin(phi);
  float cosPhi   = mcos(phi);
  return vec3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
// :207-224
inline vec3 EvalDiffuseGltf(const State& state, vec3 f0, vec3 f90, vec3 V, vec3 N, vec3 L, vec3 H, float& pdf)
{
  pdf         = 0;
  float NdotV = dot(N, V);
  float NdotL = dot(N, L);
  if(NdotL < 0.0f || NdotV < 0.0f)
    return vec3(0.0f);
  NdotL       = gclamp(NdotL, 0.001f, 1.0f);
  NdotV       = gclamp(std::fabs(NdotV), 0.001f, 1.0f);
  float VdotH = dot(V, H);
  pdf         = NdotL * M_1_OVER_PI;
  return BRDF_lambertian(f0, f90, state.mat.albedo, VdotH, state.mat.metallic);
}
// :229-263
inline vec3 EvalAnisotropicSpecularGltf(const State& state, vec3 f0, vec3 f90, vec3 V, vec3 N, vec3 L, vec3 H, float& pdf)
{
  pdf         = 0;
  float NdotL = dot(N, L);
  if(NdotL < 0.0f)
    return vec3(0.0f);
  vec3  T     = state.tangent;
  vec3  B     = state.bitangent;
  float TdotV = gclamp(dot(T, V), 0.0f, 1.0f);
  float BdotV = gclamp(dot(B, V), 0.0f, 1.0f);
  float TdotL = dot(T, L);
  float BdotL = dot(B, L);
  float TdotH = dot(T, H);
  float BdotH = dot(B, H);
  float NdotH = dot(N, H);
  float NdotV = dot(N, V);
  float VdotH = dot(V, H);
  float LdotH = dot(L, H);
  NdotL       = gclamp(NdotL, 0.001f, 1.0f);
  NdotV       = gclamp(std::fabs(NdotV), 0.001f, 1.0f);
  float at    = gmax(state.mat.roughness * (1.0f + state.mat.anisotropy), 0.001f);
  float ab    = gmax(state.mat.roughness * (1.0f - state.mat.anisotropy), 0.001f);
  pdf         = D_GGX_anisotropic(NdotH, TdotH, BdotH, state.mat.anisotropy, at, ab) / (4.0f * LdotH);
  return BRDF_specularAnisotropicGGX(f0, f90, state.mat.roughness, VdotH, NdotL, NdotV, NdotH, BdotV, TdotV, TdotL, BdotL, TdotH, BdotH,
                                     state.mat.anisotropy);
}
// :267-290
inline vec3 EvalSpecularGltf(const State& state, vec3 f0, vec3 f90, vec3 V, vec3 N, vec3 L, vec3 H, float& pdf)
{
  if(state.mat.anisotropy > 0)
    return EvalAnisotropicSpecularGltf(state, f0, f90, V, N, L, H, pdf);
  pdf         = 0;
  float NdotL = dot(N, L);
  if(NdotL < 0.0f)
    return vec3(0.0f);
  float NdotV = dot(N, V);
  float NdotH = gclamp(dot(N, H), 0.0f, 1.0f);
  float LdotH = gclamp(dot(L, H), 0.0f, 1.0f);
  float VdotH = gclamp(dot(V, H), 0.0f, 1.0f);
  NdotL       = gclamp(NdotL, 0.001f, 1.0f);
  NdotV       = gclamp(std::fabs(NdotV), 0.001f, 1.0f);
  pdf         = D_GGX(NdotH, state.mat.roughness) * NdotH / (4.0f * LdotH);
  return BRDF_specularGGX(f0, f90, state.mat.roughness, VdotH, NdotL, NdotV, NdotH);
}
// :295-321
inline vec3 EvalClearcoatGltf(const State& state, vec3 V, vec3 N, vec3 L, vec3 H, float& pdf)
{
  pdf         = 0;
  float NdotL = dot(N, L);
  if(NdotL < 0.0f)
    return vec3(0.0f);
  float NdotH = dot(N, H);
  float NdotV = dot(N, V);
  float VdotH = dot(V, H);
  float LdotH = dot(L, H);
  NdotL       = gclamp(NdotL, 0.001f, 1.0f);
  NdotV       = gclamp(std::fabs(NdotV), 0.001f, 1.0f);
  float clearcoat        = state.mat.clearcoat;
  float clearcoatFresnel = F_Schlick(0.04f, 1.0f, VdotH);
  float clearcoatAlpha   = state.mat.clearcoatRoughness * state.mat.clearcoatRoughness;
  float G                = V_GGX(NdotL, NdotV, clearcoatAlpha);
  float D                = D_GGX(NdotH, gmax(0.001f, clearcoatAlpha));
  pdf                    = D * NdotH / (4.0f * LdotH);
  return vec3(clearcoatFresnel * D * G * clearcoat);
}
// :346-349 (the stub: Appendix C-6)
inline vec3 EvalDielectricRefractionGltf(const State& state, vec3 /*V*/, vec3 N, vec3 L, vec3 /*H*/, float& pdf)
{
  pdf = std::fabs(dot(N, L));
  return state.mat.albedo;
}
// :371-439
inline vec3 PbrEval(const State& state, vec3 V, vec3 N, vec3 L, float& pdf_out)
{
  vec3 H;
  if(dot(N, L) < 0.0f)
    H = normalize(L * (1.0f / state.eta) + V);
  else
    H = normalize(L + V);
  if(dot(N, H) < 0.0f)
    H = -H;

  float transWeight = (1.0f - state.mat.metallic) * state.mat.transmission;
  vec3  brdf(0.0f), bsdf(0.0f);
  float brdfPdf = 0.0f, bsdfPdf = 0.0f;

  if(transWeight > 0.0f)
    bsdf = EvalDielectricRefractionGltf(state, V, N, L, H, bsdfPdf);

  if(transWeight < 1.0f && dot(N, L) > 0)
  {
    float pdf;
    float diffuseRatio     = 0.5f * (1.0f - state.mat.metallic);
    float specularRatio    = 1.0f - diffuseRatio;
    float primarySpecRatio = 1.0f / (1.0f + state.mat.clearcoat);

    vec3  specularCol = state.mat.f0;
    float reflectance = gmax(gmax(specularCol.x, specularCol.y), specularCol.z);
    vec3  f0          = specularCol;
    vec3  f90         = vec3(gclamp(reflectance * 50.0f, 0.0f, 1.0f));

    brdf += EvalDiffuseGltf(state, f0, f90, V, N, L, H, pdf);
    brdfPdf += pdf * diffuseRatio;

    brdf += EvalClearcoatGltf(state, V, N, L, H, pdf);
    brdfPdf += pdf * (1.0f - primarySpecRatio) * specularRatio;

    brdf += EvalSpecularGltf(state, f0, f90, V, N, L, H, pdf);
    brdfPdf += pdf * primarySpecRatio * specularRatio;
  }
  pdf_out = gmix(brdfPdf, bsdfPdf, transWeight);
  return gmix(brdf, bsdf, transWeight);
}
// :444-554
inline vec3 PbrSample(const State& state, vec3 V, vec3 N, vec3& L, float& pdf, uint32_t& seed)
{
  pdf = 0.0f;
  vec3 brdf(0.0f);

  float probability   = rnd(seed);
  float diffuseRatio  = 0.5f * (1.0f - state.mat.metallic);
  float specularRatio = 1.0f - diffuseRatio;
  float transWeight   = (1.0f - state.mat.metallic) * state.mat.transmission;

  float r1 = rnd(seed);
  float r2 = rnd(seed);

  if(rnd(seed) < transWeight)
  {
    float eta         = state.eta;
    float n1          = 1.0f;
    float n2          = state.mat.ior;
    float R0          = (n1 - n2) / (n1 + n2);
    vec3  H           = GgxSampling(state.mat.roughness, r1, r2);
    H                 = state.tangent * H.x + state.bitangent * H.y + N * H.z;
    float VdotH       = dot(V, H);
    float F           = F_Schlick(R0 * R0, 1.0f, VdotH);
    float discriminat = 1.0f - eta * eta * (1.0f - VdotH * VdotH);

    if(state.mat.thinwalled)
    {
      if(dot(state.ffnormal, state.normal) < 0.0f)
      {
        F           = 0;
        discriminat = 0;
      }
      eta = 1.00f;
    }

    if(discriminat < 0.0f || rnd(seed) < F)
    {
      L = normalize(reflect(-V, H));
    }
    else
    {
      L = normalize(refract(-V, H, eta));
      if(std::isnan(L.x) || std::isnan(L.y) || std::isnan(L.z))
        L = -V;
    }
    brdf = EvalDielectricRefractionGltf(state, V, N, L, H, pdf);
  }
  else
  {
    vec3  specularCol = state.mat.f0;
    float reflectance = gmax(gmax(specularCol.x, specularCol.y), specularCol.z);
    vec3  f0          = specularCol;
    vec3  f90         = vec3(gclamp(reflectance * 50.0f, 0.0f, 1.0f));
    vec3  T           = state.tangent;
    vec3  B           = state.bitangent;

    if(probability < diffuseRatio)
    {
      L      = CosineSampleHemisphere(r1, r2);
      L      = T * L.x + B * L.y + N * L.z;
      vec3 H = normalize(L + V);
      brdf   = EvalDiffuseGltf(state, f0, f90, V, N, L, H, pdf);
      pdf *= (1.0f - state.mat.subsurface) * diffuseRatio;
    }
    else
    {
      float primarySpecRatio = 1.0f / (1.0f + state.mat.clearcoat);
      float roughness;
      if(rnd(seed) < primarySpecRatio)
        roughness = state.mat.roughness;
      else
        roughness = state.mat.clearcoatRoughness;

      vec3 H = GgxSampling(roughness, r1, r2);
      H      = T * H.x + B * H.y + N * H.z;
      L      = reflect(-V, H);

      if(rnd(seed) < primarySpecRatio)
      {
        brdf = EvalSpecularGltf(state, f0, f90, V, N, L, H, pdf);
        pdf *= primarySpecRatio * specularRatio;
      }
      else
      {
        brdf = EvalClearcoatGltf(state, V, N, L, H, pdf);
        pdf *= (1.0f - primarySpecRatio) * specularRatio;
      }
    }
    brdf *= (1.0f - transWeight);
    pdf *= (1.0f - transWeight);
  }
  return brdf;
}

}  // namespace orc
