// BSDF evaluation and sampling on gfx950: the Disney principled model and the glTF PBR model, selected
// per frame by RtxState::pbrMode (reference: shaders/pathtrace.glsl:40-56).
//
// Behavioural contract (file:line = reference).  The quirks listed in SURVEY.md Appendix C are part of
// the specification and are kept: GTR1 sampling ignores its second random number, the clearcoat Smith
// term uses alpha = 0.25, the glTF refraction lobe evaluates to (albedo, |N.L|), the glTF specular
// branch draws two independent lobe selectors, thin-walled Disney transmission uses eta = 1.001.
//   disney_*   shaders/pbr_disney.glsl:68-229 (sampling + terms), :320-413 (lobes), :417-521 (sample), :525-599 (eval)
//   gltf_*     shaders/pbr_gltf.glsl:31-199 (terms), :204-361 (lobes), :365-439 (eval), :444-554 (sample)
// Random numbers are drawn in exactly the reference's order.
#pragma once
#include "pt_surface.h"

// ---- shared microfacet terms ------------------------------------------------------------------------
PT_DEV float schlick_weight(float u)
{
  float m  = clampf(1.0f - u, 0.0f, 1.0f);
  float m2 = m * m;
  return m2 * m2 * m;
}
PT_DEV float fresnel_dielectric(float cosI, float eta)
{
  float sinT2 = eta * eta * (1.0f - cosI * cosI);
  if(sinT2 > 1.0f)
    return 1.0f;
  float cosT = sqrtf(fmax2(1.0f - sinT2, 0.0f));
  float rs   = (eta * cosT - cosI) / (eta * cosT + cosI);
  float rp   = (eta * cosI - cosT) / (eta * cosI + cosT);
  return 0.5f * (rs * rs + rp * rp);
}
PT_DEV float gtr1(float NdotH, float a)
{
  if(a >= 1.0f)
    return PT_1_OVER_PI;
  float a2 = a * a;
  float t  = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
  return (a2 - 1.0f) / (PT_PI * pt_log(a2) * t);
}
PT_DEV float gtr2(float NdotH, float a)
{
  float a2 = a * a;
  float t  = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
  return a2 / (PT_PI * t * t);
}
PT_DEV float gtr2_aniso(float NdotH, float HdotX, float HdotY, float ax, float ay)
{
  float a = HdotX / ax;
  float b = HdotY / ay;
  float c = a * a + b * b + NdotH * NdotH;
  return 1.0f / (PT_PI * ax * ay * c * c);
}
PT_DEV float smith_ggx(float NdotV, float alphaG)
{
  float a = alphaG * alphaG;
  float b = NdotV * NdotV;
  return 1.0f / (NdotV + sqrtf(a + b - a * b));
}
PT_DEV float smith_ggx_aniso(float NdotV, float VdotX, float VdotY, float ax, float ay)
{
  float a = VdotX * ax;
  float b = VdotY * ay;
  float c = NdotV;
  return 1.0f / (NdotV + sqrtf(a * a + b * b + c * c));
}
PT_DEV f3 sample_cosine_hemisphere(float r1, float r2)
{
  float r   = sqrtf(r1);
  float phi = PT_TWO_PI * r2;
  f3    d;
  d.x = r * pt_cos(phi);
  d.y = r * pt_sin(phi);
  d.z = sqrtf(fmax2(0.0f, 1.0f - d.x * d.x - d.y * d.y));
  return d;
}
PT_DEV f3 sample_uniform_hemisphere(float r1, float r2)
{
  float r   = sqrtf(fmax2(0.0f, 1.0f - r1 * r1));
  float phi = PT_TWO_PI * r2;
  return f3{r * pt_cos(phi), r * pt_sin(phi), r1};
}
PT_DEV float power_heuristic(float a, float b)
{
  float t = a * a;
  return t / (b * b + t);
}
PT_DEV f3 sample_gtr1(float rgh, float r1)
{
  float a        = fmax2(0.001f, rgh);
  float a2       = a * a;
  float phi      = r1 * PT_TWO_PI;
  float cosTheta = sqrtf((1.0f - pt_pow(a2, 1.0f - r1)) / (1.0f - a2));
  float sinTheta = clampf(sqrtf(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
  return f3{sinTheta * pt_cos(phi), sinTheta * pt_sin(phi), cosTheta};
}
PT_DEV f3 sample_gtr2_aniso(float ax, float ay, float r1, float r2)
{
  float phi      = r1 * PT_TWO_PI;
  float sinPhi   = ay * pt_sin(phi);
  float cosPhi   = ax * pt_cos(phi);
  float tanTheta = sqrtf(r2 / (1 - r2));
  return f3{tanTheta * cosPhi, tanTheta * sinPhi, 1.0f};
}
PT_DEV f3 sample_gtr2(float rgh, float r1, float r2)
{
  float a        = fmax2(0.001f, rgh);
  float phi      = r1 * PT_TWO_PI;
  float cosTheta = sqrtf((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
  float sinTheta = clampf(sqrtf(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
  return f3{sinTheta * pt_cos(phi), sinTheta * pt_sin(phi), cosTheta};
}

// ======================================= Disney ========================================================
PT_DEV f3 disney_reflection(const Surface& s, float eta, f3 V, f3 N, f3 L, f3 H, float& pdf)
{
  if(dot3(N, L) < 0.0f)
    return splat3(0.0f);
  float F = fresnel_dielectric(dot3(V, H), eta);
  float D = gtr2(dot3(N, H), s.roughness);
  pdf     = D * dot3(N, H) * F / (4.0f * dot3(V, H));
  float G = smith_ggx(fabsf(dot3(N, L)), s.roughness) * smith_ggx(dot3(N, V), s.roughness);
  return s.albedo * F * D * G;
}
PT_DEV f3 disney_refraction(const Surface& s, float eta, f3 V, f3 N, f3 L, f3 H, float& pdf)
{
  float F     = fresnel_dielectric(fabsf(dot3(V, H)), eta);
  float D     = gtr2(dot3(N, H), s.roughness);
  float denom = dot3(L, H) * eta + dot3(V, H);
  pdf         = D * dot3(N, H) * (1.0f - F) * fabsf(dot3(L, H)) / (denom * denom);
  float G     = smith_ggx(fabsf(dot3(N, L)), s.roughness) * smith_ggx(dot3(N, V), s.roughness);
  return s.albedo * (1.0f - F) * D * G * fabsf(dot3(V, H)) * fabsf(dot3(L, H)) * 4.0f * eta * eta / (denom * denom);
}
PT_DEV f3 disney_specular(const Surface& s, f3 Cspec0, f3 V, f3 N, f3 L, f3 H, float& pdf)
{
  if(dot3(N, L) < 0.0f)
    return splat3(0.0f);
  float D  = gtr2_aniso(dot3(N, H), dot3(H, s.tangent), dot3(H, s.bitangent), s.ax, s.ay);
  pdf      = D * dot3(N, H) / (4.0f * dot3(V, H));
  float FH = schlick_weight(dot3(L, H));
  f3    F  = lerp(Cspec0, splat3(1.0f), FH);
  float G  = smith_ggx_aniso(dot3(N, L), dot3(L, s.tangent), dot3(L, s.bitangent), s.ax, s.ay);
  G *= smith_ggx_aniso(dot3(N, V), dot3(V, s.tangent), dot3(V, s.bitangent), s.ax, s.ay);
  return F * D * G;
}
PT_DEV f3 disney_clearcoat(const Surface& s, f3 V, f3 N, f3 L, f3 H, float& pdf)
{
  if(dot3(N, L) < 0.0f)
    return splat3(0.0f);
  float D  = gtr1(dot3(N, H), s.clearcoatRoughness);
  pdf      = D * dot3(N, H) / (4.0f * dot3(V, H));
  float FH = schlick_weight(dot3(L, H));
  float F  = lerp(0.04f, 1.0f, FH);
  float G  = smith_ggx(dot3(N, L), 0.25f) * smith_ggx(dot3(N, V), 0.25f);
  return splat3(0.25f * s.clearcoat * F * D * G);
}
PT_DEV f3 disney_diffuse(const Surface& s, f3 Csheen, f3 V, f3 N, f3 L, f3 H, float& pdf)
{
  if(dot3(N, L) < 0.0f)
    return splat3(0.0f);
  pdf          = dot3(N, L) * (1.0f / PT_PI);
  float FL     = schlick_weight(dot3(N, L));
  float FV     = schlick_weight(dot3(N, V));
  float FH     = schlick_weight(dot3(L, H));
  float Fd90   = 0.5f + 2.0f * dot3(L, H) * dot3(L, H) * s.roughness;
  float Fd     = lerp(1.0f, Fd90, FL) * lerp(1.0f, Fd90, FV);
  f3    Fsheen = Csheen * (FH * s.sheen);
  return ((1.0f / PT_PI) * Fd * (1.0f - s.subsurface) * s.albedo + Fsheen) * (1.0f - s.metallic);
}
PT_DEV f3 disney_subsurface(const Surface& s, f3 V, f3 N, f3 L, float& pdf)
{
  pdf      = (1.0f / PT_TWO_PI);
  float FL = schlick_weight(fabsf(dot3(N, L)));
  float FV = schlick_weight(dot3(N, V));
  float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
  return sqrt3(s.albedo) * s.subsurface * (1.0f / PT_PI) * Fd * (1.0f - s.metallic) * (1.0f - s.transmission);
}
PT_DEV f3 disney_spec_tint(const Surface& s)
{
  f3    Cdlin = s.albedo;
  float Cdlum = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;
  f3    Ctint = Cdlum > 0.0f ? Cdlin / Cdlum : splat3(1.0f);
  return lerp(lerp(splat3(1.0f), Ctint, s.specularTint) * (s.specular * 0.08f), Cdlin, s.metallic);
}

PT_DEV f3 disney_sample(const Surface& s, f3 V, f3 N, f3& L, float& pdf, uint32_t& seed)
{
  pdf  = 0.0f;
  f3 f = splat3(0.0f);

  float r1 = rng_next(seed);
  float r2 = rng_next(seed);

  float diffuseRatio = 0.5f * (1.0f - s.metallic);
  float transWeight  = (1.0f - s.metallic) * s.transmission;
  f3    Cspec0       = disney_spec_tint(s);
  f3    Csheen       = s.sheenTint;

  if(rng_next(seed) < transWeight)
  {
    f3 H = sample_gtr2(s.roughness, r1, r2);
    H    = s.tangent * H.x + s.bitangent * H.y + N * H.z;

    f3    R   = mirror(-V, H);
    float eta = s.eta;
    float F   = fresnel_dielectric(fabsf(dot3(R, H)), eta);
    if(s.thinwalled)
    {
      if(dot3(s.ffnormal, s.normal) < 0.0f)
        F = 0;
      eta = 1.001f;  // local to this sample: the integrator keeps the material's eta
    }
    if(rng_next(seed) < F)
    {
      L = unit(R);
      f = disney_reflection(s, eta, V, N, L, H, pdf);
    }
    else
    {
      L = unit(bend(-V, H, eta));
      f = disney_refraction(s, eta, V, N, L, H, pdf);
    }
    f *= transWeight;
    pdf *= transWeight;
  }
  else
  {
    if(rng_next(seed) < diffuseRatio)
    {
      if(rng_next(seed) < s.subsurface)
      {
        L = sample_uniform_hemisphere(r1, r2);
        L = s.tangent * L.x + s.bitangent * L.y - N * L.z;
        f = disney_subsurface(s, V, N, L, pdf);
        pdf *= s.subsurface * diffuseRatio;
      }
      else
      {
        L    = sample_cosine_hemisphere(r1, r2);
        L    = s.tangent * L.x + s.bitangent * L.y + N * L.z;
        f3 H = unit(L + V);
        f    = disney_diffuse(s, Csheen, V, N, L, H, pdf);
        pdf *= (1.0f - s.subsurface) * diffuseRatio;
      }
    }
    else
    {
      float primarySpecRatio = 1.0f / (1.0f + s.clearcoat);
      if(rng_next(seed) < primarySpecRatio)
      {
        f3 H = sample_gtr2_aniso(s.ax, s.ay, r1, r2);
        H    = s.tangent * H.x + s.bitangent * H.y + N * H.z;
        L    = unit(mirror(-V, H));
        f    = disney_specular(s, Cspec0, V, N, L, H, pdf);
        pdf *= primarySpecRatio * (1.0f - diffuseRatio);
      }
      else
      {
        f3 H = sample_gtr1(s.clearcoatRoughness, r1);
        H    = s.tangent * H.x + s.bitangent * H.y + N * H.z;
        L    = unit(mirror(-V, H));
        f    = disney_clearcoat(s, V, N, L, H, pdf);
        pdf *= (1.0f - primarySpecRatio) * (1.0f - diffuseRatio);
      }
    }
    f *= (1.0f - transWeight);
    pdf *= (1.0f - transWeight);
  }
  return f;
}

PT_DEV f3 disney_eval(const Surface& s, f3 V, f3 N, f3 L, float& pdf)
{
  f3 H;
  if(dot3(N, L) < 0.0f)
    H = unit(L * (1.0f / s.eta) + V);
  else
    H = unit(L + V);
  if(dot3(N, H) < 0.0f)
    H = -H;

  float diffuseRatio     = 0.5f * (1.0f - s.metallic);
  float primarySpecRatio = 1.0f / (1.0f + s.clearcoat);
  float transWeight      = (1.0f - s.metallic) * s.transmission;

  f3    brdf = splat3(0.0f), bsdf = splat3(0.0f);
  float brdfPdf = 0.0f, bsdfPdf = 0.0f;

  if(transWeight > 0.0f)
  {
    if(dot3(N, L) < 0.0f)
      bsdf = disney_refraction(s, s.eta, V, N, L, H, bsdfPdf);
    else
      bsdf = disney_reflection(s, s.eta, V, N, L, H, bsdfPdf);
  }

  float lobePdf = 0.0f;
  if(transWeight < 1.0f)
  {
    if(dot3(N, L) < 0.0f)
    {
      if(s.subsurface > 0.0f)
      {
        brdf    = disney_subsurface(s, V, N, L, lobePdf);
        brdfPdf = lobePdf * s.subsurface * diffuseRatio;
      }
    }
    else
    {
      f3 Cspec0 = disney_spec_tint(s);
      brdf += disney_diffuse(s, s.sheenTint, V, N, L, H, lobePdf);
      brdfPdf += lobePdf * (1.0f - s.subsurface) * diffuseRatio;
      brdf += disney_specular(s, Cspec0, V, N, L, H, lobePdf);
      brdfPdf += lobePdf * primarySpecRatio * (1.0f - diffuseRatio);
      brdf += disney_clearcoat(s, V, N, L, H, lobePdf);
      brdfPdf += lobePdf * (1.0f - primarySpecRatio) * (1.0f - diffuseRatio);
    }
  }
  pdf = lerp(brdfPdf, bsdfPdf, transWeight);
  return lerp(brdf, bsdf, transWeight);
}

// ======================================== glTF ==========================================================
PT_DEV float schlick_pow5(float VdotH) { return pt_pow(clampf(1.0f - VdotH, 0.0f, 1.0f), 5.0f); }
PT_DEV f3 gltf_fresnel(f3 f0, f3 f90, float VdotH) { return f0 + (f90 - f0) * schlick_pow5(VdotH); }
PT_DEV float gltf_fresnel(float f0, float f90, float VdotH) { return f0 + (f90 - f0) * schlick_pow5(VdotH); }
PT_DEV float gltf_vis_ggx(float NdotL, float NdotV, float alphaRoughness)
{
  float a2   = alphaRoughness * alphaRoughness;
  float GGXV = NdotL * sqrtf(NdotV * NdotV * (1.0f - a2) + a2);
  float GGXL = NdotV * sqrtf(NdotL * NdotL * (1.0f - a2) + a2);
  float GGX  = GGXV + GGXL;
  return GGX > 0.0f ? 0.5f / GGX : 0.0f;
}
PT_DEV float gltf_vis_ggx_aniso(float NdotL, float NdotV, float BdotV, float TdotV, float TdotL, float BdotL, float at, float ab)
{
  float GGXV = NdotL * len3(f3{at * TdotV, ab * BdotV, NdotV});
  float GGXL = NdotV * len3(f3{at * TdotL, ab * BdotL, NdotL});
  float v    = 0.5f / (GGXV + GGXL);
  return clampf(v, 0.0f, 1.0f);
}
PT_DEV float gltf_d_ggx(float NdotH, float alphaRoughness)
{
  float a2 = alphaRoughness * alphaRoughness;
  float f  = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
  return a2 / (PT_PI * f * f);
}
PT_DEV float gltf_d_ggx_aniso(float NdotH, float TdotH, float BdotH, float at, float ab)
{
  float a2 = at * ab;
  f3    f  = f3{ab * TdotH, at * BdotH, a2 * NdotH};
  float w2 = a2 / dot3(f, f);
  return a2 * w2 * w2 / PT_PI;
}
PT_DEV f3 gltf_ggx_halfvector(float alpha, float r1, float r2)
{
  float phi      = r1 * 2.0f * PT_PI;
  float cosTheta = sqrtf((1.0f - r2) / (1.0f + (alpha * alpha - 1.0f) * r2));
  float sinTheta = clampf(sqrtf(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
  return f3{sinTheta * pt_cos(phi), sinTheta * pt_sin(phi), cosTheta};
}
PT_DEV f3 gltf_diffuse(const Surface& s, f3 V, f3 N, f3 L, float& pdf)
{
  pdf         = 0;
  float NdotV = dot3(N, V);
  float NdotL = dot3(N, L);
  if(NdotL < 0.0f || NdotV < 0.0f)
    return splat3(0.0f);
  NdotL = clampf(NdotL, 0.001f, 1.0f);
  pdf   = NdotL * PT_1_OVER_PI;
  return (s.albedo / PT_PI) * (1.0f - s.metallic);
}
PT_DEV f3 gltf_specular(const Surface& s, f3 f0, f3 f90, f3 V, f3 N, f3 L, f3 H, float& pdf)
{
  pdf         = 0;
  float NdotL = dot3(N, L);
  if(NdotL < 0.0f)
    return splat3(0.0f);
  if(s.anisotropy > 0)
  {
    f3    T = s.tangent, B = s.bitangent;
    float TdotV = clampf(dot3(T, V), 0.0f, 1.0f);
    float BdotV = clampf(dot3(B, V), 0.0f, 1.0f);
    float TdotL = dot3(T, L), BdotL = dot3(B, L), TdotH = dot3(T, H), BdotH = dot3(B, H);
    float NdotH = dot3(N, H), NdotV = dot3(N, V), VdotH = dot3(V, H), LdotH = dot3(L, H);
    NdotL       = clampf(NdotL, 0.001f, 1.0f);
    NdotV       = clampf(fabsf(NdotV), 0.001f, 1.0f);
    float at    = fmax2(s.roughness * (1.0f + s.anisotropy), 0.001f);
    float ab    = fmax2(s.roughness * (1.0f - s.anisotropy), 0.001f);
    pdf         = gltf_d_ggx_aniso(NdotH, TdotH, BdotH, at, ab) / (4.0f * LdotH);
    // BRDF_specularAnisotropicGGX re-derives its own (at, ab) with a 1e-5 floor (pbr_gltf.glsl:170-171)
    float at2 = fmax2(s.roughness * (1.0f + s.anisotropy), 0.00001f);
    float ab2 = fmax2(s.roughness * (1.0f - s.anisotropy), 0.00001f);
    f3    F   = gltf_fresnel(f0, f90, VdotH);
    float Vis = gltf_vis_ggx_aniso(NdotL, NdotV, BdotV, TdotV, TdotL, BdotL, at2, ab2);
    float D   = gltf_d_ggx_aniso(NdotH, TdotH, BdotH, at2, ab2);
    return F * Vis * D;
  }
  float NdotV = dot3(N, V);
  float NdotH = clampf(dot3(N, H), 0.0f, 1.0f);
  float LdotH = clampf(dot3(L, H), 0.0f, 1.0f);
  float VdotH = clampf(dot3(V, H), 0.0f, 1.0f);
  NdotL       = clampf(NdotL, 0.001f, 1.0f);
  NdotV       = clampf(fabsf(NdotV), 0.001f, 1.0f);
  pdf         = gltf_d_ggx(NdotH, s.roughness) * NdotH / (4.0f * LdotH);
  f3    F     = gltf_fresnel(f0, f90, VdotH);
  float Vis   = gltf_vis_ggx(NdotL, NdotV, s.roughness);
  float D     = gltf_d_ggx(NdotH, fmax2(0.001f, s.roughness));
  return F * Vis * D;
}
PT_DEV f3 gltf_clearcoat(const Surface& s, f3 V, f3 N, f3 L, f3 H, float& pdf)
{
  pdf         = 0;
  float NdotL = dot3(N, L);
  if(NdotL < 0.0f)
    return splat3(0.0f);
  float NdotH = dot3(N, H), NdotV = dot3(N, V), VdotH = dot3(V, H), LdotH = dot3(L, H);
  NdotL       = clampf(NdotL, 0.001f, 1.0f);
  NdotV       = clampf(fabsf(NdotV), 0.001f, 1.0f);
  float Fc    = gltf_fresnel(0.04f, 1.0f, VdotH);
  float alpha = s.clearcoatRoughness * s.clearcoatRoughness;
  float G     = gltf_vis_ggx(NdotL, NdotV, alpha);
  float D     = gltf_d_ggx(NdotH, fmax2(0.001f, alpha));
  pdf         = D * NdotH / (4.0f * LdotH);
  return splat3(Fc * D * G * s.clearcoat);
}
PT_DEV void gltf_f0_f90(const Surface& s, f3& f0, f3& f90)
{
  float reflectance = fmax2(fmax2(s.f0.x, s.f0.y), s.f0.z);
  f0                = s.f0;
  f90               = splat3(clampf(reflectance * 50.0f, 0.0f, 1.0f));
}

PT_DEV f3 gltf_eval(const Surface& s, f3 V, f3 N, f3 L, float& pdfOut)
{
  f3 H;
  if(dot3(N, L) < 0.0f)
    H = unit(L * (1.0f / s.eta) + V);
  else
    H = unit(L + V);
  if(dot3(N, H) < 0.0f)
    H = -H;

  float transWeight = (1.0f - s.metallic) * s.transmission;
  f3    brdf = splat3(0.0f), bsdf = splat3(0.0f);
  float brdfPdf = 0.0f, bsdfPdf = 0.0f;
  if(transWeight > 0.0f)
  {
    bsdfPdf = fabsf(dot3(N, L));
    bsdf    = s.albedo;
  }
  if(transWeight < 1.0f && dot3(N, L) > 0)
  {
    float pdf;
    float diffuseRatio     = 0.5f * (1.0f - s.metallic);
    float specularRatio    = 1.0f - diffuseRatio;
    float primarySpecRatio = 1.0f / (1.0f + s.clearcoat);
    f3    f0, f90;
    gltf_f0_f90(s, f0, f90);
    brdf += gltf_diffuse(s, V, N, L, pdf);
    brdfPdf += pdf * diffuseRatio;
    brdf += gltf_clearcoat(s, V, N, L, H, pdf);
    brdfPdf += pdf * (1.0f - primarySpecRatio) * specularRatio;
    brdf += gltf_specular(s, f0, f90, V, N, L, H, pdf);
    brdfPdf += pdf * primarySpecRatio * specularRatio;
  }
  pdfOut = lerp(brdfPdf, bsdfPdf, transWeight);
  return lerp(brdf, bsdf, transWeight);
}

PT_DEV f3 gltf_sample(const Surface& s, f3 V, f3 N, f3& L, float& pdf, uint32_t& seed)
{
  pdf     = 0.0f;
  f3 brdf = splat3(0.0f);

  float probability   = rng_next(seed);
  float diffuseRatio  = 0.5f * (1.0f - s.metallic);
  float specularRatio = 1.0f - diffuseRatio;
  float transWeight   = (1.0f - s.metallic) * s.transmission;
  float r1            = rng_next(seed);
  float r2            = rng_next(seed);

  if(rng_next(seed) < transWeight)
  {
    float eta   = s.eta;
    float n2    = s.ior;
    float R0    = (1.0f - n2) / (1.0f + n2);
    f3    H     = gltf_ggx_halfvector(s.roughness, r1, r2);
    H           = s.tangent * H.x + s.bitangent * H.y + N * H.z;
    float VdotH = dot3(V, H);
    float F     = gltf_fresnel(R0 * R0, 1.0f, VdotH);
    float disc  = 1.0f - eta * eta * (1.0f - VdotH * VdotH);
    if(s.thinwalled)
    {
      if(dot3(s.ffnormal, s.normal) < 0.0f)
      {
        F    = 0;
        disc = 0;
      }
      eta = 1.00f;
    }
    if(disc < 0.0f || rng_next(seed) < F)
    {
      L = unit(mirror(-V, H));
    }
    else
    {
      L = unit(bend(-V, H, eta));
      if(isnan(L.x) || isnan(L.y) || isnan(L.z))
        L = -V;
    }
    pdf  = fabsf(dot3(N, L));
    brdf = s.albedo;
  }
  else
  {
    f3 f0, f90;
    gltf_f0_f90(s, f0, f90);
    f3 T = s.tangent, B = s.bitangent;
    if(probability < diffuseRatio)
    {
      L    = sample_cosine_hemisphere(r1, r2);
      L    = T * L.x + B * L.y + N * L.z;
      brdf = gltf_diffuse(s, V, N, L, pdf);
      pdf *= (1.0f - s.subsurface) * diffuseRatio;
    }
    else
    {
      float primarySpecRatio = 1.0f / (1.0f + s.clearcoat);
      float roughness        = (rng_next(seed) < primarySpecRatio) ? s.roughness : s.clearcoatRoughness;
      f3    H                = gltf_ggx_halfvector(roughness, r1, r2);
      H                      = T * H.x + B * H.y + N * H.z;
      L                      = mirror(-V, H);
      if(rng_next(seed) < primarySpecRatio)
      {
        brdf = gltf_specular(s, f0, f90, V, N, L, H, pdf);
        pdf *= primarySpecRatio * specularRatio;
      }
      else
      {
        brdf = gltf_clearcoat(s, V, N, L, H, pdf);
        pdf *= (1.0f - primarySpecRatio) * specularRatio;
      }
    }
    brdf *= (1.0f - transWeight);
    pdf *= (1.0f - transWeight);
  }
  return brdf;
}
