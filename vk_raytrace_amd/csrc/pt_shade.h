// Per-path shading steps of the wavefront pipeline as plain inline functions of (scene, path state, slot): camera ray (pathtrace.comp:87-105 +
// pathtrace.glsl:348-387 samplePixel), one bounce of PathTrace (pathtrace.glsl:193-343: miss -> environment, shading state, material, DirectLight,
// BSDF sample, Russian roulette), the NEE add + roulette after the shadow ray, and the accumulation of a pixel (pathtrace.comp:122-133).
// pt_render.hip wraps them in kernels (k_generate, k_shade, k_shadow_*, k_tail, k_accumulate); tests/cpp/trace_host.cpp compiles the same
// functions for the host and renders whole frames with them, held bit for bit to the oracle (tests/test_trace_host.py).
#pragma once
#include "pt_bsdf.h"
#include "pt_internal.h"
#include "pt_sky.h"
#include "pt_settle.h"

enum { SHADE_DONE = 0, SHADE_TO_SHADOW = 1, SHADE_TO_NEXT = 2 };
enum { EV_MISS = 1u, EV_HIT = 2u, EV_NEE = 4u };

// Path slot -> pixel.  A local tile is 32x32 pixels = 16 waves of 8x8 pixels, so that the 64 lanes of a
// wavefront start out as a compact 8x8 pixel block (coherent primary rays, coalesced state accesses).
PT_DEV bool slot_pixel(const FrameParams& fp, const uint32_t* slotTile, uint32_t slot, int& px, int& py)
{
  uint32_t gt = slotTile[slot >> 10];
  uint32_t in = slot & 1023u, blk = in >> 6, lane = in & 63u;
  px          = int(gt % uint32_t(fp.tilesX)) * PT_TILE + int(blk & 3u) * 8 + int(lane & 7u);
  py          = int(gt / uint32_t(fp.tilesX)) * PT_TILE + int(blk >> 2) * 8 + int(lane >> 3);
  return px < fp.width && py < fp.height;
}

// column-major mat4 * vec4 with the reference's association order
PT_DEV f4 mat4_mul(const float* m, f4 v)
{
  f4 c0 = f4{m[0], m[1], m[2], m[3]}, c1 = f4{m[4], m[5], m[6], m[7]}, c2 = f4{m[8], m[9], m[10], m[11]}, c3 = f4{m[12], m[13], m[14], m[15]};
  return ((c0 * v.x + c1 * v.y) + c2 * v.z) + c3 * v.w;
}

// shaders/common.glsl:67-74
PT_DEV f2 spherical_uv(f3 v)
{
  float gamma = pt_asin(-v.y);
  float theta = pt_atan2(v.z, v.x);
  return f2{theta * PT_1_OVER_PI * 0.5f + 0.5f, gamma * PT_1_OVER_PI + 0.5f};
}
// shaders/common.glsl:98-113 (Ray Tracing Gems ch. 6)
PT_DEV f3 offset_ray(f3 p, f3 n)
{
  const float intScale = 256.0f, floatScale = 1.0f / 65536.0f, origin = 1.0f / 32.0f;
  int         ox = int(intScale * n.x), oy = int(intScale * n.y), oz = int(intScale * n.z);
  f3          pi = f3{__int_as_float(__float_as_int(p.x) + ((p.x < 0) ? -ox : ox)), __int_as_float(__float_as_int(p.y) + ((p.y < 0) ? -oy : oy)),
                      __int_as_float(__float_as_int(p.z) + ((p.z < 0) ? -oz : oz))};
  return f3{fabsf(p.x) < origin ? p.x + floatScale * n.x : pi.x, fabsf(p.y) < origin ? p.y + floatScale * n.y : pi.y,
            fabsf(p.z) < origin ? p.z + floatScale * n.z : pi.z};
}

// shaders/common.glsl:39-62
PT_DEV float heat_fade(float low, float high, float value)
{
  float mid = (low + high) * 0.5f, range = (high - low) * 0.5f;
  float x   = 1.0f - clampf(fabsf(mid - value) / range, 0.0f, 1.0f);
  return smooth(0.0f, 1.0f, x);
}
PT_DEV f3 heat_temperature(float intensity)
{
  const f3 blue = f3{0.f, 0.f, 1.f}, cyan = f3{0.f, 1.f, 1.f}, green = f3{0.f, 1.f, 0.f}, yellow = f3{1.f, 1.f, 0.f}, red = f3{1.f, 0.f, 0.f};
  return (((blue * heat_fade(-0.25f, 0.25f, intensity) + cyan * heat_fade(0.0f, 0.5f, intensity)) + green * heat_fade(0.25f, 0.75f, intensity)) + yellow * heat_fade(0.5f, 1.0f, intensity))
         + red * smooth(0.75f, 1.0f, intensity);
}

// the camera ray of a path (pixel px, py; frame of the batch fb): seed, sub-pixel jitter, depth of field.  `seed`: in / out -- the caller passes the state
// the stream continues from when fp.sample > 0 (pathtrace.comp:97-105), it is (re)seeded here for the first sample of a frame.
PT_DEV void camera_ray(const DeviceScene& S, const FrameParams& fp, uint32_t fb, int px, int py, uint32_t& seed, f3& org, f3& dir)
{
  pt_RtxState st = fp.st;
  st.frame += int(fb);
  if(fp.sample == 0)
    seed = rng_tea(uint32_t(st.size[0]) * uint32_t(py) + uint32_t(px), uint32_t(fp.variant == PT_VARIANT_RTX ? st.frame : st.frame * st.maxSamples));

  f2 jitter = f2{0.5f, 0.5f};
  if(st.frame != 0)
  {
    jitter.x = rng_next(seed);
    jitter.y = rng_next(seed);
  }
  f2 center = f2{float(px), float(py)} + jitter;
  f2 inUV   = f2{center.x / float(st.size[0]), center.y / float(st.size[1])};
  f2 d      = inUV * 2.0f - f2{1.0f, 1.0f};

  const pt_SceneCamera& cam = S.camera;
  f4 origin    = mat4_mul(cam.viewInverse, f4{0, 0, 0, 1});
  f4 target    = mat4_mul(cam.projInverse, f4{d.x, d.y, 1, 1});
  f3 tn        = unit(xyz(target));
  f4 direction = mat4_mul(cam.viewInverse, f4{tn.x, tn.y, tn.z, 0});

  f3    focalPoint = xyz(direction) * cam.focalDist;
  float cam_r1     = rng_next(seed) * PT_TWO_PI;
  float cam_r2     = rng_next(seed) * cam.aperture;
  f4    cam_right  = mat4_mul(cam.viewInverse, f4{1, 0, 0, 0});
  f4    cam_up     = mat4_mul(cam.viewInverse, f4{0, 1, 0, 0});
  f3    lens       = (xyz(cam_right) * pt_cos(cam_r1) + xyz(cam_up) * pt_sin(cam_r1)) * sqrtf(cam_r2);
  dir              = unit(focalPoint - lens);
  org              = xyz(origin) + lens;
}
// ... and into the path state (k_generate; the packet kernel computes it itself when FrameParams::regen is set)
PT_DEV void generate_ray(const DeviceScene& S, const RenderBuffers& rb, const FrameParams& fp, uint32_t slot, uint32_t fb, int px, int py)
{
  uint32_t seed = fp.sample == 0 ? 0u : __float_as_uint(rb.ps.rayD[slot].w);  // the stream continues across the samples of a frame (pathtrace.comp:97-105)
  f3       org, dir;
  camera_ray(S, fp, fb, px, py, seed, org, dir);
  rb.ps.rayO[slot]   = make_float4(org.x, org.y, org.z, 0.f);
  rb.ps.rayD[slot]   = make_float4(dir.x, dir.y, dir.z, __uint_as_float(seed));
  // throughput = 1, radiance = 0, absorption = 0 (pathtrace.glsl:201-203) are not written: shade_path knows them at depth 0 (48 B per sample
  // that would be written here and read back there).  Without a single bounce nobody else writes the radiance the accumulate step reads.
  if(fp.st.maxDepth == 0)
    rb.ps.rad[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- environment (shaders/env_sampling.glsl:38-135) ---------------------------------------------------------
PT_DEV f3 env_importance_sample(const DeviceScene& S, f3 xi, f3& toLight, float& pdf)
{
  const uint32_t    width = uint32_t(S.envW), height = uint32_t(S.envH);
  const uint32_t    size = width * height;
  uint32_t          idx  = uint32_t(xi.x * float(size));
  idx                    = idx < size - 1 ? idx : size - 1;
  const pt_EnvAccel e    = S.envAccel[idx];
  uint32_t          envIdx;
  if(xi.y < e.q)
  {
    envIdx = idx;
    xi.y /= e.q;
    pdf = e.pdf;
  }
  else
  {
    envIdx = e.alias;
    xi.y   = (xi.y - e.q) / (1.0f - e.q);
    pdf    = e.aliasPdf;
  }
  const uint32_t px = envIdx % width, py = envIdx / width;
  const float    u        = (float(px) + xi.y) / float(width);
  const float    phi      = u * (2.0f * PT_PI) - PT_PI;
  const float    sin_phi  = pt_sin(phi), cos_phi = pt_cos(phi);
  const float    step     = PT_PI / float(height);
  const float    theta0   = float(py) * step;
  const float    cosTheta = pt_cos(theta0) * (1.0f - xi.z) + pt_cos(theta0 + step) * xi.z;
  const float    theta    = pt_acos(cosTheta);
  const float    sinTheta = pt_sin(theta);
  const float    v        = theta * PT_1_OVER_PI;
  toLight                 = f3{cos_phi * sinTheta, cosTheta, sin_phi * sinTheta};
  return sample_env(S, f2{u, v});
}

PT_DEV float range_attenuation(float range, float distance)  // shaders/punctual.glsl:28-36
{
  if(range <= 0.0f)
    return 1.0f;
  return fmax2(fmin2(1.0f - pt_pow(distance / range, 4.0f), 1.0f), 0.0f) / pt_pow(distance, 2.0f);
}
PT_DEV float spot_attenuation(f3 pointToLight, f3 spotDir, float outerCos, float innerCos)  // shaders/punctual.glsl:39-51
{
  float c = dot3(unit(spotDir), unit(-pointToLight));
  if(c > outerCos)
  {
    if(c < innerCos)
      return smooth(outerCos, innerCos, c);
    return 1.0f;
  }
  return 0.0f;
}

PT_DEV f3 bsdf_eval(int pbrMode, const Surface& s, f3 V, f3 N, f3 L, float& pdf) { return pbrMode == 0 ? disney_eval(s, V, N, L, pdf) : gltf_eval(s, V, N, L, pdf); }
PT_DEV f3 bsdf_sample(int pbrMode, const Surface& s, f3 V, f3 N, f3& L, float& pdf, uint32_t& seed)
{
  return pbrMode == 0 ? disney_sample(s, V, N, L, pdf, seed) : gltf_sample(s, V, N, L, pdf, seed);
}

// ---- k_shade ----------------------------------------------------------------------------------------------
// One path: everything between the closest-hit trace and the shadow trace of a bounce.
// Returns where the path goes next (SHADE_*) and ORs the events it counted into `events` (EV_*).
// MODE: 0 / 1 = the common case compiled on its own -- Disney / glTF BSDF, no debug output, no sun & sky, no punctual lights (the host picks the
// kernel from the frame's uniform state), so that neither the other BSDF nor sun_and_sky() nor the debug / light branches cost registers or
// instruction-cache space; -1 = everything decided at run time.  Same arithmetic in every instantiation.
template <int MODE, class RB>
PT_DEV int shade_path(const DeviceScene& S, const RB& rb, const FrameParams& fp, uint32_t slot, int depth, uint32_t& events)
{
  const pt_RtxState& st   = fp.st;
  const int          pbrMode  = MODE >= 0 ? MODE : st.pbrMode;
  const bool         useSky   = MODE >= 0 ? false : (S.sunsky.in_use == 1);
  const int          nbLights = MODE >= 0 ? 0 : S.camera.nbLights;
  const float4       dw   = rb.ps.rayD[slot];
  const f3           rdir = xyz(dw);
  uint32_t           seed = __float_as_uint(dw.w);
  const float4       hit  = rb.ps.hit[slot];
  // depth 0: the initial values of pathtrace.glsl:201-203, which generate_ray therefore does not store (the same floats either way)
  f3                 radiance   = depth == 0 ? splat3(0.0f) : xyz(rb.ps.rad[slot]);
  f3                 throughput = depth == 0 ? splat3(1.0f) : xyz(rb.ps.thr[slot]);
  const int          dbg        = MODE >= 0 ? PT_DEBUG_NONE : st.debugging_mode;

  // ---- miss: environment (pathtrace.glsl:204-228) ----
  if(__float_as_uint(hit.y) == BVH_NONE)
  {
    f3   result;
    bool done = false;
    if(dbg != PT_DEBUG_NONE)
    {
      done = true;
      if(depth != st.maxDepth - 1)
        result = splat3(0.0f);
      else if(dbg == PT_DEBUG_RADIANCE)
        result = radiance;
      else if(dbg == PT_DEBUG_WEIGHT)
        result = throughput;
      else if(dbg == PT_DEBUG_RAYDIR)
        result = (rdir + splat3(1.0f)) * 0.5f;
      else
        done = false;
    }
    if(!done)
    {
      events |= EV_MISS;
      f3 env = useSky ? sun_and_sky(S.sunsky, rdir) : sample_env(S, spherical_uv(rdir));
      result = radiance + (env * st.hdrMultiplier * throughput);
    }
    rb.ps.rad[slot] = make_float4(result.x, result.y, result.z, 0.f);
    return SHADE_DONE;
  }

  // ---- hit ----
  events |= EV_HIT;
  // (Starting DirectLight's environment sample here -- its three draws are the first of the bounce when there are no punctual lights and no sun & sky,
  // so its two round trips could overlap the geometry / material chain -- measured 1 % SLOWER: the seven values it keeps alive across the material
  // code spill (96 -> 168 B of scratch at 128 VGPRs), profiles/r04e_*.)
  uint32_t     hitInst, hitPrim;
  int          hitMat = -2;  // material index when the shading line carries it (-2: take it from the instance record)
  VertexTriple vt;
  bool         haveVt = false;
  if(S.twoLevel)
  {  // the hit names the world triangle (store_hit): a BLAS leaf is shared by every instance of its mesh
    const uint32_t w = __float_as_uint(hit.y);
    hitInst          = instance_of_world_tri(S, w);
    hitPrim          = w - S.instTriBase[hitInst];
  }
  else
  {
    const uint32_t hslot = __float_as_uint(hit.y);
    if(S.shadeTris)
    {  // the slot's shading line: vertex attributes + (instance, primitive)
      vt              = fetch_triangle_slot(S, hslot);
      haveVt          = true;
      const float4 id = S.shadeTris[size_t(hslot) * PT_SHADE_REC_QUADS + 6];
      hitInst         = __float_as_uint(id.x);
      hitPrim         = __float_as_uint(id.y);
      hitMat          = int(__float_as_uint(id.z));  // the instance's material index: the material fetch need not wait for the instance record
    }
    else
    {
      const TriRec tr = S.tris[hslot];
      hitInst         = __float_as_uint(tr.e1n.w);
      hitPrim         = __float_as_uint(tr.e2p.w);
    }
  }
  const InstanceRec& I = S.instances[hitInst];
  Surface            sf;
  f3                 vcolor;
  if(!haveVt)
    vt = fetch_triangle(S, I, hitPrim);
  surface_at_hit(S, I, vt, hit.z, hit.w, sf, vcolor);
  const f3 hitPos = sf.position;
  sf.ffnormal     = dot3(sf.normal, rdir) <= 0.0f ? sf.normal : -sf.normal;
  const int matIndex = hitMat != -2 ? hitMat : I.materialIndex;
  resolve_material_at(S, matIndex < 0 ? 0 : matIndex, rdir, sf);
  sf.albedo *= vcolor;

  if(dbg != PT_DEBUG_NONE && dbg < PT_DEBUG_RADIANCE)  // pathtrace.glsl:61-83,255-256
  {
    f3 r = f3{1000.f, 0.f, 0.f};
    switch(dbg)
    {
      case PT_DEBUG_METALLIC: r = splat3(sf.metallic); break;
      case PT_DEBUG_NORMAL: r = (sf.normal + splat3(1.0f)) * .5f; break;
      case PT_DEBUG_BASECOLOR: r = sf.albedo; break;
      case PT_DEBUG_EMISSIVE: r = sf.emission; break;
      case PT_DEBUG_ALPHA: r = splat3(sf.alpha); break;
      case PT_DEBUG_ROUGHNESS: r = splat3(sf.roughness); break;
      case PT_DEBUG_TEXCOORD: r = f3{sf.uv.x, sf.uv.y, 0.f}; break;
      case PT_DEBUG_TANGENT: r = (sf.tangent + splat3(1.0f)) * .5f; break;
    }
    rb.ps.rad[slot] = make_float4(r.x, r.y, r.z, 0.f);
    return SHADE_DONE;
  }
  if(sf.unlit)  // KHR_materials_unlit
  {
    f3 r            = radiance + sf.albedo * throughput;
    rb.ps.rad[slot] = make_float4(r.x, r.y, r.z, 0.f);
    return SHADE_DONE;
  }

  f3 absorption = depth == 0 ? splat3(0.0f) : xyz(rb.ps.absorb[slot]);
  if(dot3(sf.normal, sf.ffnormal) > 0.0f)
    absorption = splat3(0.0f);
  radiance += sf.emission * throughput;
  throughput *= exp3(-absorption * hit.x);

  // ---- DirectLight (pathtrace.glsl:97-188): the contribution is added after the shadow ray ----
  f3    neeRadiance = splat3(0.0f), lightDir = splat3(0.0f);
  float lightDist = 1e32f;
  bool  visible   = false;
  {
    f3    lightContrib;
    float lightPdf;
    bool  isLight = false;
    float pSelect = st.hdrMultiplier > 0.0f ? 0.5f : 1.0f;
    if(nbLights != 0 && rng_next(seed) <= pSelect)
    {
      isLight            = true;
      int            li  = int(fmin2(rng_next(seed) * float(nbLights), float(nbLights)));
      li                 = li < nbLights - 1 ? li : nbLights - 1;
      const pt_Light lt  = S.lights[li];
      const f3       ldir = f3{lt.direction[0], lt.direction[1], lt.direction[2]};
      f3             pointToLight = -ldir;
      float          rangeAtt = 1.0f, spotAtt = 1.0f;
      if(lt.type != PT_LIGHT_DIRECTIONAL)
        pointToLight = f3{lt.position[0], lt.position[1], lt.position[2]} - sf.position;
      lightDist = len3(pointToLight);
      if(lt.type != PT_LIGHT_DIRECTIONAL)
        rangeAtt = range_attenuation(lt.range, lightDist);
      if(lt.type == PT_LIGHT_SPOT)
        spotAtt = spot_attenuation(pointToLight, ldir, lt.outerConeCos, lt.innerConeCos);
      lightContrib = f3{lt.color[0], lt.color[1], lt.color[2]} * (rangeAtt * spotAtt * lt.intensity);
      lightDir     = unit(pointToLight);
      lightPdf     = 1.0f;
    }
    else if(useSky)
    {
      float sunRadius = (0.00465f * 10.0f) * S.sunsky.sun_disk_scale;
      f3    sd        = f3{S.sunsky.sun_direction[0], S.sunsky.sun_direction[1], S.sunsky.sun_direction[2]};
      f3    T, B;
      make_frame(sd, T, B);
      f3 dd;
      dd.x         = rng_next(seed) * sunRadius;
      dd.y         = rng_next(seed) * sunRadius;
      dd.z         = sqrtf(fmax2(0.0f, 1.0f - dd.x * dd.x - dd.y * dd.y));
      lightDir     = unit(T * dd.x + B * dd.y + sd * dd.z);
      lightContrib = sun_and_sky(S.sunsky, lightDir);
      lightPdf     = 0.5f;
      lightContrib *= st.hdrMultiplier;
    }
    else
    {
      float a = rng_next(seed), b = rng_next(seed), c = rng_next(seed);
      events |= EV_NEE;
      lightContrib = env_importance_sample(S, f3{a, b, c}, lightDir, lightPdf);
      lightContrib *= st.hdrMultiplier;
    }
    if(dot3(lightDir, sf.ffnormal) > 0.0f)  // (state.isSubsurface is always false here: Sample() works on a copy)
    {
      float bsdfPdf = 0.0f;
      f3    f       = bsdf_eval(pbrMode, sf, -rdir, sf.ffnormal, lightDir, bsdfPdf);
      float mis     = isLight ? 1.0f : fmax2(0.0f, power_heuristic(lightPdf, bsdfPdf));
      neeRadiance   = f * mis * fabsf(dot3(lightDir, sf.ffnormal)) * lightContrib / lightPdf;
      visible       = true;
    }
  }
  neeRadiance *= throughput;

  // ---- BSDF sample ----
  f3    L;
  float pdf = 0.0f;
  f3    f   = bsdf_sample(pbrMode, sf, -rdir, sf.ffnormal, L, pdf, seed);

  if(dot3(sf.ffnormal, L) < 0.0f)
    absorption = -log3(sf.attenuationColor) / splat3(sf.attenuationDistance);

  if(pdf > 0.0f)
  {
    throughput *= f * fabsf(dot3(sf.ffnormal, L)) / pdf;
  }
  else
  {  // `break`: the path ends before its shadow ray
    rb.ps.rad[slot]    = make_float4(radiance.x, radiance.y, radiance.z, 0.f);
    rb.ps.rayD[slot].w = __uint_as_float(seed);
    return SHADE_DONE;
  }

  if(dbg != PT_DEBUG_NONE && depth == st.maxDepth - 1)
  {
    f3   r;
    bool ret = true;
    if(dbg == PT_DEBUG_RADIANCE)
      r = neeRadiance;
    else if(dbg == PT_DEBUG_WEIGHT)
      r = throughput;
    else if(dbg == PT_DEBUG_RAYDIR)
      r = (L + splat3(1.0f)) * 0.5f;
    else
      ret = false;
    if(ret)
    {
      rb.ps.rad[slot] = make_float4(r.x, r.y, r.z, 0.f);
      return SHADE_DONE;
    }
  }

  // Russian roulette probability (RR_DEPTH 0) from the updated throughput
  float rrPcont = fmin2(fmax2(throughput.x, fmax2(throughput.y, throughput.z)) * sf.eta * sf.eta + 0.001f, 0.95f);

  f3 nextO = offset_ray(hitPos, dot3(L, sf.ffnormal) > 0 ? sf.ffnormal : -sf.ffnormal);

  rb.ps.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.f);
  if(visible)
  {  // the shadow kernel adds the contribution and draws the roulette (finish_bounce_core)
    rb.ps.rayO[slot]   = make_float4(nextO.x, nextO.y, nextO.z, 0.f);
    rb.ps.rayD[slot]   = make_float4(L.x, L.y, L.z, __uint_as_float(seed));
    rb.ps.thr[slot]    = make_float4(throughput.x, throughput.y, throughput.z, rrPcont);
    rb.ps.absorb[slot] = make_float4(absorption.x, absorption.y, absorption.z, lightDist);
    rb.ps.neeDir[slot] = make_float4(lightDir.x, lightDir.y, lightDir.z, 1.f);
    rb.ps.neeRad[slot] = make_float4(neeRadiance.x, neeRadiance.y, neeRadiance.z, 0.f);
    return SHADE_TO_SHADOW;
  }
  // no shadow ray for this bounce: the Russian-roulette draw follows the BSDF draws directly (pathtrace.glsl:333-338).  A path that ends here
  // leaves only its radiance and the RNG state behind (the stream continues across the samples of a frame); the ray, throughput and
  // absorption of a bounce that is never traced are not written
  const bool die = rng_next(seed) >= rrPcont;
  if(die || depth == st.maxDepth - 1)
  {
    rb.ps.rayD[slot].w = __uint_as_float(seed);
    return SHADE_DONE;
  }
  throughput /= rrPcont;
  rb.ps.rayO[slot]   = make_float4(nextO.x, nextO.y, nextO.z, 0.f);
  rb.ps.rayD[slot]   = make_float4(L.x, L.y, L.z, __uint_as_float(seed));
  rb.ps.thr[slot]    = make_float4(throughput.x, throughput.y, throughput.z, rrPcont);
  rb.ps.absorb[slot] = make_float4(absorption.x, absorption.y, absorption.z, lightDist);
  return SHADE_TO_NEXT;
}

// ---- shadow + Russian roulette ---------------------------------------------------------------------------------
// NEE contribution if unoccluded, then Russian roulette (pathtrace.glsl:327-338); survivors go to the next bounce.
// Returns true when the path survives the roulette (the caller queues it for the next bounce, or traces its next ray right away: k_trace_p);
// `seed` comes back as the RNG state after the roulette draw (also stored in the path state).
template <class RB>
PT_DEV bool finish_bounce_core(const RB& rb, uint32_t slot, bool inShadow, uint32_t& seed)
{
  if(!inShadow)
  {
    float4 r = rb.ps.rad[slot];
    float4 c = rb.ps.neeRad[slot];
    r.x += c.x;
    r.y += c.y;
    r.z += c.z;
    rb.ps.rad[slot] = r;
  }
  float4      t   = rb.ps.thr[slot];
  const float pc  = t.w;
  const bool  die = rng_next(seed) >= pc;
  rb.ps.rayD[slot].w = __uint_as_float(seed);
  if(die)
    return false;
  t.x /= pc;
  t.y /= pc;
  t.z /= pc;
  rb.ps.thr[slot] = t;
  return true;
}
// folds the frames of the batch into the running mean of pixel slot `pslot` (firefly clamp, samples of a frame, heat-map palette)
PT_DEV void accumulate_pixel(const RenderBuffers& rb, const FrameParams& fp, uint32_t pslot)
{
  const pt_RtxState& st = fp.st;
  // the frames of the batch fold into the running mean in frame order (one thread per pixel: exact sequence)
  f3 acc = xyz(rb.frame[pslot]);
  for(uint32_t fb = 0; fb < fp.batch; ++fb)
  {
    const uint32_t slot = fb * fp.numSlots + pslot;
    f3             r    = xyz(rb.ps.rad[slot]);
    float          lum  = dot3(r, f3{0.212671f, 0.715160f, 0.072169f});
    if(lum > st.fireflyClampThreshold)
      r *= st.fireflyClampThreshold / lum;

    const bool heat = st.debugging_mode == PT_DEBUG_HEATMAP;
    float4     prev = (fp.sample == 0) ? make_float4(0.f, 0.f, 0.f, 0.f) : rb.ps.sum[slot];
    f3         sum  = xyz(prev);
    sum += r;
    const float cost = prev.w + (heat ? rb.ps.rayO[slot].w : 0.0f);  // nanoseconds this pixel's samples spent in the trace and shade kernels
    if(fp.sample + 1 < st.maxSamples)
    {
      rb.ps.sum[slot] = make_float4(sum.x, sum.y, sum.z, cost);
      continue;
    }
    f3 pixel = sum / float(st.maxSamples);
    if(heat)
    {  // pathtrace.comp:108-119
      const float low = float(st.minHeatmap), high = float(st.maxHeatmap);
      pixel           = heat_temperature(clampf((cost - low) / (high - low), 0.0f, 1.0f));
    }
    const int frame = st.frame + int(fb);
    acc             = frame > 0 ? lerp(acc, pixel, 1.0f / float(frame + 1)) : pixel;
  }
  if(fp.sample + 1 == st.maxSamples)
    rb.frame[pslot] = make_float4(acc.x, acc.y, acc.z, 1.f);
}

