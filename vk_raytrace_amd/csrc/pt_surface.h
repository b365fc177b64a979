// Surface reconstruction at a hit for the gfx950 path tracer: software texture sampling, the
// octahedral unit-vector decode, the shading frame and the glTF material resolve.
//
// Behavioural contract (file:line = reference):
//   decode_oct          shaders/compress.glsl:142-180
//   surface_at_hit      shaders/shade_state.glsl:63-145 (handedness of vertex 0 only, :114)
//   resolve_material    shaders/gltf_material.glsl:52-93,104-193 (ffnormal re-derived after normal mapping)
//   alpha_test          shaders/traceray_rq.glsl:32-102 (uv handedness bit NOT cleared, one RNG draw)
//   texture taps        Vulkan unnormalised-coordinate rules at LOD 0
#pragma once
#include "pt_device.h"

// ---- textures -------------------------------------------------------------------------------------
// Integer texel coordinate -> [0, n) per the Vulkan address modes.  AMD GPUs have no integer divide (each `%` is
// ~40 instructions), so power-of-two sizes -- flagged per texture -- wrap by masking (identical result, two's complement).
PT_DEV int wrap_index(int i, int n, int mode, bool pot)
{
  if(mode == PT_WRAP_CLAMP_TO_EDGE)
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
  if(mode == PT_WRAP_MIRRORED_REPEAT)
  {
    int p = 2 * n;
    int m = pot ? (i & (p - 1)) : (i % p);
    if(m < 0)
      m += p;
    m -= n;
    int mir = m >= 0 ? m : -(1 + m);
    return (n - 1) - mir;
  }
  if(pot)
    return i & (n - 1);
  int m = i % n;
  return m < 0 ? m + n : m;
}

PT_DEV uint32_t tex_layers(const TexRec& tr) { return ((uint32_t(tr.tiled) >> 8) & 3u) + 1u; }
PT_DEV uint32_t tex_layer(const TexRec& tr) { return (uint32_t(tr.tiled) >> 10) & 3u; }
PT_DEV f4 texel_bytes(const uint32_t* pool, const TexRec& tr, int ix, int iy)
{
  uint32_t p = pool[tr.offset + tex_index(tr.w, wrap_index(ix, tr.w, tr.wrapS, (tr.pot & 1) != 0), wrap_index(iy, tr.h, tr.wrapT, (tr.pot & 2) != 0), (tr.tiled & 1) != 0) * tex_layers(tr) + tex_layer(tr)];
  return f4{float(p & 0xffu), float((p >> 8) & 0xffu), float((p >> 16) & 0xffu), float(p >> 24)};
}

// RGBA8 texels are filtered as 0..255 floats and scaled by 1/255 once (the numerical contract both
// sides of the parity tests use; Vulkan leaves filter precision to the implementation).
PT_DEV f4 sample_rgba8_rec(const uint32_t* texels, const TexRec& tr, f2 uv)
{
  const float  s255 = 1.0f / 255.0f;
  float        x = uv.x * float(tr.w), y = uv.y * float(tr.h);
  if(tr.mag == PT_FILTER_NEAREST)
    return texel_bytes(texels, tr, (int)floorf(x), (int)floorf(y)) * s255;
  x -= 0.5f;
  y -= 0.5f;
  float fx = floorf(x), fy = floorf(y);
  float a = x - fx, b = y - fy;
  int   x0 = (int)fx, y0 = (int)fy;
  f4    top = texel_bytes(texels, tr, x0, y0) * (1.0f - a) + texel_bytes(texels, tr, x0 + 1, y0) * a;
  f4    bot = texel_bytes(texels, tr, x0, y0 + 1) * (1.0f - a) + texel_bytes(texels, tr, x0 + 1, y0 + 1) * a;
  return (top * (1.0f - b) + bot * b) * s255;
}
PT_DEV f4 sample_rgba8(const DeviceScene& S, int id, f2 uv) { return sample_rgba8_rec(S.texels, S.texRecs[id], uv); }

// Batched texture fetch (round 5: +1 % on the bench line, profiles/r05d_*): resolve_material reached its four common textures -- normal, emissive,
// metallic-roughness, base colour -- through up to eight DEPENDENT round trips (descriptor, then texels, per texture, each behind its own `if`).
// The four 16-byte descriptors are stored per material (DeviceScene::matLines, in the material's 128-byte line), so that they arrive with the material, and the texels of the
// textures a material HAS are requested up front, so that those requests are in flight together.  First forms that also fetched the absent
// textures (as texture 0) were 7 % SLOWER (profiles/r04tb_batched_texture_fetch.txt): k_shade is short of requests, not of latency.
// The tap is sample_rgba8_rec cut in two: where the texels are (tex_tap) and what is made of them (tex_filter) -- the same expressions in the same order.
struct TexTap {
  uint32_t i[4];  // texel indices into the pool: (x0,y0) (x0+1,y0) (x0,y0+1) (x0+1,y0+1); four times the nearest texel for a NEAREST tap
  float    a, b;
  bool     nearest;
};
PT_DEV TexTap tex_tap(const TexRec& tr, f2 uv)
{
  TexTap t;
  float  x = uv.x * float(tr.w), y = uv.y * float(tr.h);
  t.nearest = tr.mag == PT_FILTER_NEAREST;
  const bool     ps = (tr.pot & 1) != 0, pt = (tr.pot & 2) != 0, tiled = (tr.tiled & 1) != 0;
  const uint32_t K = tex_layers(tr), base = tr.offset + tex_layer(tr);
  if(t.nearest)
  {
    t.a = t.b = 0.0f;
    t.i[0] = t.i[1] = t.i[2] = t.i[3] = base + tex_index(tr.w, wrap_index((int)floorf(x), tr.w, tr.wrapS, ps), wrap_index((int)floorf(y), tr.h, tr.wrapT, pt), tiled) * K;
    return t;
  }
  x -= 0.5f;
  y -= 0.5f;
  const float fx = floorf(x), fy = floorf(y);
  t.a = x - fx;
  t.b = y - fy;
  const int x0 = (int)fx, y0 = (int)fy;
  const int wx0 = wrap_index(x0, tr.w, tr.wrapS, ps), wx1 = wrap_index(x0 + 1, tr.w, tr.wrapS, ps), wy0 = wrap_index(y0, tr.h, tr.wrapT, pt), wy1 = wrap_index(y0 + 1, tr.h, tr.wrapT, pt);
  t.i[0] = base + tex_index(tr.w, wx0, wy0, tiled) * K;
  t.i[1] = base + tex_index(tr.w, wx1, wy0, tiled) * K;
  t.i[2] = base + tex_index(tr.w, wx0, wy1, tiled) * K;
  t.i[3] = base + tex_index(tr.w, wx1, wy1, tiled) * K;
  return t;
}
PT_DEV f4 texel_unpack(uint32_t p) { return f4{float(p & 0xffu), float((p >> 8) & 0xffu), float((p >> 16) & 0xffu), float(p >> 24)}; }
PT_DEV f4 tex_filter(const TexTap& t, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3)
{
  const float s255 = 1.0f / 255.0f;
  if(t.nearest)
    return texel_unpack(p0) * s255;
  const f4 top = texel_unpack(p0) * (1.0f - t.a) + texel_unpack(p1) * t.a;
  const f4 bot = texel_unpack(p2) * (1.0f - t.a) + texel_unpack(p3) * t.a;
  return (top * (1.0f - t.b) + bot * t.b) * s255;
}

// Environment: RGBA32F, LINEAR, U repeat / V clamp (reference: src/hdr_sampling.cpp:68-77)
PT_DEV f3 sample_env(const DeviceScene& S, f2 uv)
{
  float x = uv.x * float(S.envW) - 0.5f, y = uv.y * float(S.envH) - 0.5f;
  float fx = floorf(x), fy = floorf(y);
  float a = x - fx, b = y - fy;
  int   x0 = (int)fx, y0 = (int)fy;
  const bool wpot = (S.envW & (S.envW - 1)) == 0;
  int   xa = wrap_index(x0, S.envW, PT_WRAP_REPEAT, wpot), xb = wrap_index(x0 + 1, S.envW, PT_WRAP_REPEAT, wpot);
  int   ya = wrap_index(y0, S.envH, PT_WRAP_CLAMP_TO_EDGE, false), yb = wrap_index(y0 + 1, S.envH, PT_WRAP_CLAMP_TO_EDGE, false);
  f3    t00 = xyz(S.env[size_t(ya) * S.envW + xa]), t10 = xyz(S.env[size_t(ya) * S.envW + xb]);
  f3    t01 = xyz(S.env[size_t(yb) * S.envW + xa]), t11 = xyz(S.env[size_t(yb) * S.envW + xb]);
  f3    top = t00 * (1.0f - a) + t10 * a;
  f3    bot = t01 * (1.0f - a) + t11 * a;
  return top * (1.0f - b) + bot * b;
}

// ---- unit-vector codec ---------------------------------------------------------------------------
PT_DEV float snorm15(int v)
{
  return (v >= 0) ? (__uint_as_float(0x3F800000u | (uint32_t(v) << 8)) - 1.0f) : (__uint_as_float(0xBF800000u | (uint32_t(-v) << 8)) + 1.0f);
}
PT_DEV f3 decode_oct(uint32_t packed)
{
  if(packed == ~0u)
    return splat3(3.402823466e+38f);
  int       x  = int(packed & 0xFFFFu) - 32767;
  int       y  = int(packed >> 16) - 32767;
  const int mx = x >> 31, my = y >> 31;
  const int t0 = 32767 + mx + my;
  const int ym = y ^ my;
  const int t1 = t0 - (x ^ mx);
  const int z  = t1 - ym;
  float     zf;
  if(z < 0)
  {
    x  = (t0 - ym) ^ mx;
    y  = t1 ^ my;
    zf = __uint_as_float(0xBF800000u | (uint32_t(-z) << 8)) + 1.0f;
  }
  else
  {
    zf = __uint_as_float(0x3F800000u | (uint32_t(z) << 8)) - 1.0f;
  }
  return unit(f3{snorm15(x), snorm15(y), zf});
}
PT_DEV f4 unpack_unorm4(uint32_t p) { return f4{float(p & 0xffu) / 255.0f, float((p >> 8) & 0xffu) / 255.0f, float((p >> 16) & 0xffu) / 255.0f, float(p >> 24) / 255.0f}; }

// ---- material after textures (reference: shaders/globals.glsl:67-97 Material + the State fields the BSDFs read)
struct Surface {
  // frame
  f3 position, normal, ffnormal, tangent, bitangent;
  f2 uv;
  // material
  f3    albedo, emission, f0, sheenTint, attenuationColor;
  float metallic, roughness, ax, ay, anisotropy, clearcoat, clearcoatRoughness, transmission, ior, eta;
  float attenuationDistance, alpha, sheen;
  float specular, specularTint, subsurface;  // constants 0.5 / 1 / 0 (gltf_material.glsl:108-112)
  bool  unlit, thinwalled;
};

struct VertexTriple {
  float4 a0, b0, a1, b1, a2, b2;  // a = pos.xyz + normal bits, b = uv.xy + tangent bits + colour bits
};
PT_DEV VertexTriple fetch_triangle(const DeviceScene& S, const InstanceRec& I, uint32_t prim)
{
  const uint32_t* t  = S.indices + I.firstIndex + 3 * size_t(prim);
  const uint32_t  i0 = I.vertexOffset + t[0], i1 = I.vertexOffset + t[1], i2 = I.vertexOffset + t[2];
  VertexTriple    v;
  v.a0 = S.vertices[size_t(i0) * 2];
  v.b0 = S.vertices[size_t(i0) * 2 + 1];
  v.a1 = S.vertices[size_t(i1) * 2];
  v.b1 = S.vertices[size_t(i1) * 2 + 1];
  v.a2 = S.vertices[size_t(i2) * 2];
  v.b2 = S.vertices[size_t(i2) * 2 + 1];
  return v;
}

// shaders/common.glsl:80-92 (== shade_state.glsl:34-39)
PT_DEV void make_frame(f3 N, f3& T, f3& B)
{
  T = unit((fabsf(N.z) > 0.99999f) ? f3{-N.x * N.y, 1.0f - N.y * N.y, -N.y * N.z} : f3{-N.x * N.z, -N.y * N.z, 1.0f - N.z * N.z});
  B = cross3(T, N);
}

// Opacity of a non-opaque candidate at barycentrics (bu,bv): baseColorFactor.a x texture alpha, thresholded
// for ALPHA_MASK (reference: shaders/traceray_rq.glsl:32-94).  Reads the flattened AlphaRec / AlphaMat records;
// the arithmetic is the reference's (interpolate raw uvs, row-vector uvTransform, bilinear tap).
// `useMap`: let the material's opacity map (pt_device.h) answer when it can -- the traversal only asks whether the
// opacity is <= 0, >= 1 or in between, so a classified block returns exactly 0 or 1 without touching the texels.
template <bool useMap>
PT_DEV float opacity_eval(const DeviceScene& S, const AlphaRec& ar, float bu, float bv)
{
  // the 80-byte record in ONE round trip: left to itself the compiler fetches the first quad, waits, looks at `tex`, and only then fetches the rest
  // (a dependent round trip more on every evaluation, inside the triangle step of the trace kernels)
  AlphaMat am;
  {
    const uint4* q  = reinterpret_cast<const uint4*>(S.alphaMats + ar.material);
    const uint4  q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : : "v"(q1.x), "v"(q2.x), "v"(q3.x), "v"(q4.x));  // (keeps the compiler from sinking these loads behind the test of q0's `tex`)
#endif
    am.factorA = __uint_as_float(q0.x); am.cutoff = __uint_as_float(q0.y); am.mode = int32_t(q0.z); am.tex = int32_t(q0.w);
    am.m[0] = __uint_as_float(q1.x); am.m[1] = __uint_as_float(q1.y); am.m[2] = __uint_as_float(q1.z); am.m[3] = __uint_as_float(q1.w);
    am.m[4] = __uint_as_float(q2.x); am.m[5] = __uint_as_float(q2.y); am.m[6] = __uint_as_float(q2.z); am.m[7] = __uint_as_float(q2.w);
    am.texOffset = q3.x; am.texW = int32_t(q3.y); am.texH = int32_t(q3.z); am.texMag = int32_t(q3.w);
    am.texWrap = int32_t(q4.x); am.mapOffset = q4.y; am._pad[0] = am._pad[1] = 0u;
  }
  float          a  = am.factorA;
  if(am.tex > -1)
  {
    const float b0 = 1.0f - bu - bv;
    f2          uv = f2{ar.uv0[0], ar.uv0[1]} * b0 + f2{ar.uv1[0], ar.uv1[1]} * bu + f2{ar.uv2[0], ar.uv2[1]} * bv;
    f2          tuv = f2{((uv.x * am.m[0] + uv.y * am.m[1]) + 1.0f * am.m[2]) + 1.0f * am.m[3], ((uv.x * am.m[4] + uv.y * am.m[5]) + 1.0f * am.m[6]) + 1.0f * am.m[7]};
    if(am.texWrap & ALPHA_FAST_TAP)
    {
      // REPEAT x REPEAT on power-of-two sizes: the arithmetic of sample_rgba8_rec on the alpha channel alone
      const bool linear = am.texMag != PT_FILTER_NEAREST;
      float      x = tuv.x * float(am.texW), y = tuv.y * float(am.texH);
      if(linear)
      {
        x -= 0.5f;
        y -= 0.5f;
      }
      const float fx = floorf(x), fy = floorf(y);
      const int   x0 = (int)fx, y0 = (int)fy;
      const int   mx = am.texW - 1, my = am.texH - 1;
      const int   xa = x0 & mx, ya = y0 & my;
      if(useMap && am.mapOffset != ALPHA_NO_MAP && fabsf(x) <= 3.0e38f && fabsf(y) <= 3.0e38f)
      {
        const uint32_t bidx = (uint32_t(ya) >> ALPHA_MAP_SHIFT) * (uint32_t(am.texW) >> ALPHA_MAP_SHIFT) + (uint32_t(xa) >> ALPHA_MAP_SHIFT);
        const uint32_t st   = (S.alphaMaps[am.mapOffset + (bidx >> 4)] >> ((bidx & 15u) * 2u)) & 3u;
        if(st != ALPHA_ST_UNKNOWN)
          return st == ALPHA_ST_ONE ? 1.0f : 0.0f;
      }
      const uint32_t* tp = S.texels + am.texOffset;
      const bool      til = (am.texWrap & ALPHA_TILED) != 0;
      float           ta;
      if(!linear)
        ta = float(tp[tex_index(am.texW, xa, ya, til)] >> 24);
      else
      {
        const int   xb = (x0 + 1) & mx, yb = (y0 + 1) & my;
        const float t00 = float(tp[tex_index(am.texW, xa, ya, til)] >> 24), t10 = float(tp[tex_index(am.texW, xb, ya, til)] >> 24);
        const float t01 = float(tp[tex_index(am.texW, xa, yb, til)] >> 24), t11 = float(tp[tex_index(am.texW, xb, yb, til)] >> 24);
        const float fa = x - fx, fb = y - fy;
        const float top = t00 * (1.0f - fa) + t10 * fa;
        const float bot = t01 * (1.0f - fa) + t11 * fa;
        ta              = top * (1.0f - fb) + bot * fb;
      }
      a *= ta * (1.0f / 255.0f);
    }
    else
    {
      TexRec tr;
      tr.offset = am.texOffset; tr.w = am.texW; tr.h = am.texH; tr.mag = am.texMag; tr.wrapS = am.texWrap & 0xff; tr.wrapT = (am.texWrap >> 8) & 0xff; tr.pot = (am.texWrap >> 16) & 3; tr.tiled = (am.texWrap & ALPHA_TILED) ? 1 : 0;
      a *= sample_rgba8_rec(S.texels, tr, tuv).w;
    }
  }
  return (am.mode == PT_ALPHA_MASK) ? (a > am.cutoff ? 1.0f : 0.0f) : a;
}
// exact value (the stochastic test of the key-ordered fallback compares a draw with it)
PT_DEV float opacity_from(const DeviceScene& S, const AlphaRec& ar, float bu, float bv) { return opacity_eval<false>(S, ar, bu, bv); }
// <= 0, >= 1 or the exact value in between (all the two-pass traversal needs)
#if defined(PT_ALPHA_ALWAYS_ONE)  // (measurement only, changes the image: every non-opaque candidate counts as fully opaque WITHOUT being evaluated -- what the
                                  // evaluation itself costs, as opposed to the rays that continue through transparent texels)
PT_DEV float opacity_class(const DeviceScene& S, const AlphaRec& ar, float bu, float bv) { return 1.0f; }
#elif defined(PT_ALPHA_EVAL_THEN_ONE)  // (measurement only, changes the image: the evaluation runs in full, its result is overruled -- against
                                      // PT_ALPHA_ALWAYS_ONE this isolates what the evaluation costs where it stands, inside the triangle step)
PT_DEV float opacity_class(const DeviceScene& S, const AlphaRec& ar, float bu, float bv)
{
  const float v = opacity_eval<true>(S, ar, bu, bv);
  return v > -1.0f ? 1.0f : v;
}
#elif defined(PT_NO_ALPHA_MAP)  // (measurement only)
PT_DEV float opacity_class(const DeviceScene& S, const AlphaRec& ar, float bu, float bv) { return opacity_eval<false>(S, ar, bu, bv); }
#else
PT_DEV float opacity_class(const DeviceScene& S, const AlphaRec& ar, float bu, float bv) { return opacity_eval<true>(S, ar, bu, bv); }
#endif
PT_DEV float hit_opacity(const DeviceScene& S, uint32_t slot, float bu, float bv) { return opacity_from(S, S.alphaRecs[slot], bu, bv); }

// Stochastic alpha (any-hit): returns true when the candidate is kept.  Draws exactly one random number
// (reference: shaders/traceray_rq.glsl:96-101).
PT_DEV bool alpha_test(const DeviceScene& S, uint32_t slot, float bu, float bv, uint32_t& seed)
{
  float opacity = hit_opacity(S, slot, bu, bv);
  return !(rng_next(seed) > opacity);
}

// Interpolated attributes -> world-space shading frame.  Returns vertex colour in `vcolor`.
// the same six float4 from the per-slot copy (DeviceScene::shadeTris, written by pt_accel.hip k_shade_tris)
PT_DEV VertexTriple fetch_triangle_slot(const DeviceScene& S, uint32_t slot)
{
  const float4* p = S.shadeTris + size_t(slot) * PT_SHADE_REC_QUADS;
  VertexTriple  v;
  v.a0 = p[0]; v.b0 = p[1]; v.a1 = p[2]; v.b1 = p[3]; v.a2 = p[4]; v.b2 = p[5];
  return v;
}
PT_DEV void surface_at_hit(const DeviceScene& S, const InstanceRec& I, const VertexTriple& v, float bu, float bv, Surface& sf, f3& vcolor)
{
  const float        b0 = 1.0f - bu - bv;
  const f3           p0 = xyz(v.a0), p1 = xyz(v.a1), p2 = xyz(v.a2);
  const f3           pos = p0 * b0 + p1 * bu + p2 * bv;
  sf.position            = xform_point(I.objectToWorld, pos);

  f3 n  = unit(decode_oct(__float_as_uint(v.a0.w)) * b0 + decode_oct(__float_as_uint(v.a1.w)) * bu + decode_oct(__float_as_uint(v.a2.w)) * bv);
  f3 wn = unit(xform_rowvec(n, I.worldToObject));
  f3 gn = unit(cross3(p1 - p0, p2 - p0));
  f3 wg = unit(xform_rowvec(gn, I.worldToObject));

  float h0 = (__float_as_int(v.b0.y) & 1) == 1 ? 1.0f : -1.0f;
  f3    tg = decode_oct(__float_as_uint(v.b0.z)) * b0 + decode_oct(__float_as_uint(v.b1.z)) * bu + decode_oct(__float_as_uint(v.b2.z)) * bv;
  tg       = unit(tg);
  f3 wt    = unit(xform_dir(I.objectToWorld, tg));
  wt       = unit(wt - wn * dot3(wt, wn));
  f3 wb    = cross3(wn, wt) * h0;

  auto clear_lsb = [](float y) { return __uint_as_float(__float_as_uint(y) & ~1u); };
  sf.uv          = f2{v.b0.x, clear_lsb(v.b0.y)} * b0 + f2{v.b1.x, clear_lsb(v.b1.y)} * bu + f2{v.b2.x, clear_lsb(v.b2.y)} * bv;

  f4 col = unpack_unorm4(__float_as_uint(v.b0.w)) * b0 + unpack_unorm4(__float_as_uint(v.b1.w)) * bu + unpack_unorm4(__float_as_uint(v.b2.w)) * bv;
  vcolor = f3{col.x, col.y, col.z};

  if(dot3(wn, wg) <= 0)
    wn *= -1.0f;
  sf.normal    = wn;
  sf.tangent   = wt;
  sf.bitangent = wb;
}

PT_DEV f4 srgb_to_linear(f4 c)
{
  f3 l = pow3(f3{c.x, c.y, c.z}, 2.2f);
  return f4{l.x, l.y, l.z, c.w};
}

// glTF material + KHR extensions -> Surface (everything the BSDFs need).  `rayDir` is the incoming ray.
// md: the material's four texture descriptors (quads 3-6 of its line)
PT_DEV void resolve_material(const DeviceScene& S, const pt_GltfShadeMaterial& m, const uint4* md, f3 rayDir, Surface& sf)
{
  sf.specular     = 0.5f;
  sf.subsurface   = 0.0f;
  sf.specularTint = 1.0f;

  const float* um = m.uvTransform;
  sf.uv           = f2{((sf.uv.x * um[0] + sf.uv.y * um[1]) + 1.0f * um[2]) + 1.0f * um[3], ((sf.uv.x * um[4] + sf.uv.y * um[5]) + 1.0f * um[6]) + 1.0f * um[7]};
  const f3 T0 = sf.tangent, B0 = sf.bitangent, N0 = sf.normal;  // TBN before normal mapping

  const bool      hasN = m.normalTexture > -1, hasE = m.emissiveTexture > -1, hasM = m.pbrMetallicRoughnessTexture > -1, hasB = m.pbrBaseColorTexture > -1;
  const uint4     dN = md[0], dE = md[1], dM = md[2], dB = md[3];
  const uint32_t* tx = S.texels;
  // each texture behind its `if`, but nothing waits inside the blocks
  TexTap   tN{}, tE{}, tM{}, tB{};
  uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  if(hasN) { tN = tex_tap(tex_desc_unpack(dN), sf.uv); n0 = tx[tN.i[0]]; n1 = tx[tN.i[1]]; n2 = tx[tN.i[2]]; n3 = tx[tN.i[3]]; }
  if(hasE) { tE = tex_tap(tex_desc_unpack(dE), sf.uv); e0 = tx[tE.i[0]]; e1 = tx[tE.i[1]]; e2 = tx[tE.i[2]]; e3 = tx[tE.i[3]]; }
  if(hasM) { tM = tex_tap(tex_desc_unpack(dM), sf.uv); m0 = tx[tM.i[0]]; m1 = tx[tM.i[1]]; m2 = tx[tM.i[2]]; m3 = tx[tM.i[3]]; }
  if(hasB) { tB = tex_tap(tex_desc_unpack(dB), sf.uv); b0 = tx[tB.i[0]]; b1 = tx[tB.i[1]]; b2 = tx[tB.i[2]]; b3 = tx[tB.i[3]]; }
#define PT_TAP_N tex_filter(tN, n0, n1, n2, n3)
#define PT_TAP_E tex_filter(tE, e0, e1, e2, e3)
#define PT_TAP_M tex_filter(tM, m0, m1, m2, m3)
#define PT_TAP_B tex_filter(tB, b0, b1, b2, b3)
  if(m.normalTexture > -1)
  {
    f3 nv       = xyz(PT_TAP_N);
    nv          = unit(nv * 2.0f - 1.0f);
    nv          = nv * f3{m.normalTextureScale, m.normalTextureScale, 1.0f};
    sf.normal   = unit(basis_mul(T0, B0, N0, nv));
    sf.ffnormal = dot3(sf.normal, rayDir) <= 0.0f ? sf.normal : -sf.normal;
    make_frame(sf.ffnormal, sf.tangent, sf.bitangent);
  }

  sf.emission = f3{m.emissiveFactor[0], m.emissiveFactor[1], m.emissiveFactor[2]};
  if(m.emissiveTexture > -1)
    sf.emission *= xyz(srgb_to_linear(PT_TAP_E));

  // metallic-roughness (gltf_material.glsl:52-93)
  float dielectricSpecular = (m.ior - 1) / (m.ior + 1);
  dielectricSpecular *= dielectricSpecular;
  float rough = m.pbrRoughnessFactor, metal = m.pbrMetallicFactor;
  if(m.pbrMetallicRoughnessTexture > -1)
  {
    f4 mr = PT_TAP_M;
    rough = mr.y * rough;
    metal = mr.z * metal;
  }
  f4 base = f4{m.pbrBaseColorFactor[0], m.pbrBaseColorFactor[1], m.pbrBaseColorFactor[2], m.pbrBaseColorFactor[3]};
  if(m.pbrBaseColorTexture > -1)
    base = base * srgb_to_linear(PT_TAP_B);
#undef PT_TAP_N
#undef PT_TAP_E
#undef PT_TAP_M
#undef PT_TAP_B
  sf.f0        = lerp(splat3(dielectricSpecular), f3{base.x, base.y, base.z}, metal);
  sf.albedo    = f3{base.x, base.y, base.z};
  sf.metallic  = metal;
  sf.alpha     = base.w;
  sf.roughness = fmax2(rough, 0.001f);

  sf.transmission = m.transmissionFactor;
  if(m.transmissionTexture > -1)
    sf.transmission *= sample_rgba8(S, m.transmissionTexture, sf.uv).x;

  sf.ior = m.ior;
  sf.eta = dot3(sf.normal, sf.ffnormal) > 0.0f ? (1.0f / sf.ior) : sf.ior;

  sf.unlit      = (m.unlit == 1);
  sf.anisotropy = m.anisotropy;
  float aspect  = sqrtf(1.0f - m.anisotropy * 0.9f);
  sf.ax         = fmax2(0.001f, sf.roughness / aspect);
  sf.ay         = fmax2(0.001f, sf.roughness * aspect);
  if(m.anisotropy > 0)
  {
    sf.tangent   = unit(basis_mul(T0, B0, N0, f3{m.anisotropyDirection[0], m.anisotropyDirection[1], m.anisotropyDirection[2]}));
    sf.bitangent = unit(cross3(sf.normal, sf.tangent));
  }

  sf.attenuationColor    = f3{m.attenuationColor[0], m.attenuationColor[1], m.attenuationColor[2]};
  sf.attenuationDistance = m.attenuationDistance;
  sf.thinwalled          = m.thicknessFactor == 0;

  sf.clearcoat          = m.clearcoatFactor;
  sf.clearcoatRoughness = m.clearcoatRoughness;
  if(m.clearcoatTexture > -1)
    sf.clearcoat *= sample_rgba8(S, m.clearcoatTexture, sf.uv).x;
  if(m.clearcoatRoughnessTexture > -1)
    sf.clearcoatRoughness *= sample_rgba8(S, m.clearcoatRoughnessTexture, sf.uv).y;
  sf.clearcoatRoughness = fmax2(sf.clearcoatRoughness, 0.001f);

  f4 sh        = unpack_unorm4(m.sheen);
  sf.sheenTint = f3{sh.x, sh.y, sh.z};
  sf.sheen     = sh.w;
}

// The material of a hit, by index: from its 128-byte line alone when the material is MAT_SIMPLE (pt_device.h), else from the full record.
PT_DEV void resolve_material_at(const DeviceScene& S, int matIndex, f3 rayDir, Surface& sf)
{
  const uint4* line = S.matLines + size_t(matIndex) * PT_MAT_LINE_QUADS;
  const uint4  q0 = line[0], q1 = line[1], q2 = line[2];
  const uint4  md[4] = {line[3], line[4], line[5], line[6]};
  if(q2.w & MAT_SIMPLE)
  {
    pt_GltfShadeMaterial m;
    m.pbrBaseColorFactor[0] = __uint_as_float(q0.x); m.pbrBaseColorFactor[1] = __uint_as_float(q0.y); m.pbrBaseColorFactor[2] = __uint_as_float(q0.z); m.pbrBaseColorFactor[3] = __uint_as_float(q0.w);
    m.emissiveFactor[0] = __uint_as_float(q1.x); m.emissiveFactor[1] = __uint_as_float(q1.y); m.emissiveFactor[2] = __uint_as_float(q1.z);
    m.normalTextureScale = __uint_as_float(q1.w);
    m.pbrRoughnessFactor = __uint_as_float(q2.x); m.pbrMetallicFactor = __uint_as_float(q2.y); m.ior = __uint_as_float(q2.z);
    m.normalTexture = (q2.w & MAT_HAS_NORMAL) ? 0 : -1; m.emissiveTexture = (q2.w & MAT_HAS_EMISSIVE) ? 0 : -1;
    m.pbrMetallicRoughnessTexture = (q2.w & MAT_HAS_MR) ? 0 : -1; m.pbrBaseColorTexture = (q2.w & MAT_HAS_BASE) ? 0 : -1;
    // what mat_is_simple checked, as constants
    for(int k = 0; k < 16; ++k)
      m.uvTransform[k] = 0.0f;
    m.uvTransform[0] = 1.0f; m.uvTransform[5] = 1.0f;
    m.unlit = 0; m.transmissionFactor = 0.0f; m.transmissionTexture = -1;
    m.anisotropy = 0.0f; m.anisotropyDirection[0] = 0.0f; m.anisotropyDirection[1] = 1.0f; m.anisotropyDirection[2] = 0.0f;
    m.attenuationColor[0] = m.attenuationColor[1] = m.attenuationColor[2] = 1.0f;
    m.thicknessFactor = 0.0f; m.thicknessTexture = -1; m.attenuationDistance = 3.4028235e38f;
    m.clearcoatFactor = 0.0f; m.clearcoatRoughness = 0.0f; m.clearcoatTexture = -1; m.clearcoatRoughnessTexture = -1; m.sheen = 0u;
    m.alphaMode = 0; m.alphaCutoff = 0.0f; m.doubleSided = 0; m._pad0 = 0; m._pad1 = 0;  // (not read by the shading)
    resolve_material(S, m, md, rayDir, sf);
  }
  else
    resolve_material(S, S.materials[matIndex], md, rayDir, sf);
}
