// TEST INFRASTRUCTURE -- the CPU oracle's C entry points (see oracle/README.md).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
// only as the checker / the reported CPU baseline.  The product (libptmi.so) never links or calls it.
//
// PARITY PINNED to the reference's own code: oracle/_ref/libref.so (the reference's shaders and host sources compiled where they
// lie by the recipe oracle/ref_glue/) agrees with this restatement bit for bit -- tests/test_oracle_vs_ref.py, and through the
// fixtures it mints, tests/test_golden.py.  Unpinned remains only what the reference has no code for (BVH build / traversal /
// ray-triangle test: the Vulkan driver; fixed here as the trace contract T1-T6 of orc_scene.h).
#include <omp.h>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include "orc_path.h"

using namespace orc;

int orc::g_math_mode = 0;

// post.frag + tonemapping.glsl -------------------------------------------------------------------
namespace {

// shaders/tonemapping.glsl:29-32
inline vec3 linearTosRGB(vec3 c) { return gpow(c, vec3(1.0f / 2.2f)); }
// :36-39
inline vec3 sRGBToLinear(vec3 c) { return gpow(c, vec3(2.2f)); }
// :48-57
inline vec3 toneMapUncharted2Impl(vec3 color)
{
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  return ((color * (color * A + C * B) + D * E) / (color * (color * A + B) + D * F)) - E / F;
}
// :59-65
inline vec3 toneMapUncharted(vec3 color)
{
  const float W   = 11.2f;
  color           = toneMapUncharted2Impl(color * 2.0f);
  vec3 whiteScale = vec3(1.0f) / toneMapUncharted2Impl(vec3(W));
  return linearTosRGB(color * whiteScale);
}
// shaders/post.frag:48-54
inline vec3 dither(vec3 linear_color, vec3 noise, float quant)
{
  vec3 c0    = gfloor(linearTosRGB(linear_color) / quant) * quant;
  vec3 c1    = c0 + quant;
  vec3 discr = gmix(sRGBToLinear(c0), sRGBToLinear(c1), noise);
  return vec3(discr.x < linear_color.x ? c1.x : c0.x, discr.y < linear_color.y ? c1.y : c0.y, discr.z < linear_color.z ? c1.z : c0.z);
}
// shaders/post.frag:64-70 (RGB2XYZ is a column-major mat3 constructor: XYZ.y = dot(row 1))
inline vec3 toneExposure(vec3 RGB, float logAvgLum, const pt_Tonemapper& tm)
{
  // mat3(0.4124564, 0.3575761, 0.1804375, 0.2126729, 0.7151522, 0.0721750, 0.0193339, 0.1191920, 0.9503041) * RGB:
  // columns are (0.4124564,0.3575761,0.1804375) ... so component y = 0.3575761*R + 0.7151522*G + 0.1191920*B
  float XYZy = (0.3575761f * RGB.x + 0.7151522f * RGB.y) + 0.1191920f * RGB.z;
  float Y    = (tm.key / logAvgLum) * XYZy;
  float Yd   = (Y * (1.0f + Y / (tm.Ywhite * tm.Ywhite))) / (1.0f + Y);
  return RGB / XYZy * Yd;
}

}  // namespace

struct orc_ctx {
  Scene       scene;
  Stats       stats;
  int         threads = 0;
  int         variant = 0;  // 0: ray-query path, 1: RT-pipeline path (orc_path.h Tracer::variant)
  std::string err;
};

extern "C" {

orc_ctx* orc_create() { return new orc_ctx(); }
void     orc_destroy(orc_ctx* c) { delete c; }
const char* orc_last_error(orc_ctx* c) { return c->err.c_str(); }

int orc_set_threads(orc_ctx* c, int n)
{
  c->threads = n;
  return 0;
}
// 0: fp32 libm (default), 1: double-precision functions rounded to fp32 (noise-floor calibration)
int orc_set_math_mode(int mode)
{
  g_math_mode = mode;
  return 0;
}
int orc_set_variant(orc_ctx* c, int variant)
{
  c->variant = variant ? 1 : 0;
  return 0;
}
int orc_use_any_hit(orc_ctx* c, int enable)
{
  c->scene.anyHit = enable != 0;
  if(!c->scene.nodes.empty())
    c->scene.build_world();
  return 0;
}
int orc_set_use_bvh(orc_ctx* c, int use)
{
  c->scene.useBvh = use != 0;
  return 0;
}

int orc_set_scene(orc_ctx* c, const pt_SceneDesc* d)
{
  if(!d || !c->scene.set(d))
  {
    c->err = "invalid scene description";
    return -1;
  }
  return 0;
}

int orc_set_env(orc_ctx* c, const float* rgba, int w, int h, float* integral, float* average)
{
  if(!rgba || w <= 0 || h <= 0)
    return -1;
  c->scene.env.assign(rgba, rgba + size_t(w) * h * 4);
  c->scene.envW = w;
  c->scene.envH = h;
  create_environment_accel(rgba, (uint32_t)w, (uint32_t)h, c->scene.envAccel, c->scene.envIntegral, c->scene.envAverage);
  if(integral) *integral = c->scene.envIntegral;
  if(average) *average = c->scene.envAverage;
  return 0;
}
int orc_set_camera(orc_ctx* c, const pt_SceneCamera* cam)
{
  c->scene.camera = *cam;
  return 0;
}
int orc_set_sunsky(orc_ctx* c, const pt_SunAndSky* ss)
{
  c->scene.sunsky = *ss;
  return 0;
}

// Renders one frame (state->maxSamples samples per pixel) into accum (row-major W*H RGBA32F, in/out).
// If pixel_ids != NULL only those n pixels (id = y*W + x) are rendered (bounded CPU-baseline sample).
int orc_render_frame(orc_ctx* c, const pt_RtxState* state, float* accum, const uint32_t* pixel_ids, uint64_t n)
{
  const int W = state->size[0], H = state->size[1];
  if(W <= 0 || H <= 0 || !accum)
    return -1;
  if(c->scene.sunsky.in_use != 1 && c->scene.env.empty())
  {
    c->err = "no environment set";
    return -1;
  }
  const int nthreads = c->threads > 0 ? c->threads : omp_get_max_threads();
  Stats     total;
#pragma omp parallel num_threads(nthreads)
  {
    Tracer tr(c->scene, *state, c->variant);
    if(pixel_ids)
    {
#pragma omp for schedule(dynamic, 64)
      for(int64_t i = 0; i < (int64_t)n; ++i)
      {
        uint32_t id = pixel_ids[i];
        tr.render_pixel(int(id % W), int(id / W), accum + size_t(id) * 4);
      }
    }
    else
    {
      const int tx = (W + 7) / 8, ty = (H + 7) / 8;
#pragma omp for schedule(dynamic, 4)
      for(int t = 0; t < tx * ty; ++t)
      {
        int x0 = (t % tx) * 8, y0 = (t / tx) * 8;
        for(int y = y0; y < std::min(y0 + 8, H); ++y)
          for(int x = x0; x < std::min(x0 + 8, W); ++x)
            tr.render_pixel(x, y, accum + (size_t(y) * W + x) * 4);
      }
    }
#pragma omp critical
    total.add(tr.stats);
  }
  c->stats.add(total);
  return 0;
}

// `nframes` consecutive frames (frame = first_frame ...) of the listed pixels inside ONE thread team (bench.py's cpu_baseline leg: a fork / join
// of a few hundred threads per frame would be most of a small frame's time).  Per pixel the frames are folded in frame order, as always.
int orc_render_frames(orc_ctx* c, const pt_RtxState* state, int first_frame, int nframes, float* accum, const uint32_t* pixel_ids, uint64_t n)
{
  const int W = state->size[0], H = state->size[1];
  if(W <= 0 || H <= 0 || !accum || !pixel_ids)
    return -1;
  if(c->scene.sunsky.in_use != 1 && c->scene.env.empty())
  {
    c->err = "no environment set";
    return -1;
  }
  const int nthreads = c->threads > 0 ? c->threads : omp_get_max_threads();
  Stats     total;
#pragma omp parallel num_threads(nthreads)
  {
    Stats mine;
    for(int f = first_frame; f < first_frame + nframes; ++f)
    {
      pt_RtxState st = *state;
      st.frame       = f;
      Tracer tr(c->scene, st, c->variant);
#pragma omp for schedule(dynamic, 16)
      for(int64_t i = 0; i < (int64_t)n; ++i)
      {
        uint32_t id = pixel_ids[i];
        tr.render_pixel(int(id % W), int(id / W), accum + size_t(id) * 4);
      }  // (implicit barrier: frame f is complete for every pixel before f + 1 starts)
      mine.add(tr.stats);
    }
#pragma omp critical
    total.add(mine);
  }
  c->stats.add(total);
  return 0;
}

// 12 uint64: samples closestRays shadowRays shadedHits misses alphaTests neeLookups nodesVisited trisTested texTaps nodesShadow trisShadow
int orc_get_stats(orc_ctx* c, uint64_t* out)
{
  const Stats& s = c->stats;
  uint64_t     v[12] = {s.samples, s.closestRays, s.shadowRays, s.shadedHits, s.misses, s.alphaTests, s.neeLookups, s.nodesVisited, s.trisTested, s.texTaps, s.nodesShadow, s.trisShadow};
  std::memcpy(out, v, sizeof(v));
  return 0;
}
int orc_reset_stats(orc_ctx* c)
{
  c->stats = Stats();
  return 0;
}
uint32_t orc_num_triangles(orc_ctx* c) { return (uint32_t)c->scene.tris.size(); }

// ---- the offscreen image as post.frag sees it ---------------------------------------------------------------------------------
// RenderOutput::genMipmap (src/render_output.cpp:188-193 -> nvvk::cmdGenerateMipmaps): level i = vkCmdBlitImage(VK_FILTER_LINEAR) of
// level i-1, extent max(1, e/2), floor(log2(max(w,h))) + 1 levels.  Blit per Vulkan 1.3 "Image Copies with Scaling": the centre of the
// dst texel scaled into src space, unnormalised linear filtering, clamp-to-edge; horizontal lerps first, then the vertical one.
static void blit_linear(const float* src, int sw, int sh, float* dst, int dw, int dh)
{
  const float su = float(sw) / float(dw), sv = float(sh) / float(dh);
#pragma omp parallel for schedule(static) if(dw * dh > 4096)
  for(int y = 0; y < dh; ++y)
    for(int x = 0; x < dw; ++x)
    {
      float u = (float(x) + 0.5f) * su - 0.5f, v = (float(y) + 0.5f) * sv - 0.5f;
      float fu = std::floor(u), fv = std::floor(v);
      float a = u - fu, b = v - fv;
      int   x0 = std::min(std::max((int)fu, 0), sw - 1), x1 = std::min(std::max((int)fu + 1, 0), sw - 1);
      int   y0 = std::min(std::max((int)fv, 0), sh - 1), y1 = std::min(std::max((int)fv + 1, 0), sh - 1);
      for(int k = 0; k < 4; ++k)
      {
        float t00 = src[(size_t(y0) * sw + x0) * 4 + k], t10 = src[(size_t(y0) * sw + x1) * 4 + k];
        float t01 = src[(size_t(y1) * sw + x0) * 4 + k], t11 = src[(size_t(y1) * sw + x1) * 4 + k];
        float top = t00 * (1.0f - a) + t10 * a, bot = t01 * (1.0f - a) + t11 * a;
        dst[(size_t(y) * dw + x) * 4 + k] = top * (1.0f - b) + bot * b;
      }
    }
}
struct MipImage {
  std::vector<std::vector<float>> level;
  std::vector<int>                w, h;
  // the sampler RenderOutput creates (render_output.cpp:98-100, a zeroed VkSamplerCreateInfo): NEAREST texel, NEAREST mip, REPEAT
  vec4 fetch(vec2 uv, int lod) const
  {
    lod = std::min(std::max(lod, 0), (int)level.size() - 1);
    int i = (int)std::floor(uv.x * float(w[lod])), j = (int)std::floor(uv.y * float(h[lod]));
    i %= w[lod]; if(i < 0) i += w[lod];
    j %= h[lod]; if(j < 0) j += h[lod];
    const float* p = &level[lod][(size_t(j) * w[lod] + i) * 4];
    return vec4(p[0], p[1], p[2], p[3]);
  }
};
static void build_mips(MipImage& im, bool chain)
{
  if(!chain)
    return;
  for(int m = std::max(im.w[0], im.h[0]); m > 1; m >>= 1)
  {
    int pw = im.w.back(), ph = im.h.back();
    int nw = pw > 1 ? pw / 2 : 1, nh = ph > 1 ? ph / 2 : 1;
    std::vector<float> d(size_t(nw) * nh * 4);
    blit_linear(im.level.back().data(), pw, ph, d.data(), nw, nh);
    im.level.push_back(std::move(d));
    im.w.push_back(nw);
    im.h.push_back(nh);
  }
}
// level `lod` of the chain (test access)
int orc_mip_chain(const float* rgba, int W, int H, int lod, float* out, int* outW, int* outH)
{
  MipImage im;
  im.level.push_back(std::vector<float>(rgba, rgba + size_t(W) * H * 4));
  im.w.push_back(W); im.h.push_back(H);
  build_mips(im, true);
  if(lod >= 0 && lod < (int)im.level.size())
  {
    *outW = im.w[lod]; *outH = im.h[lod];
    if(out) std::memcpy(out, im.level[lod].data(), im.level[lod].size() * 4);
  }
  return (int)im.level.size();
}

// shaders/post.frag:98-147 with TONEMAP_UNCHARTED on a dispW x dispH viewport.  The offscreen image has the viewport's size and holds the
// w x h accumulation image in its top-left corner (w = dispW / descalingLevel while navigating, src/sample_example.cpp:410-413; texels
// outside are zero here -- the reference keeps whatever an earlier frame left there); tm->zoom = 1 / descalingLevel (:378).
// out8: RGBA8 (floor(v * 255 + 0.5)); outF (may be NULL): fragColor as floats.
int orc_tonemap_zoom(const pt_Tonemapper* tm, const float* accum, int w, int h, int dispW, int dispH, uint8_t* out8, float* outF)
{
  if(!tm || !accum || w <= 0 || h <= 0 || dispW < w || dispH < h)
    return -1;
  MipImage im;
  im.level.emplace_back(size_t(dispW) * dispH * 4, 0.0f);
  im.w.push_back(dispW); im.h.push_back(dispH);
  for(int y = 0; y < h; ++y)
    std::memcpy(&im.level[0][size_t(y) * dispW * 4], accum + size_t(y) * w * 4, size_t(w) * 16);
  build_mips(im, (tm->autoExposure & 1) != 0);  // sample_example.cpp:423-427: the chain is only generated with auto-exposure on
  auto luminance = [](vec3 c) { return dot(c, vec3(0.2126f, 0.7152f, 0.0722f)); };
#pragma omp parallel for schedule(static)
  for(int y = 0; y < dispH; ++y)
  {
    for(int x = 0; x < dispW; ++x)
    {
      vec2 uvCoords((float(x) + 0.5f) / float(dispW), (float(y) + 0.5f) / float(dispH));  // passthrough.vert interpolated at the pixel centre
      vec2 uvz = uvCoords * tm->zoom;
      vec4 hdr4 = im.fetch(uvz, 0);  // post.frag:101
      vec3 hdr  = hdr4.xyz();
      if(((tm->autoExposure >> 0) & 1) == 1)
      {
        vec4  avg     = im.fetch(vec2(0.5f, 0.5f), 20);  // :105 -- lod 20 clamps to the 1x1 level
        float avgLum2 = luminance(avg.xyz());
        // :66 RGB2XYZ is a column-major mat3 constructor: XYZ.y = 0.3575761 R + 0.7151522 G + 0.1191920 B
        float XYZy = (0.3575761f * hdr.x + 0.7151522f * hdr.y) + 0.1191920f * hdr.z;
        float Y    = (tm->key / avgLum2) * XYZy;
        float Yd;
        if(((tm->autoExposure >> 1) & 1) == 1)
        {
          // toneLocalExposure :72-96
          float       La = 0.0f;
          float       factor = tm->key / avgLum2;
          float       epsilon = 0.05f, phi = 2.0f;
          const float scale[7] = {1, 2, 4, 8, 16, 32, 64};
          for(int i = 0; i < 7; ++i)
          {
            float v1 = luminance(im.fetch(uvz, i).xyz()) * factor;
            float v2 = luminance(im.fetch(uvz, i + 1).xyz()) * factor;
            if(std::fabs(v1 - v2) / ((tm->key * mpow(2.0f, phi) / (scale[i] * scale[i])) + v1) > epsilon)
            {
              La = v1;
              break;
            }
            else
              La = v2;
          }
          Yd = Y / (1.0f + La);
        }
        else
          Yd = (Y * (1.0f + Y / (tm->Ywhite * tm->Ywhite))) / (1.0f + Y);  // toneExposure :64-70
        hdr = hdr / XYZy * Yd;
      }
      vec3 color = toneMapUncharted(hdr * tm->avgLum);
      if(tm->dither > 0)
      {
        uint32_t r[3] = {(uint32_t)(float(x) + 0.5f), (uint32_t)(float(y) + 0.5f), 0u};  // uvec3(gl_FragCoord.xy, 0)
        pcg3d(r);
        vec3 noise(uintBitsToFloat(0x3f800000u | (r[0] >> 9)) - 1.0f, uintBitsToFloat(0x3f800000u | (r[1] >> 9)) - 1.0f,
                   uintBitsToFloat(0x3f800000u | (r[2] >> 9)) - 1.0f);
        color = dither(sRGBToLinear(color), noise, 1.f / 255.f);
      }
      color      = gclamp(gmix(vec3(0.5f), color, tm->contrast), 0.0f, 1.0f);
      color      = gpow(color, vec3(1.0f / tm->brightness));
      vec3  i    = vec3(dot(color, vec3(0.299f, 0.587f, 0.114f)));
      color      = gmix(i, color, tm->saturation);
      vec2  uv   = ((uvCoords * vec2(tm->renderingRatio[0], tm->renderingRatio[1])) - vec2(0.5f)) * 2.0f;
      color *= 1.0f - dot(uv, uv) * tm->vignette;
      const size_t o = (size_t(y) * dispW + x) * 4;
      if(outF)
      {
        outF[o] = color.x; outF[o + 1] = color.y; outF[o + 2] = color.z; outF[o + 3] = hdr4.w;
      }
      if(out8)
      {
        for(int k = 0; k < 3; ++k)
          out8[o + k] = (uint8_t)std::floor(gclamp(color[k], 0.0f, 1.0f) * 255.0f + 0.5f);
        out8[o + 3] = (uint8_t)std::floor(gclamp(hdr4.w, 0.0f, 1.0f) * 255.0f + 0.5f);
      }
    }
  }
  return 0;
}
int orc_tonemap(const pt_Tonemapper* tm, const float* accum, int W, int H, uint8_t* out) { return orc_tonemap_zoom(tm, accum, W, H, W, H, out, nullptr); }

// ---- small known-answer entry points ------------------------------------------------------------
uint32_t orc_tea(uint32_t a, uint32_t b) { return tea(a, b); }
void     orc_pcg_stream(uint32_t seed, uint32_t n, uint32_t* out_words, float* out_floats, uint32_t* out_state)
{
  for(uint32_t i = 0; i < n; ++i)
  {
    uint32_t s = seed;
    uint32_t w = pcg(s);
    if(out_words) out_words[i] = w;
    if(out_floats) out_floats[i] = uintBitsToFloat(0x3f800000u | (w >> 9)) - 1.0f;
    seed = s;
  }
  if(out_state) *out_state = seed;
}
void orc_pcg3d(uint32_t* v) { pcg3d(v); }
uint32_t orc_compress_unit_vec(const float* v) { return compress_unit_vec(vec3(v[0], v[1], v[2])); }
void     orc_decompress_unit_vec(uint32_t p, float* out)
{
  vec3 v = decompress_unit_vec(p);
  out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
void orc_offset_ray(const float* p, const float* n, float* out)
{
  vec3 v = OffsetRay(vec3(p[0], p[1], p[2]), vec3(n[0], n[1], n[2]));
  out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
void orc_pack_vertices(uint32_t n, const float* pos, const float* nrm, const float* tan4, const float* uv, const float* col4, pt_VertexAttributes* out)
{
  for(uint32_t i = 0; i < n; ++i)
    pack_vertex(pos + 3 * i, nrm + 3 * i, tan4 + 4 * i, uv + 2 * i, col4 + 4 * i, out + i);
}
void orc_camera_lookat(const float* eye, const float* center, const float* up, float fov, float aspect, pt_SceneCamera* out)
{
  camera_lookat(eye, center, up, fov, aspect, out);
}
void orc_build_env_accel(const float* rgba, int w, int h, pt_EnvAccel* out, float* integral, float* average)
{
  std::vector<pt_EnvAccel> acc;
  create_environment_accel(rgba, (uint32_t)w, (uint32_t)h, acc, *integral, *average);
  std::memcpy(out, acc.data(), acc.size() * sizeof(pt_EnvAccel));
}
void orc_sampler_from_gltf(int has, int mag, int min, int ws, int wt, pt_TextureDesc* io) { sampler_from_gltf(has, mag, min, ws, wt, io); }
void orc_sun_and_sky(const pt_SunAndSky* ss, const float* dir, float* out)
{
  vec3 c = sky::sun_and_sky(*ss, vec3(dir[0], dir[1], dir[2]));
  out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void orc_sample_texture(orc_ctx* c, int id, float u, float v, float* out)
{
  vec4 t = c->scene.sample_texture(id, vec2(u, v), nullptr);
  out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
}

// ---- the "driver" side of oracle/_ref (oracle/ref_glue/ref_driver.h RefHooks): the reference's shader code compiled from its own
// sources calls back here for what the Vulkan driver would supply -- triangle candidates in the order of the trace contract
// (T1-T6), the instance matrices, the Appendix-F samplers.  `user` is the orc_ctx.
int orc_hook_query(void* user, const float* o, const float* d, float tmax, float tPrev, uint32_t wPrev, int want, float* t, float* u, float* v, uint32_t* w)
{
  const Scene& sc = static_cast<orc_ctx*>(user)->scene;
  Candidate    c  = sc.query(vec3(o[0], o[1], o[2]), vec3(d[0], d[1], d[2]), tmax, tPrev, wPrev, want, nullptr);
  if(!c.found)
    return 0;
  *t = c.t; *u = c.u; *v = c.v; *w = c.w;
  return 1;
}
void orc_hook_tri_info(void* user, uint32_t w, int* node, int* prim, int* custom, int* opaque)
{
  const Scene&    sc = static_cast<orc_ctx*>(user)->scene;
  const WorldTri& tr = sc.tris[w];
  *node   = (int)tr.node;
  *prim   = (int)tr.prim;
  *custom = sc.nodes[tr.node].primMesh;
  *opaque = (tr.flags & TRI_OPAQUE) ? 1 : 0;
}
void orc_hook_instance(void* user, int node, float* o2w12, float* w2o12)
{
  const Scene& sc = static_cast<orc_ctx*>(user)->scene;
  for(int c = 0; c < 4; ++c)
    for(int k = 0; k < 3; ++k)
    {
      o2w12[3 * c + k] = sc.objectToWorld[node].c[c][k];
      w2o12[3 * c + k] = sc.worldToObject[node].c[c][k];
    }
}
void orc_hook_sample_texture(void* user, int id, float u, float v, float* out)
{
  vec4 t = static_cast<orc_ctx*>(user)->scene.sample_texture(id, vec2(u, v), nullptr);
  out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
}
void orc_hook_sample_env(void* user, float u, float v, float* out)
{
  vec3 t = static_cast<orc_ctx*>(user)->scene.sample_env(vec2(u, v));
  out[0] = t.x; out[1] = t.y; out[2] = t.z;
}
const pt_EnvAccel* orc_env_accel(orc_ctx* c) { return c->scene.envAccel.data(); }

// ---- function-level probes with the signatures of oracle/ref_glue/ref_comp.cpp (compared bit for bit in tests/test_oracle_vs_ref.py)
void orc_temperature(float x, float* o)
{
  vec3 c = temperature(x);
  o[0] = c.x; o[1] = c.y; o[2] = c.z;
}
void orc_spherical_uv(const float* d, float* o)
{
  vec2 r = GetSphericalUv(vec3(d[0], d[1], d[2]));
  o[0] = r.x; o[1] = r.y;
}
void orc_coordinate_system(const float* n, float* t, float* b)
{
  vec3 T, B;
  CreateCoordinateSystem(vec3(n[0], n[1], n[2]), T, B);
  t[0] = T.x; t[1] = T.y; t[2] = T.z; b[0] = B.x; b[1] = B.y; b[2] = B.z;
}
float orc_range_attenuation(float range, float dist) { return Tracer::getRangeAttenuation(range, dist); }
float orc_spot_attenuation(const float* p2l, const float* dir, float outerCos, float innerCos)
{
  return Tracer::getSpotAttenuation(vec3(p2l[0], p2l[1], p2l[2]), vec3(dir[0], dir[1], dir[2]), outerCos, innerCos);
}
static void orc_fill_state(State& st, const float* m, const float* N, const float* T, const float* B, float eta, int thin)
{
  st.depth = 0; st.eta = eta;
  st.position = vec3(0); st.normal = vec3(N[0], N[1], N[2]); st.ffnormal = st.normal;
  st.tangent = vec3(T[0], T[1], T[2]); st.bitangent = vec3(B[0], B[1], B[2]); st.texCoord = vec2(0, 0);
  Material& a = st.mat;
  a.albedo = vec3(m[0], m[1], m[2]); a.specular = m[3]; a.emission = vec3(0); a.anisotropy = m[4]; a.metallic = m[5]; a.roughness = m[6];
  a.subsurface = m[7]; a.specularTint = m[8]; a.sheen = m[9]; a.sheenTint = vec3(m[10], m[11], m[12]); a.clearcoat = m[13];
  a.clearcoatRoughness = m[14]; a.transmission = m[15]; a.ior = m[16]; a.attenuationColor = vec3(1); a.attenuationDistance = 1;
  a.ax = m[17]; a.ay = m[18]; a.f0 = vec3(m[19], m[20], m[21]); a.alpha = 1; a.unlit = false; a.thinwalled = thin != 0;
}
void orc_bsdf_eval(int pbrMode, const float* m, const float* N, const float* T, const float* B, float eta, int thin, const float* V, const float* L, float* f, float* pdf)
{
  State st;
  orc_fill_state(st, m, N, T, B, eta, thin);
  float p = 0.0f;
  vec3  v(V[0], V[1], V[2]), l(L[0], L[1], L[2]);
  vec3  r = pbrMode == 0 ? DisneyEval(st, v, st.ffnormal, l, p) : PbrEval(st, v, st.ffnormal, l, p);
  f[0] = r.x; f[1] = r.y; f[2] = r.z; *pdf = p;
}
void orc_bsdf_sample(int pbrMode, const float* m, const float* N, const float* T, const float* B, float eta, int thin, const float* V, uint32_t* seed, float* L, float* f,
                     float* pdf)
{
  State st;
  orc_fill_state(st, m, N, T, B, eta, thin);
  float p = 0.0f;
  vec3  v(V[0], V[1], V[2]), l(0);
  vec3  r = pbrMode == 0 ? DisneySample(st, v, st.ffnormal, l, p, *seed) : PbrSample(st, v, st.ffnormal, l, p, *seed);
  L[0] = l.x; L[1] = l.y; L[2] = l.z; f[0] = r.x; f[1] = r.y; f[2] = r.z; *pdf = p;
}

// Closest-hit known answers: n rays -> (t, node, prim, u, v); seeds are per-ray RNG states (in/out)
void orc_trace_closest(orc_ctx* c, uint32_t n, const float* org, const float* dir, uint32_t* seeds, float* out_t, int32_t* out_node,
                       int32_t* out_prim, float* out_uv)
{
  pt_RtxState st{};
#pragma omp parallel
  {
    Tracer tr(c->scene, st, c->variant);
#pragma omp for schedule(dynamic, 256)
    for(int64_t i = 0; i < (int64_t)n; ++i)
    {
      tr.prd.seed = seeds ? seeds[i] : 0u;
      Ray r{vec3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), vec3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2])};
      tr.ClosestHit(r);
      bool hit    = tr.prd.hitT != INFINITY_RT;
      out_t[i]    = tr.prd.hitT;
      out_node[i] = hit ? tr.prd.instanceID : -1;
      out_prim[i] = hit ? tr.prd.primitiveID : -1;
      out_uv[2 * i]     = hit ? tr.prd.baryCoord.x : 0.f;
      out_uv[2 * i + 1] = hit ? tr.prd.baryCoord.y : 0.f;
      if(seeds) seeds[i] = tr.prd.seed;
    }
  }
}

}  // extern "C"
