// TEST INFRASTRUCTURE -- part of the oracle/_ref recipe (see oracle/ref_glue/README.md). Not linked into the product.
//
// Compiles the reference's display pass -- shaders/post.frag with tonemapping.glsl and random.glsl, rewritten lexically by
// glsl2cpp.py -- and runs it like the full-screen triangle of RenderOutput::run (src/render_output.cpp:174-182) does: one fragment
// per pixel, uvCoords interpolated from shaders/passthrough.vert, the offscreen image bound with the sampler the reference creates
// for it (zeroed VkSamplerCreateInfo: NEAREST / NEAREST / REPEAT, maxLod = FLT_MAX; render_output.cpp:98-100) and the mip chain
// RenderOutput::genMipmap produces (render_output.cpp:188-193 -> nvvk::cmdGenerateMipmaps: level i is a VK_FILTER_LINEAR
// vkCmdBlitImage of level i-1, extent max(1, e/2); nvvk::mipLevels = floor(log2(max(w,h))) + 1).
#include <omp.h>
#include <cstring>
#include <vector>
#include "../../include/pt_types.h"
#include "ref_driver.h"

namespace glslc {
namespace refpost {
#include "post.frag"  // the generated file in the scratch directory (reference: shaders/post.frag)
static_assert(sizeof(Tonemapper) == sizeof(pt_Tonemapper), "host_device.h layout");
}  // namespace refpost

// vkCmdBlitImage with VK_FILTER_LINEAR from (sw x sh) to (dw x dh), whole image to whole image.  Vulkan 1.3 "Image Copies with Scaling":
// the dst texel centre (x + 0.5, y + 0.5) is scaled into src space; the src image is sampled with unnormalised coordinates, linear
// filter, clamp-to-edge.  Weights and the order of the lerps are the driver's; fixed here like the Appendix-F sampler of the oracle:
// horizontal lerp of the two rows first, then the vertical lerp, each as a*(1-t) + b*t.
static void blit_linear(const float* src, int sw, int sh, float* dst, int dw, int dh)
{
  const float su = float(sw) / float(dw), sv = float(sh) / float(dh);
  for(int y = 0; y < dh; ++y)
    for(int x = 0; x < dw; ++x)
    {
      float u = (float(x) + 0.5f) * su - 0.5f, v = (float(y) + 0.5f) * sv - 0.5f;
      float fu = ::floorf(u), fv = ::floorf(v);
      float a = u - fu, b = v - fv;
      int   x0 = (int)fu, y0 = (int)fv, x1 = x0 + 1, y1 = y0 + 1;
      x0 = x0 < 0 ? 0 : (x0 > sw - 1 ? sw - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > sw - 1 ? sw - 1 : x1);
      y0 = y0 < 0 ? 0 : (y0 > sh - 1 ? sh - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > sh - 1 ? sh - 1 : y1);
      for(int k = 0; k < 4; ++k)
      {
        float t00 = src[(size_t(y0) * sw + x0) * 4 + k], t10 = src[(size_t(y0) * sw + x1) * 4 + k];
        float t01 = src[(size_t(y1) * sw + x0) * 4 + k], t11 = src[(size_t(y1) * sw + x1) * 4 + k];
        float top = t00 * (1.0f - a) + t10 * a, bot = t01 * (1.0f - a) + t11 * a;
        dst[(size_t(y) * dw + x) * 4 + k] = top * (1.0f - b) + bot * b;
      }
    }
}
struct MipChain {
  std::vector<std::vector<float>> data;
  std::vector<RefMip>             mips;
};
static void build_chain(const float* rgba, int W, int H, MipChain& mc)
{
  int levels = 1;
  for(int m = W > H ? W : H; m > 1; m >>= 1)
    levels++;
  mc.data.assign(levels, {});
  mc.mips.assign(levels, RefMip{});
  mc.mips[0] = RefMip{W, H, rgba};
  int w = W, h = H;
  for(int i = 1; i < levels; ++i)
  {
    int nw = w > 1 ? w / 2 : 1, nh = h > 1 ? h / 2 : 1;
    mc.data[i].resize(size_t(nw) * nh * 4);
    blit_linear(mc.mips[i - 1].px, w, h, mc.data[i].data(), nw, nh);
    mc.mips[i] = RefMip{nw, nh, mc.data[i].data()};
    w = nw; h = nh;
  }
}
}  // namespace glslc

using namespace glslc;
using namespace glslc::refpost;

extern "C" {

// Number of mip levels and the texels of one level (for comparing the chain itself)
int ref_mip_chain(const float* rgba, int W, int H, int level, float* out, int* outW, int* outH)
{
  MipChain mc;
  build_chain(rgba, W, H, mc);
  if(level >= 0 && level < (int)mc.mips.size())
  {
    *outW = mc.mips[level].w; *outH = mc.mips[level].h;
    if(out)
      std::memcpy(out, mc.mips[level].px, size_t(*outW) * *outH * 16);
  }
  return (int)mc.mips.size();
}

// RenderOutput::genMipmap + RenderOutput::run on an accumulation image: out receives the fragment shader's fragColor (RGBA32F) per
// pixel of a W x H viewport (the reference's swapchain then stores it as UNORM8).
int ref_tonemap(const pt_Tonemapper* t, const float* rgba, int W, int H, float* out)
{
  MipChain mc;
  build_chain(rgba, W, H, mc);
  std::memcpy(&tm, t, sizeof(Tonemapper));
  inImage.kind = 2; inImage.w = W; inImage.h = H;
  inImage.mips = mc.mips.data(); inImage.numMips = (int)mc.mips.size();
#pragma omp parallel for schedule(static)
  for(int y = 0; y < H; ++y)
    for(int x = 0; x < W; ++x)
    {
      // passthrough.vert: v_texCoord = (0,0) (2,0) (0,2) at the corners of a triangle covering [-1,1]^2, interpolated at the pixel centre
      uvCoords     = vec2((float(x) + 0.5f) / float(W), (float(y) + 0.5f) / float(H));
      gl_FragCoord = vec4(float(x) + 0.5f, float(y) + 0.5f, 0.0f, 1.0f);
      shader_main();
      float* o = out + (size_t(y) * W + x) * 4;
      o[0] = fragColor.x; o[1] = fragColor.y; o[2] = fragColor.z; o[3] = fragColor.w;
    }
  return 0;
}
}
