// TEST INFRASTRUCTURE -- part of the oracle/_ref recipe (see oracle/ref_glue/README.md). Not linked into the product.
//
// The reference's HOST-side sources on the path, compiled unmodified where they lie (-I /root/reference/src, -I /root/reference):
//   src/hdr_sampling.cpp            alias table of the environment importance sampling (:107-248), loadEnvironment (:56-99)
//   shaders/compress.glsl (C++ branch, :31-98 + the shared :111-139)   compress_unit_vec / packUnorm4x8 / roundEven as scene.cpp uses them
// against stubs/ (declarations of the Vulkan / nvpro_core / stb names those files mention; the allocator records what it is handed).
#include "stubs/vk_stub.h"

float* (*ref_stbi_loadf_hook)(const char* path, int* w, int* h) = nullptr;

#define private public  // buildAliasmap / createEnvironmentAccel are private members (src/hdr_sampling.hpp:61-62)
#include "hdr_sampling.cpp"
#undef private

// host_device.h (C++ branch) has brought vec3 / vec4 / uint into the global namespace; compress.glsl's host branch calls abs (-> <cmath>'s float
// overload) and normalize (-> argument-dependent lookup) unqualified
#include "shaders/compress.glsl"

extern "C" {

// HdrSampling::createEnvironmentAccel (src/hdr_sampling.cpp:187-248) on caller-provided RGBA32F pixels
int ref_env_accel(const float* rgba, int w, int h, EnvAccel* out, float* integral, float* average)
{
  HdrSampling hs;
  VkExtent2D  size{(uint32_t)w, (uint32_t)h};
  auto        acc = hs.createEnvironmentAccel(rgba, size);
  std::memcpy(out, acc.data(), acc.size() * sizeof(EnvAccel));
  *integral = hs.getIntegral();
  *average  = hs.getAverage();
  return (int)acc.size();
}

// HdrSampling::loadEnvironment (src/hdr_sampling.cpp:56-99) with the image decoder supplied by the caller (stb_image is not part of
// the reference tree): returns what the reference uploads -- the RGBA32F texels, the sampler it creates, the EnvAccel buffer.
int ref_load_environment(const char* path, float* (*loader)(const char*, int*, int*), float* rgba_out, int* w, int* h, EnvAccel* accel_out, float* integral,
                         float* average, int* sampler4)
{
  ref_stbi_loadf_hook = loader;
  HdrSampling             hs;
  nvvk::ResourceAllocator alloc;
  hs.setup(nullptr, nullptr, 0, &alloc);
  hs.loadEnvironment(path);
  *w = (int)hs.m_texHdr.extent.width;
  *h = (int)hs.m_texHdr.extent.height;
  if(rgba_out)
    std::memcpy(rgba_out, hs.m_texHdr.pixels.data(), hs.m_texHdr.pixels.size() * sizeof(float));
  if(accel_out)
    std::memcpy(accel_out, hs.m_accelImpSmpl.bytes.data(), hs.m_accelImpSmpl.bytes.size());
  *integral   = hs.getIntegral();
  *average    = hs.getAverage();
  sampler4[0] = hs.m_texHdr.sampler.magFilter; sampler4[1] = hs.m_texHdr.sampler.minFilter;
  sampler4[2] = hs.m_texHdr.sampler.addressModeU; sampler4[3] = hs.m_texHdr.sampler.addressModeV;
  return 0;
}

uint32_t ref_host_compress_unit_vec(const float* v) { return compress_unit_vec(vec3(v[0], v[1], v[2])); }
uint32_t ref_host_pack_unorm4x8(const float* v) { return packUnorm4x8(vec4(v[0], v[1], v[2], v[3])); }
float    ref_host_round_even(float x) { return roundEven(x); }
}
