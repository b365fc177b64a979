// TEST INFRASTRUCTURE -- part of the oracle/_ref recipe.  The handful of Vulkan / nvpro_core / glm / stb declarations that
// /root/reference/src/hdr_sampling.{hpp,cpp} and the host branch of shaders/{host_device.h,compress.glsl} mention, so that those
// files compile UNMODIFIED where they lie.  Nothing here computes anything: the allocator records what it is handed, so a test can
// read back what the reference uploaded.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#define GLSLC_KEEP_MATH_DEFINES 1  // hdr_sampling.cpp uses <cmath>'s M_PI (the double constant)
#include "../glsl_compat.h"

namespace glm {
using vec2  = glslc::vec2;
using vec3  = glslc::vec3;
using vec4  = glslc::vec4;
using ivec2 = glslc::ivec2;
using mat4  = glslc::mat4;
}  // namespace glm

typedef struct VkDevice_T*         VkDevice;
typedef struct VkPhysicalDevice_T* VkPhysicalDevice;
typedef struct VkQueue_T*          VkQueue;
typedef struct VkImage_T*          VkImage;
typedef struct VkBuffer_T*         VkBuffer;
typedef struct VkCommandBuffer_T*  VkCommandBuffer;
typedef uint64_t                   VkDeviceSize;
typedef uint32_t                   VkFlags;
#define VK_NULL_HANDLE nullptr
struct VkExtent2D {
  uint32_t width, height;
};
enum VkStructureType { VK_STRUCTURE_TYPE_SAMPLER_CREATE_INFO = 31 };
enum VkFilter { VK_FILTER_NEAREST = 0, VK_FILTER_LINEAR = 1 };
enum VkSamplerMipmapMode { VK_SAMPLER_MIPMAP_MODE_NEAREST = 0, VK_SAMPLER_MIPMAP_MODE_LINEAR = 1 };
enum VkSamplerAddressMode { VK_SAMPLER_ADDRESS_MODE_REPEAT = 0, VK_SAMPLER_ADDRESS_MODE_MIRRORED_REPEAT = 1, VK_SAMPLER_ADDRESS_MODE_CLAMP_TO_EDGE = 2 };
enum VkFormat { VK_FORMAT_UNDEFINED = 0, VK_FORMAT_R32G32B32A32_SFLOAT = 109 };
enum { VK_BUFFER_USAGE_STORAGE_BUFFER_BIT = 0x20 };
struct VkSamplerCreateInfo {
  VkStructureType      sType;
  const void*          pNext;
  VkFlags              flags;
  VkFilter             magFilter, minFilter;
  VkSamplerMipmapMode  mipmapMode;
  VkSamplerAddressMode addressModeU, addressModeV, addressModeW;
  float                mipLodBias;
  uint32_t             anisotropyEnable;
  float                maxAnisotropy;
  uint32_t             compareEnable;
  int                  compareOp;
  float                minLod, maxLod;
  int                  borderColor;
  uint32_t             unnormalizedCoordinates;
};
struct VkImageCreateInfo {
  VkExtent2D extent;
  VkFormat   format;
};
struct VkImageViewCreateInfo {
  VkImage image;
};
inline void vkGetDeviceQueue(VkDevice, uint32_t, uint32_t, VkQueue* q) { *q = nullptr; }

namespace nvvk {
struct Image {
  VkImage            image = nullptr;
  std::vector<float> pixels;  // what createImage was handed
  VkExtent2D         extent{0, 0};
};
struct Texture {
  VkImage             image = nullptr;
  std::vector<float>  pixels;
  VkExtent2D          extent{0, 0};
  VkSamplerCreateInfo sampler{};
};
struct Buffer {
  VkBuffer             buffer = nullptr;
  std::vector<uint8_t> bytes;  // what createBuffer was handed
};
inline VkImageCreateInfo     makeImage2DCreateInfo(const VkExtent2D& e, VkFormat f) { return VkImageCreateInfo{e, f}; }
inline VkImageViewCreateInfo makeImageViewCreateInfo(VkImage i, const VkImageCreateInfo&) { return VkImageViewCreateInfo{i}; }
struct ScopeCommandBuffer {
  ScopeCommandBuffer(VkDevice, uint32_t, VkQueue) {}
  operator VkCommandBuffer() const { return nullptr; }
};
struct DebugUtil {
  void setup(VkDevice) {}
};
class ResourceAllocator
{
public:
  void  destroy(Texture& t) { t = Texture(); }
  void  destroy(Buffer& b) { b = Buffer(); }
  Image createImage(VkCommandBuffer, VkDeviceSize size, const void* data, const VkImageCreateInfo& info)
  {
    Image im;
    im.image  = reinterpret_cast<VkImage>(this);
    im.extent = info.extent;
    im.pixels.assign((const float*)data, (const float*)data + size / sizeof(float));
    return im;
  }
  Texture createTexture(const Image& im, const VkImageViewCreateInfo&, const VkSamplerCreateInfo& s)
  {
    Texture t;
    t.image = im.image; t.pixels = im.pixels; t.extent = im.extent; t.sampler = s;
    return t;
  }
  template <class T> Buffer createBuffer(VkCommandBuffer, const std::vector<T>& v, VkFlags)
  {
    Buffer b;
    b.buffer = reinterpret_cast<VkBuffer>(this);
    b.bytes.assign((const uint8_t*)v.data(), (const uint8_t*)(v.data() + v.size()));
    return b;
  }
  void finalizeAndReleaseStaging() {}
};
}  // namespace nvvk
#define NAME_VK(x) (void)0

// stb_image: the decoder itself is third-party code outside /root/reference; the test supplies the loader (e.g. the product's
// Radiance .hdr reader) through this pointer
#define STBI_rgb_alpha 4
extern "C" float* (*ref_stbi_loadf_hook)(const char* path, int* w, int* h);
inline float* stbi_loadf(const char* path, int* w, int* h, int* comp, int)
{
  *comp = 3;
  return ref_stbi_loadf_hook ? ref_stbi_loadf_hook(path, w, h) : nullptr;
}
inline void stbi_image_free(void* p) { std::free(p); }
