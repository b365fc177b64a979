// TEST INFRASTRUCTURE -- part of the oracle/_ref recipe (see README.md). Not linked into the product.
//
// The any-hit stage of the reference's ray-tracing pipeline, shaders/pathtrace.rahit, compiled from the lexically rewritten source.
// It declares `layout(location = 0) rayPayloadInEXT PtPayload prd` and draws its stochastic alpha test from prd.seed -- on WHATEVER payload
// the trace call named: the shadow ray passes location 1 (ShadowHitPayload, whose first member is the seed copy), so the bytes of that
// payload are laid over the start of `prd` here, exactly what the aliasing does on a GPU (traceray_rtx.glsl:54-55, pathtrace.rahit:44,112).
#include <cstring>
#include <vector>
#include "../../include/pt_types.h"
#include "ref_driver.h"

namespace glslc {
namespace rahit {
#include "pathtrace.rahit"
static std::vector<InstanceData> s_geoInfo;
static std::vector<sampler2D>    s_textures;
}  // namespace rahit

// true: the intersection is accepted (the shader returned without ignoreIntersectionEXT)
bool ref_rtx_rahit(void* payload, size_t bytes, vec2 attribs)
{
  const size_t n = bytes < sizeof(rahit::prd) ? bytes : sizeof(rahit::prd);
  std::memcpy(&rahit::prd, payload, n);
  rahit::bary           = attribs;
  gl_IgnoreIntersection = false;
  rahit::shader_main();
  std::memcpy(payload, &rahit::prd, n);
  return !gl_IgnoreIntersection;
}
void ref_rtx_rahit_bind(const pt_SceneDesc* d)
{
  using namespace rahit;
  s_geoInfo.resize(d->numPrimMeshes);
  for(uint32_t i = 0; i < d->numPrimMeshes; ++i)
  {
    s_geoInfo[i].vertexAddress = (uint64_t)(uintptr_t)(d->vertices + d->primMeshes[i].vertexOffset);
    s_geoInfo[i].indexAddress  = (uint64_t)(uintptr_t)(d->indices + d->primMeshes[i].firstIndex);
    s_geoInfo[i].materialIndex = d->primMeshes[i].materialIndex;
  }
  geoInfo   = s_geoInfo.data();
  materials = reinterpret_cast<const GltfShadeMaterial*>(d->materials);
  s_textures.resize(d->numTextures);
  for(uint32_t i = 0; i < d->numTextures; ++i)
  {
    s_textures[i].kind = 0;
    s_textures[i].id   = (int)i;
    s_textures[i].w    = d->textures[i].width;
    s_textures[i].h    = d->textures[i].height;
  }
  texturesMap = s_textures.data();
}
}  // namespace glslc
