// pt_renderer.hpp -- header-only C++ shim with the method set and call order of the reference's abstract
// Renderer (reference: src/renderer.h:30-48), implemented on the C ABI of libptmi.so (pt_api.h).
//
// Vulkan does not exist on the target, so the Vulkan handle parameters of the reference signature are dropped
// here; INTEGRATION.md shows the variant that keeps the reference's exact signatures inside its own tree.
// Methods are `void` like the reference's; failures are reported through lastError() / ok() (the reference
// itself asserts or ignores VkResults, src/rtx_pipeline.cpp:209,237).
#pragma once
#include <string>
#include "pt_api.h"

namespace ptmi {

struct Extent2D {
  uint32_t width, height;
};

class Renderer {  // src/renderer.h:30-48
public:
  virtual ~Renderer() = default;
  virtual void              setup(int deviceOrdinal)                                          = 0;  // (VkDevice, VkPhysicalDevice, familyIndex, allocator)
  virtual void              destroy()                                                         = 0;
  virtual void              run(const Extent2D& size)                                         = 0;  // (cmdBuf, size, profiler, descSets)
  virtual void              create(const Extent2D& size, const pt_SceneDesc* scene = nullptr) = 0;  // (size, layouts, Scene*)
  virtual const std::string name()                                                            = 0;
  void                      setPushContants(const pt_RtxState& state) { m_state = state; }          // (sic) src/renderer.h:44
  pt_RtxState               m_state{};
};

class HipPathTracer : public Renderer {
public:
  ~HipPathTracer() override { destroy(); }
  void setup(int deviceOrdinal) override
  {
    if(!m_ctx)
      check(pt_create(deviceOrdinal, &m_ctx));
  }
  void destroy() override
  {
    if(m_ctx)
      pt_destroy(m_ctx);
    m_ctx = nullptr;
  }
  void create(const Extent2D& size, const pt_SceneDesc* scene = nullptr) override
  {
    if(scene && check(pt_set_scene(m_ctx, scene)))
      check(pt_build_accel(m_ctx));
    check(pt_resize(m_ctx, int(size.width), int(size.height)));
  }
  void run(const Extent2D& size) override
  {
    m_state.size[0] = int(size.width);
    m_state.size[1] = int(size.height);
    check(pt_render_frame(m_ctx, &m_state));
  }
  const std::string name() override { return pt_renderer_name(); }

  // what the reference hands over through descriptor sets 2 and 3
  void setCamera(const pt_SceneCamera& c) { check(pt_set_camera(m_ctx, &c)); }
  // PT_VARIANT_RAYQUERY (the reference's RayQuery renderer, default) or PT_VARIANT_RTX (its RtxPipeline)
  void setVariant(int variant) { check(pt_set_variant(m_ctx, variant)); }
  void setSunAndSky(const pt_SunAndSky& s) { check(pt_set_sunsky(m_ctx, &s)); }
  void setEnvironment(const float* rgba32f, int w, int h, float* integral, float* average) { check(pt_set_env(m_ctx, rgba32f, w, h, integral, average)); }
  // HdrSampling::loadEnvironment (src/hdr_sampling.cpp:56-99) for a Radiance .hdr file: decode (the stbi_loadf of :64), upload, alias table
  bool loadEnvironment(const char* hdrPath, float* integral, float* average)
  {
    float* px = nullptr;
    int    w = 0, h = 0;
    char   err[256] = {0};
    if(pt_hdr_load(hdrPath, &px, &w, &h, err, sizeof(err)) != PT_OK)
    {
      m_error = err;
      return false;
    }
    const bool ok = check(pt_set_env(m_ctx, px, w, h, integral, average));
    pt_hdr_free(px);
    return ok;
  }
  // Scene::load for a .gltf / .glb file (src/scene.cpp:56-118): imports, uploads, builds the acceleration structure and sets the
  // file's first camera (or a fit to the bounding box) for the given aspect ratio.  Returns false and keeps lastError() on failure.
  bool loadGltf(const char* path, float aspect)
  {
    pt_GltfScene* sc = nullptr;
    char          msg[512];
    if(pt_gltf_load(path, &sc, msg, sizeof(msg)) != PT_OK)
    {
      m_status = PT_ERR_INVALID;
      m_error  = msg;
      return false;
    }
    const pt_SceneDesc* d  = pt_gltf_desc(sc);
    bool                ok = check(pt_set_scene(m_ctx, d)) && check(pt_build_accel(m_ctx));
    float               eye[3], center[3], up[3], fov;
    pt_SceneCamera      cam{};
    if(ok && pt_gltf_camera(sc, eye, center, up, &fov) == PT_OK && check(pt_camera_lookat(eye, center, up, fov, aspect, &cam)))
    {
      cam.nbLights = int(d->numLights);
      ok           = check(pt_set_camera(m_ctx, &cam));
    }
    pt_gltf_free(sc);  // pt_set_scene copied everything
    return ok;
  }
  // SampleExample::screenPicking (src/sample_example.cpp:468-511)
  bool pick(float x, float y, const pt_SceneCamera& cam, pt_PickResult* out) { return check(pt_pick(m_ctx, x, y, cam.viewInverse, cam.projInverse, out)); }
  void readAccum(float* rgba32f) { check(pt_read_accum(m_ctx, rgba32f)); }
  void writeAccum(const float* rgba32f) { check(pt_write_accum(m_ctx, rgba32f)); }  // checkpoint restore
  void useAnyHit(bool enable) { check(pt_use_any_hit(m_ctx, enable ? 1 : 0)); }  // RtxPipeline::useAnyHit
  void setAccelMode(int mode) { check(pt_set_accel_mode(m_ctx, mode)); }  // PT_ACCEL_TWO_LEVEL: AccelStructure's BLAS per prim-mesh + TLAS (src/accelstruct.cpp:110-162)
  void updateInstances(const pt_Node* nodes, uint32_t n) { check(pt_update_instances(m_ctx, nodes, n)); }  // new node.worldMatrix values: TLAS refit
  void tonemap(const pt_Tonemapper& tm, uint8_t* rgba8) { check(pt_tonemap(m_ctx, &tm, rgba8)); }
  // while SampleExample de-scales (m_descaling, src/sample_example.cpp:410-413): viewport of dispW x dispH from the reduced-size render
  void tonemapZoom(const pt_Tonemapper& tm, int dispW, int dispH, uint8_t* rgba8) { check(pt_tonemap_zoom(m_ctx, &tm, dispW, dispH, rgba8)); }
  // the display pass with frames in flight (main.cpp:213 prepareFrame / :261 submitFrame): begin after every frame, end returns the oldest image
  void tonemapBegin(const pt_Tonemapper& tm, int dispW, int dispH) { check(pt_tonemap_begin(m_ctx, &tm, dispW, dispH)); }
  void tonemapEnd(uint8_t* rgba8) { check(pt_tonemap_end(m_ctx, rgba8)); }
  int  tonemapPending() const { return pt_tonemap_pending(m_ctx); }

  bool               ok() const { return m_status == PT_OK; }
  int                status() const { return m_status; }
  const std::string& lastError() const { return m_error; }
  pt_context*        context() { return m_ctx; }

private:
  bool check(int rc)
  {
    m_status = rc;
    if(rc != PT_OK)
      m_error = pt_last_error(m_ctx);
    return rc == PT_OK;
  }
  pt_context* m_ctx    = nullptr;
  int         m_status = PT_OK;
  std::string m_error;
};

}  // namespace ptmi
