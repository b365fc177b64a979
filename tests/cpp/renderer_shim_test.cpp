// Host-only check that the C++ shim compiles against pt_api.h and links with libptmi.so.
// With a GPU: renders one frame of a two-triangle scene.  Without: setup() must fail loudly (no fallback).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "pt_renderer.hpp"

int main(int argc, char** argv)
{
  ptmi::HipPathTracer r;
  if(r.name() != "HIP")
    return 2;
  r.setup(0);
  if(!r.ok())
  {
    std::printf("NO_DEVICE status=%d msg=%s\n", r.status(), r.lastError().c_str());
    return r.status() == PT_ERR_NO_DEVICE ? 0 : 3;
  }
  const float pos[12] = {-1, -1, 0, 1, -1, 0, 1, 1, 0, -1, 1, 0}, nrm[12] = {0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};
  const float tan[16] = {1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1}, uv[8] = {0, 0, 1, 0, 1, 1, 0, 1};
  float       col[16];
  for(float& c : col) c = 1.f;
  pt_VertexAttributes v[4];
  pt_pack_vertices(4, pos, nrm, tan, uv, col, v);
  uint32_t             idx[6] = {0, 1, 2, 0, 2, 3};
  pt_PrimMesh          pm{0, 4, 0, 6, 0};
  pt_Node              nd{};
  for(int i = 0; i < 4; ++i) nd.worldMatrix[i * 5] = 1.f;
  pt_GltfShadeMaterial m{};
  m.pbrBaseColorFactor[0] = m.pbrBaseColorFactor[1] = m.pbrBaseColorFactor[2] = 0.8f;
  m.pbrBaseColorFactor[3] = 1.f;
  m.pbrBaseColorTexture = m.pbrMetallicRoughnessTexture = m.emissiveTexture = m.normalTexture = m.transmissionTexture = m.clearcoatTexture = m.clearcoatRoughnessTexture = m.thicknessTexture = -1;
  m.pbrRoughnessFactor = 1.f; m.ior = 1.5f; m.doubleSided = 1; m.attenuationDistance = 3.4e38f;
  m.attenuationColor[0] = m.attenuationColor[1] = m.attenuationColor[2] = 1.f;
  for(int i = 0; i < 4; ++i) m.uvTransform[i * 5] = 1.f;
  pt_SceneDesc sd{v, 4, idx, 6, &pm, 1, &nd, 1, &m, 1, nullptr, 0, nullptr, 0};
  r.create({64, 64}, &sd);
  std::vector<float> env(16 * 8 * 4, 1.f);
  float              integral = 0, average = 0;
  r.setEnvironment(env.data(), 16, 8, &integral, &average);
  pt_SceneCamera cam;
  const float    eye[3] = {0, 0, 3}, ctr[3] = {0, 0, 0}, up[3] = {0, 1, 0};
  pt_camera_lookat(eye, ctr, up, 45.f, 1.f, &cam);
  r.setCamera(cam);
  pt_RtxState st{0, 10, 1, 4.f * integral, 1.f, 0, 0, 0, {64, 64}, 0, 65000};
  r.setPushContants(st);
  r.run({64, 64});
  std::vector<float> img(64 * 64 * 4);
  r.readAccum(img.data());
  if(!r.ok())
  {
    std::printf("ERROR %s\n", r.lastError().c_str());
    return 4;
  }
  const bool first = std::fabs(img[0] - 1.f) < 1e-5f && img[(32 * 64 + 32) * 4] > 0.1f;
  {  // AccelStructure's own shape (BLAS per prim-mesh + TLAS): same pixels; then the node moves out of view and back (TLAS refit)
    const std::vector<float> flat = img;
    r.setAccelMode(PT_ACCEL_TWO_LEVEL);
    r.setPushContants(st);
    r.run({64, 64});
    r.readAccum(img.data());
    if(!r.ok() || std::memcmp(flat.data(), img.data(), flat.size() * sizeof(float)) != 0)
    {
      std::printf("ERROR two-level image differs (%s)\n", r.lastError().c_str());
      return 8;
    }
    pt_Node moved = nd;
    moved.worldMatrix[12] = 50.f;
    r.updateInstances(&moved, 1);
    r.setPushContants(st);
    r.run({64, 64});
    r.readAccum(img.data());
    if(!r.ok() || std::fabs(img[(32 * 64 + 32) * 4] - 1.f) > 1e-5f)  // the constant environment where the quad was
    {
      std::printf("ERROR updateInstances: centre=%.3f (%s)\n", img[(32 * 64 + 32) * 4], r.lastError().c_str());
      return 9;
    }
    r.updateInstances(&nd, 1);
    r.setPushContants(st);
    r.run({64, 64});
    r.readAccum(img.data());
    if(!r.ok() || std::memcmp(flat.data(), img.data(), flat.size() * sizeof(float)) != 0)
    {
      std::printf("ERROR image after moving back differs (%s)\n", r.lastError().c_str());
      return 10;
    }
    r.setAccelMode(PT_ACCEL_FLAT);
  }
  {  // the display pass with frames in flight (tonemapBegin / tonemapEnd) hands out the images of the synchronous pass
    pt_Tonemapper        tm{1.f, 1.f, 1.f, 0.f, 1.f, 1.f, {1.f, 1.f}, 0, 0.5f, 0.5f, 1};
    std::vector<uint8_t> a(64 * 64 * 4), b(64 * 64 * 4), sync(64 * 64 * 4);
    r.setPushContants(st);
    r.run({64, 64});
    r.tonemapBegin(tm, 64, 64);
    r.setPushContants(st);
    r.run({64, 64});  // frame 0 again: the same image
    r.tonemapBegin(tm, 64, 64);
    const int pending = r.tonemapPending();
    r.tonemapEnd(a.data());
    r.tonemapEnd(b.data());
    r.tonemap(tm, sync.data());
    if(!r.ok() || pending != 2 || r.tonemapPending() != 0 || a != b || b != sync)
    {
      std::printf("ERROR pipelined display pass (pending %d, %s)\n", pending, r.lastError().c_str());
      return 11;
    }
  }
  if(argc > 1)
  {  // Scene::load path: the same quad as a .glb written by vk_raytrace_amd.gltf.save_gltf must render the same image
    std::vector<float> ref = img;
    if(!r.loadGltf(argv[1], 1.f))
    {
      std::printf("ERROR loadGltf: %s\n", r.lastError().c_str());
      return 6;
    }
    r.create({64, 64});
    r.setPushContants(st);
    r.run({64, 64});
    r.readAccum(img.data());
    float worst = 0.f;
    for(size_t i = 0; i < ref.size(); ++i)
      worst = std::fmax(worst, std::fabs(ref[i] - img[i]));
    if(!r.ok() || worst > 1e-5f)
    {
      std::printf("ERROR glTF render differs (%s)\n", r.lastError().c_str());
      return 7;
    }
  }
  std::printf("OK corner=%.3f centre=%.3f\n", img[0], img[(32 * 64 + 32) * 4]);
  return first ? 0 : 5;
}
