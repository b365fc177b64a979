// TEST INFRASTRUCTURE -- part of the CPU oracle (see oracle/README.md). Not linked into the product.
//
// Scene storage, texture sampling and the ray/scene queries of the CPU oracle.
//
// The reference delegates BVH build + traversal + the ray/triangle test to the Vulkan driver
// (src/accelstruct.cpp:125-126,161; shaders/traceray_rq.glsl:110-134), so there is no reference
// code to restate for them.  What IS specified is restated here as a
// BVH-independent "trace contract" that the HIP kernels implement identically:
//
//  T1 world triangles: node n (in order), triangle k of its prim-mesh -> world index w = base[n]+k,
//     vertices p_i = ((M.c0*x + M.c1*y) + M.c2*z) + M.c3 in fp32, e1 = p1-p0, e2 = p2-p0.
//  T2 ray/triangle: Moeller-Trumbore on (p0,e1,e2):  pv = cross(d,e2); det = dot(e1,pv);
//     det == 0 -> miss; inv = 1/det; tv = o-p0; u = dot(tv,pv)*inv; u<0||u>1 -> miss;
//     qv = cross(tv,e1); v = dot(d,qv)*inv; v<0||u+v>1 -> miss; t = dot(e2,qv)*inv.
//     bary = (u,v) weights vertex 1 and 2 (E8).
//  T3 facing (E3): front-facing <=> det > 0 for an instance with det(M3x3) > 0, flipped when
//     det(M3x3) < 0 (the facing test is an object-space property).  Back faces are culled unless the
//     material is doubleSided (E4).
//  T4 candidates are ordered by the key (t, w).  A query returns the smallest key strictly greater
//     than a given previous key with 0 < t < tmax  (E1, E2).
//  T5 ClosestHit: walk candidates in key order; an opaque candidate (E5) commits; a non-opaque one
//     runs HitTest (one rand(prd.seed) draw, traceray_rq.glsl:98) and commits iff it passes (E6).
//     This is the behaviour of a driver whose traversal is perfectly front-to-back.
//  T6 AnyHit (shadow): the same walk bounded by maxDist: candidates in key order, an opaque one commits without a draw, a non-opaque
//     one runs HitTest; the first commit ends the ray (E7).  Vulkan leaves candidate order to the implementation; front-to-back is the one
//     order that T5 already uses, and it lets a shadow ray stop at the nearest certain hit instead of searching the whole ray for an
//     opaque triangle behind it (round 2: the "any opaque triangle first" order of round 1 made shadow rays 3.6x as expensive as
//     closest-hit rays on the GPU).
//
// "Parity unpinned" for T1-T6 themselves: the reference holds no code or test vectors for traversal; everything the
// reference DOES specify (the shader code driving these queries) is pinned through oracle/_ref.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../include/pt_types.h"
#include "glsl_math.h"
#include "orc_host.h"

namespace orc {

struct Texture {
  std::vector<uint8_t> px;
  int                  w = 0, h = 0, mag = 1, wrapS = 0, wrapT = 0;
};

struct WorldTri {
  vec3     p0, e1, e2;
  uint32_t node;   // TLAS instance index (prd.instanceID)
  uint32_t prim;   // triangle index inside the prim-mesh (prd.primitiveID)
  uint32_t flags;  // bit0 opaque, bit1 cull disabled, bit2 winding flipped by the instance transform
};
enum { TRI_OPAQUE = 1, TRI_NOCULL = 2, TRI_FLIP = 4 };

struct BvhNode {
  float    bmin[3], bmax[3];
  uint32_t left, right;  // children (inner) ; for leaves: left = first tri slot, right = count | 0x80000000
};

struct Stats {
  uint64_t samples = 0, closestRays = 0, shadowRays = 0, shadedHits = 0, misses = 0, alphaTests = 0, neeLookups = 0;
  uint64_t nodesVisited = 0, trisTested = 0, texTaps = 0;
  uint64_t nodesShadow = 0, trisShadow = 0;  // the part of nodesVisited / trisTested spent on shadow rays
  void     add(const Stats& o)
  {
    samples += o.samples; closestRays += o.closestRays; shadowRays += o.shadowRays; shadedHits += o.shadedHits;
    misses += o.misses; alphaTests += o.alphaTests; neeLookups += o.neeLookups; nodesVisited += o.nodesVisited;
    trisTested += o.trisTested; texTaps += o.texTaps; nodesShadow += o.nodesShadow; trisShadow += o.trisShadow;
  }
};

struct Candidate {
  bool     found = false;
  float    t = 0, u = 0, v = 0;
  uint32_t w = 0;
};

struct Scene {
  std::vector<pt_VertexAttributes>  vertices;
  std::vector<uint32_t>             indices;
  std::vector<pt_PrimMesh>          primMeshes;
  std::vector<pt_Node>              nodes;
  std::vector<pt_GltfShadeMaterial> materials;
  std::vector<pt_Light>             lights;
  std::vector<Texture>              textures;
  // derived
  std::vector<mat4x3>   objectToWorld, worldToObject;
  std::vector<WorldTri> tris;
  std::vector<uint32_t> triOrder;  // BVH leaf order -> world index
  std::vector<BvhNode>  bvh;
  bool                  useBvh = true;
  bool                  anyHit = true;  // RtxPipeline::useAnyHit (src/rtx_pipeline.cpp:269-276): false = hit groups without an any-hit shader, every triangle opaque
  // environment
  std::vector<float>       env;  // RGBA32F
  int                      envW = 0, envH = 0;
  std::vector<pt_EnvAccel> envAccel;
  float                    envIntegral = 1.f, envAverage = 1.f;
  pt_SceneCamera           camera{};
  pt_SunAndSky             sunsky{};

  // ------------------------------------------------------------------------------------------
  bool set(const pt_SceneDesc* d)
  {
    vertices.assign(d->vertices, d->vertices + d->numVertices);
    indices.assign(d->indices, d->indices + d->numIndices);
    primMeshes.assign(d->primMeshes, d->primMeshes + d->numPrimMeshes);
    nodes.assign(d->nodes, d->nodes + d->numNodes);
    materials.assign(d->materials, d->materials + d->numMaterials);
    lights.clear();
    if(d->numLights)
      lights.assign(d->lights, d->lights + d->numLights);
    textures.clear();
    for(uint32_t i = 0; i < d->numTextures; ++i)
    {
      Texture t;
      t.w = d->textures[i].width;
      t.h = d->textures[i].height;
      t.px.assign(d->textures[i].rgba8, d->textures[i].rgba8 + size_t(t.w) * t.h * 4);
      t.mag   = d->textures[i].magFilter;
      t.wrapS = d->textures[i].wrapS;
      t.wrapT = d->textures[i].wrapT;
      textures.push_back(std::move(t));
    }
    return build_world();
  }

  // T1 + instance flags (src/accelstruct.cpp:144-149)
  bool build_world()
  {
    tris.clear();
    objectToWorld.resize(nodes.size());
    worldToObject.resize(nodes.size());
    for(size_t n = 0; n < nodes.size(); ++n)
    {
      const pt_Node& nd = nodes[n];
      if(nd.primMesh < 0 || (size_t)nd.primMesh >= primMeshes.size())
        return false;
      const pt_PrimMesh& pm = primMeshes[nd.primMesh];
      if(pm.materialIndex >= (int)materials.size())
        return false;
      const pt_GltfShadeMaterial& mat = materials[std::max(0, pm.materialIndex)];
      const float*                m   = nd.worldMatrix;
      mat4x3&                     o2w = objectToWorld[n];
      for(int c = 0; c < 4; ++c)
        o2w.c[c] = vec3(m[c * 4 + 0], m[c * 4 + 1], m[c * 4 + 2]);
      double md[16], inv[16];
      for(int i = 0; i < 16; ++i)
        md[i] = m[i];
      md[3] = md[7] = md[11] = 0.0;
      md[15]                 = 1.0;
      if(!invert4x4(md, inv))
        return false;
      for(int c = 0; c < 4; ++c)
        worldToObject[n].c[c] = vec3((float)inv[c * 4 + 0], (float)inv[c * 4 + 1], (float)inv[c * 4 + 2]);
      double det3 = md[0] * (md[5] * md[10] - md[9] * md[6]) - md[4] * (md[1] * md[10] - md[9] * md[2]) + md[8] * (md[1] * md[6] - md[5] * md[2]);

      uint32_t flags = 0;
      if(!anyHit || mat.alphaMode == 0 || (mat.pbrBaseColorFactor[3] == 1.0f && mat.pbrBaseColorTexture == -1))
        flags |= TRI_OPAQUE;
      if(mat.doubleSided == 1)
        flags |= TRI_NOCULL;
      if(det3 < 0.0)
        flags |= TRI_FLIP;

      for(uint32_t k = 0; k < pm.indexCount / 3; ++k)
      {
        vec3 p[3];
        for(int j = 0; j < 3; ++j)
        {
          uint32_t vi = indices[pm.firstIndex + 3 * k + j];
          if(vi >= pm.vertexCount)
            return false;
          const float* q = vertices[pm.vertexOffset + vi].position;
          p[j]           = mul_point(o2w, vec3(q[0], q[1], q[2]));
        }
        WorldTri t;
        t.p0    = p[0];
        t.e1    = p[1] - p[0];
        t.e2    = p[2] - p[0];
        t.node  = (uint32_t)n;
        t.prim  = k;
        t.flags = flags;
        tris.push_back(t);
      }
    }
    build_bvh();
    return true;
  }

  // ---- T2/T3 ---------------------------------------------------------------------------------
  static inline bool intersect(const WorldTri& tr, vec3 o, vec3 d, float& t, float& u, float& v)
  {
    vec3  pv  = cross(d, tr.e2);
    float det = dot(tr.e1, pv);
    if(det == 0.0f)
      return false;
    if(!(tr.flags & TRI_NOCULL))
    {
      bool front = (tr.flags & TRI_FLIP) ? (det < 0.0f) : (det > 0.0f);
      if(!front)
        return false;
    }
    float inv = 1.0f / det;
    vec3  tv  = o - tr.p0;
    u         = dot(tv, pv) * inv;
    if(u < 0.0f || u > 1.0f)
      return false;
    vec3 qv = cross(tv, tr.e1);
    v       = dot(d, qv) * inv;
    if(v < 0.0f || u + v > 1.0f)
      return false;
    t = dot(tr.e2, qv) * inv;
    return true;
  }

  // ---- a binned-SAH BVH2 (leaves <= 4 triangles).  It only makes the oracle fast -- results do not depend on it --
  // and its node / triangle visit counts are the "binary-node equivalents" of the algorithmic-bytes model
  void build_bvh()
  {
    bvh.clear();
    triOrder.resize(tris.size());
    for(size_t i = 0; i < tris.size(); ++i)
      triOrder[i] = (uint32_t)i;
    if(tris.empty())
      return;
    std::vector<float> cen(tris.size() * 3), lo(tris.size() * 3), hi(tris.size() * 3);
    for(size_t i = 0; i < tris.size(); ++i)
    {
      vec3 a = tris[i].p0, b = tris[i].p0 + tris[i].e1, c = tris[i].p0 + tris[i].e2;
      for(int k = 0; k < 3; ++k)
      {
        lo[i * 3 + k]  = std::min(a[k], std::min(b[k], c[k]));
        hi[i * 3 + k]  = std::max(a[k], std::max(b[k], c[k]));
        cen[i * 3 + k] = 0.5f * (lo[i * 3 + k] + hi[i * 3 + k]);
      }
    }
    bvh.reserve(tris.size() * 2);
    bvh.push_back(BvhNode{});
    struct Job { uint32_t node, first, count; };
    std::vector<Job> stack{{0u, 0u, (uint32_t)tris.size()}};
    while(!stack.empty())
    {
      Job j = stack.back();
      stack.pop_back();
      float bmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, bmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for(uint32_t i = j.first; i < j.first + j.count; ++i)
      {
        uint32_t w = triOrder[i];
        for(int k = 0; k < 3; ++k)
        {
          bmn[k] = std::min(bmn[k], lo[w * 3 + k]);
          bmx[k] = std::max(bmx[k], hi[w * 3 + k]);
          cmn[k] = std::min(cmn[k], cen[w * 3 + k]);
          cmx[k] = std::max(cmx[k], cen[w * 3 + k]);
        }
      }
      // pad by a relative epsilon: the box test must never reject a triangle the triangle test accepts
      for(int k = 0; k < 3; ++k)
      {
        float pad = 1e-5f * std::max(std::fabs(bmn[k]), std::fabs(bmx[k])) + 1e-30f;
        bvh[j.node].bmin[k] = bmn[k] - pad;
        bvh[j.node].bmax[k] = bmx[k] + pad;
      }
      if(j.count <= 4)
      {
        bvh[j.node].left  = j.first;
        bvh[j.node].right = j.count | 0x80000000u;
        continue;
      }
      // binned SAH (16 bins per axis); falls back to the median of the widest axis when no split helps
      const int NB = 16;
      int       bestAx = -1, bestBin = -1;
      float     bestCost = FLT_MAX;
      auto      half_area = [](const float* mn, const float* mx) {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
      };
      for(int ax = 0; ax < 3; ++ax)
      {
        float ext = cmx[ax] - cmn[ax];
        if(!(ext > 0.0f)) continue;
        uint32_t cnt[NB] = {0};
        float    bmnb[NB][3], bmxb[NB][3];
        for(int b = 0; b < NB; ++b)
          for(int k = 0; k < 3; ++k) { bmnb[b][k] = FLT_MAX; bmxb[b][k] = -FLT_MAX; }
        for(uint32_t i = j.first; i < j.first + j.count; ++i)
        {
          uint32_t w = triOrder[i];
          int      b = std::min(NB - 1, int((cen[w * 3 + ax] - cmn[ax]) / ext * NB));
          cnt[b]++;
          for(int k = 0; k < 3; ++k) { bmnb[b][k] = std::min(bmnb[b][k], lo[w * 3 + k]); bmxb[b][k] = std::max(bmxb[b][k], hi[w * 3 + k]); }
        }
        float    la[NB], ra[NB];
        uint32_t lc[NB], rc[NB];
        float    mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        uint32_t c = 0;
        for(int b = 0; b < NB; ++b)
        {
          c += cnt[b];
          for(int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], bmnb[b][k]); mx[k] = std::max(mx[k], bmxb[b][k]); }
          lc[b] = c; la[b] = c ? half_area(mn, mx) : 0.f;
        }
        for(int k = 0; k < 3; ++k) { mn[k] = FLT_MAX; mx[k] = -FLT_MAX; }
        c = 0;
        for(int b = NB - 1; b >= 0; --b)
        {
          c += cnt[b];
          for(int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], bmnb[b][k]); mx[k] = std::max(mx[k], bmxb[b][k]); }
          rc[b] = c; ra[b] = c ? half_area(mn, mx) : 0.f;
        }
        for(int b = 0; b < NB - 1; ++b)
        {
          if(lc[b] == 0 || rc[b + 1] == 0) continue;
          float cost = la[b] * lc[b] + ra[b + 1] * rc[b + 1];
          if(cost < bestCost) { bestCost = cost; bestAx = ax; bestBin = b; }
        }
      }
      uint32_t mid;
      if(bestAx >= 0)
      {
        float ext = cmx[bestAx] - cmn[bestAx];
        auto  it  = std::partition(triOrder.begin() + j.first, triOrder.begin() + j.first + j.count, [&](uint32_t w) {
          return std::min(NB - 1, int((cen[w * 3 + bestAx] - cmn[bestAx]) / ext * NB)) <= bestBin;
        });
        mid = uint32_t(it - triOrder.begin());
      }
      else
      {
        int ax = 0;
        if(cmx[1] - cmn[1] > cmx[ax] - cmn[ax]) ax = 1;
        if(cmx[2] - cmn[2] > cmx[ax] - cmn[ax]) ax = 2;
        mid = j.first + j.count / 2;
        std::nth_element(triOrder.begin() + j.first, triOrder.begin() + mid, triOrder.begin() + j.first + j.count,
                         [&](uint32_t a, uint32_t b) { return cen[a * 3 + ax] < cen[b * 3 + ax] || (cen[a * 3 + ax] == cen[b * 3 + ax] && a < b); });
      }
      uint32_t l = (uint32_t)bvh.size();
      bvh.push_back(BvhNode{});
      bvh.push_back(BvhNode{});
      bvh[j.node].left  = l;
      bvh[j.node].right = l + 1;
      stack.push_back({l, j.first, mid - j.first});
      stack.push_back({l + 1, mid, j.first + j.count - mid});
    }
  }

  static inline bool key_less(float ta, uint32_t wa, float tb, uint32_t wb) { return ta < tb || (ta == tb && wa < wb); }

  // T4: smallest key (t,w) > (tPrev,wPrev) with t < tmax among triangles selected by `mask`:
  //   want = 0: every triangle, 1: non-opaque only, 2: opaque only (first found, early out -- any-hit)
  Candidate query(vec3 o, vec3 d, float tmax, float tPrev, uint32_t wPrev, int want, Stats* st) const
  {
    Candidate best;
    float     bestT = tmax;  // exclusive bound until something is found
    auto      test  = [&](uint32_t w) -> bool {
      const WorldTri& tr = tris[w];
      if(want == 1 && (tr.flags & TRI_OPAQUE)) return false;
      if(want == 2 && !(tr.flags & TRI_OPAQUE)) return false;
      if(st) st->trisTested++;
      float t, u, v;
      if(!intersect(tr, o, d, t, u, v)) return false;
      if(!(t < tmax)) return false;
      if(!key_less(tPrev, wPrev, t, w)) return false;  // also enforces t > 0 for the initial key (0, ~0u)
      if(best.found && !key_less(t, w, best.t, best.w)) return false;
      best.found = true; best.t = t; best.u = u; best.v = v; best.w = w;
      bestT = t;
      return true;
    };
    if(!useBvh || bvh.empty())
    {
      for(uint32_t w = 0; w < tris.size(); ++w)
        if(test(w) && want == 2)
          return best;
      return best;
    }
    vec3     inv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128];
    int      sp   = 0;
    stack[sp++]   = 0;
    while(sp)
    {
      const BvhNode& nd = bvh[stack[--sp]];
      if(st) st->nodesVisited++;
      float tn = 0.0f, tf = best.found ? bestT : tmax;
      bool  hit = true;
      for(int k = 0; k < 3; ++k)
      {
        float t0 = (nd.bmin[k] - o[k]) * inv[k];
        float t1 = (nd.bmax[k] - o[k]) * inv[k];
        if(std::isnan(t0) || std::isnan(t1)) continue;  // origin on a slab plane of a flat axis
        if(t0 > t1) std::swap(t0, t1);
        t1 *= 1.0000004f;
        t0 *= 0.9999996f;
        if(t0 > tn) tn = t0;
        if(t1 < tf) tf = t1;
        if(tn > tf) { hit = false; break; }
      }
      if(!hit) continue;
      if(nd.right & 0x80000000u)
      {
        uint32_t cnt = nd.right & 0x7fffffffu;
        for(uint32_t i = 0; i < cnt; ++i)
          if(test(triOrder[nd.left + i]) && want == 2)
            return best;
      }
      else
      {
        if(sp + 2 > 128) { std::fprintf(stderr, "oracle: bvh stack overflow\n"); break; }
        stack[sp++] = nd.left;
        stack[sp++] = nd.right;
      }
    }
    return best;
  }

  // ---- textures (Vulkan unnormalised-coordinate rules, LOD 0 only) --------
  static inline int wrap_coord(int i, int n, int mode)
  {
    if(mode == PT_WRAP_CLAMP_TO_EDGE)
      return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    if(mode == PT_WRAP_MIRRORED_REPEAT)
    {
      int p = 2 * n;
      int m = i % p;
      if(m < 0) m += p;
      m -= n;
      int mir = m >= 0 ? m : -(1 + m);
      return (n - 1) - mir;
    }
    int m = i % n;
    return m < 0 ? m + n : m;
  }

  // Material textures: RGBA8 texels are filtered as 0..255 floats and scaled by (1/255) once.
  vec4 sample_texture(int id, vec2 uv, Stats* st) const
  {
    const Texture& tx = textures[id];
    if(st) st->texTaps++;
    const float s255 = 1.0f / 255.0f;
    float       x = uv.x * float(tx.w), y = uv.y * float(tx.h);
    auto        texel = [&](int ix, int iy) {
      const uint8_t* p = &tx.px[(size_t(wrap_coord(iy, tx.h, tx.wrapT)) * tx.w + wrap_coord(ix, tx.w, tx.wrapS)) * 4];
      return vec4(float(p[0]), float(p[1]), float(p[2]), float(p[3]));
    };
    if(tx.mag == PT_FILTER_NEAREST)
      return texel((int)std::floor(x), (int)std::floor(y)) * s255;
    x -= 0.5f;
    y -= 0.5f;
    float fx = std::floor(x), fy = std::floor(y);
    float a = x - fx, b = y - fy;
    int   x0 = (int)fx, y0 = (int)fy;
    vec4  top = texel(x0, y0) * (1.0f - a) + texel(x0 + 1, y0) * a;
    vec4  bot = texel(x0, y0 + 1) * (1.0f - a) + texel(x0 + 1, y0 + 1) * a;
    return (top * (1.0f - b) + bot * b) * s255;
  }

  // Environment: RGBA32F, LINEAR, U repeat / V clamp-to-edge (src/hdr_sampling.cpp:68-77)
  vec3 sample_env(vec2 uv) const
  {
    float x = uv.x * float(envW) - 0.5f, y = uv.y * float(envH) - 0.5f;
    float fx = std::floor(x), fy = std::floor(y);
    float a = x - fx, b = y - fy;
    int   x0 = (int)fx, y0 = (int)fy;
    auto  texel = [&](int ix, int iy) {
      const float* p = &env[(size_t(wrap_coord(iy, envH, PT_WRAP_CLAMP_TO_EDGE)) * envW + wrap_coord(ix, envW, PT_WRAP_REPEAT)) * 4];
      return vec3(p[0], p[1], p[2]);
    };
    vec3 top = texel(x0, y0) * (1.0f - a) + texel(x0 + 1, y0) * a;
    vec3 bot = texel(x0, y0 + 1) * (1.0f - a) + texel(x0 + 1, y0 + 1) * a;
    return top * (1.0f - b) + bot * b;
  }
};

}  // namespace orc
