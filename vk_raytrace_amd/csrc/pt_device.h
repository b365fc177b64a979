// Device-resident data layout of the gfx950 path tracer (see DESIGN.md "Data layout in HBM").
#pragma once
#include "../../include/pt_types.h"
#include "pt_math.h"

// ---- acceleration structure -------------------------------------------------------------------
// World-space triangle record, 48 B, three aligned 16-byte loads.  Built on device from the node
// transforms (trace contract T1).  Stored in BVH leaf order.
struct TriRec {
  float4 p0w;  // p0.xyz, w = bits: world triangle index (low 29 bits) | flags << 29
  float4 e1n;  // e1.xyz, w = bits: node (TLAS instance) index
  float4 e2p;  // e2.xyz, w = bits: primitive index inside the prim-mesh
};
#define TRI_OPAQUE 1u  // instance FORCE_OPAQUE        (reference: src/accelstruct.cpp:144-146)
#define TRI_NOCULL 2u  // TRIANGLE_FACING_CULL_DISABLE (reference: src/accelstruct.cpp:148-149)
#define TRI_FLIP 4u    // det(instance transform) < 0: facing is an object-space property
#define TRI_INDEX_MASK 0x1fffffffu

// BVH2 node, 64 B: both child boxes live in the parent so one node fetch decides both children.
struct BvhNode {
  float4 a;  // lmin.x lmin.y lmin.z lmax.x
  float4 b;  // lmax.y lmax.z rmin.x rmin.y
  float4 c;  // rmin.z rmax.x rmax.y rmax.z
  uint4  d;  // left, right (bit31 set: leaf -> TriRec slot), 0, 0
};
#define BVH_LEAF 0x80000000u
#define BVH_NONE 0xffffffffu

// ---- scene records -------------------------------------------------------------------------------
// One per TLAS instance (glTF node): what the reference reads through gl_InstanceCustomIndex ->
// InstanceData -> buffer_reference (shaders/host_device.h:200-205) plus the two 4x3 matrices of the
// hit payload (shaders/globals.glsl:53-63), flattened to one 128-byte record.
struct InstanceRec {
  Affine   objectToWorld;
  Affine   worldToObject;
  uint32_t vertexOffset;
  uint32_t firstIndex;
  int32_t  materialIndex;
  int32_t  primMesh;
  uint32_t triBase;  // world index of this node's first triangle
  uint32_t triCount;
  uint32_t flags;    // TRI_* flags shared by all its triangles
  uint32_t _pad;
};

struct TexRec {
  uint32_t offset;  // texel offset into the RGBA8 pool
  int32_t  w, h;
  int32_t  mag, wrapS, wrapT;
  int32_t  _pad[2];
};

struct DeviceScene {
  const float4*               vertices;  // pt_VertexAttributes as 2 x float4
  const uint32_t*             indices;
  const InstanceRec*          instances;
  const pt_GltfShadeMaterial* materials;
  const pt_Light*             lights;
  const TexRec*               texRecs;
  const uint32_t*             texels;  // RGBA8 pool
  const BvhNode*              bvh;
  const TriRec*               tris;
  const float4*               env;  // RGBA32F lat-long
  const pt_EnvAccel*          envAccel;
  int32_t                     envW, envH;
  uint32_t                    numTris;
  uint32_t                    numInstances;
  pt_SceneCamera              camera;
  pt_SunAndSky                sunsky;
};

// ---- wavefront path state (SoA of float4, one slot per local pixel) --------------------------------
struct PathState {
  float4* rayO;    // origin.xyz, -
  float4* rayD;    // direction.xyz, bits(seed)
  float4* thr;     // throughput.xyz, rrPcont
  float4* rad;     // radiance.xyz, -
  float4* absorb;  // absorption.xyz, lightDist
  float4* neeDir;  // lightDir.xyz, visible (1/0)
  float4* neeRad;  // vcontrib.radiance.xyz, -
  float4* hit;     // t, bits(tri slot in leaf order | 0xffffffff miss), u, v
  float4* sum;     // per-frame sample sum (maxSamples > 1)
};

struct Counters {
  unsigned long long closestRays, shadowRays, shadedHits, misses, alphaTests, neeLookups, nodesVisited, trisTested;
  unsigned int       stackOverflow, _pad;
};

struct FrameParams {
  pt_RtxState st;
  int32_t     width, height;  // full image size
  int32_t     tilesX, tilesY;
  int32_t     rank, nranks;
  uint32_t    numLocalTiles;
  uint32_t    numSlots;  // numLocalTiles * 1024
  int32_t     sample;    // index of the sample inside this frame
};
