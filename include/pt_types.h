/*
 * pt_types.h -- plain-C mirror of the structures the reference shares between
 * host and device (reference: shaders/host_device.h:107-281), plus the flat
 * scene description that replaces the reference's descriptor sets at the
 * drop-in boundary (reference: shaders/layouts.glsl:37-52).
 *
 * Layout rule: scalar block layout == packed C with 4-byte alignment.
 * All matrices are column-major (element [col*4 + row]) like GLSL / glm.
 * Sizes are asserted below; SURVEY.md Appendix A lists the dword offsets.
 */
#ifndef PT_TYPES_H
#define PT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_TILE 32 /* framebuffer tile edge in pixels (multi-GPU shard granularity) */

/* reference: shaders/host_device.h:88-102 (DebugMode) */
enum pt_DebugMode {
  PT_DEBUG_NONE = 0,
  PT_DEBUG_BASECOLOR = 1,
  PT_DEBUG_NORMAL = 2,
  PT_DEBUG_METALLIC = 3,
  PT_DEBUG_EMISSIVE = 4,
  PT_DEBUG_ALPHA = 5,
  PT_DEBUG_ROUGHNESS = 6,
  PT_DEBUG_TEXCOORD = 7,
  PT_DEBUG_TANGENT = 8,
  PT_DEBUG_RADIANCE = 9,
  PT_DEBUG_WEIGHT = 10,
  PT_DEBUG_RAYDIR = 11,
  PT_DEBUG_HEATMAP = 12
};

/* reference: shaders/host_device.h:126-131 */
enum { PT_ALPHA_OPAQUE = 0, PT_ALPHA_MASK = 1, PT_ALPHA_BLEND = 2 };
/* reference: shaders/host_device.h:211-213 */
enum { PT_LIGHT_DIRECTIONAL = 0, PT_LIGHT_POINT = 1, PT_LIGHT_SPOT = 2 };

/* reference: shaders/host_device.h:183-196 -- the per-frame kernel config (was a push constant) */
typedef struct pt_RtxState {
  int32_t frame;
  int32_t maxDepth;
  int32_t maxSamples;
  float   fireflyClampThreshold;
  float   hdrMultiplier;
  int32_t debugging_mode;
  int32_t pbrMode; /* 0 Disney, 1 glTF */
  int32_t _pad0;
  int32_t size[2];
  int32_t minHeatmap;
  int32_t maxHeatmap;
} pt_RtxState;

/* reference: shaders/host_device.h:107-115 */
typedef struct pt_SceneCamera {
  float   viewInverse[16];
  float   projInverse[16];
  float   focalDist;
  float   aperture;
  int32_t nbLights;
} pt_SceneCamera;

/* reference: shaders/host_device.h:117-124 */
typedef struct pt_VertexAttributes {
  float    position[3];
  uint32_t normal;      /* 16+16 bit octahedral */
  float    texcoord[2]; /* LSB of texcoord[1] = tangent handedness */
  uint32_t tangent;     /* 16+16 bit octahedral */
  uint32_t color;       /* RGBA8 unorm */
} pt_VertexAttributes;

/* reference: shaders/host_device.h:133-179 */
typedef struct pt_GltfShadeMaterial {
  float   pbrBaseColorFactor[4];
  int32_t pbrBaseColorTexture;
  float   pbrMetallicFactor;
  float   pbrRoughnessFactor;
  int32_t pbrMetallicRoughnessTexture;
  int32_t emissiveTexture;
  int32_t _pad0;
  float   emissiveFactor[3];
  int32_t alphaMode;
  float   alphaCutoff;
  int32_t doubleSided;
  int32_t normalTexture;
  float   normalTextureScale;
  float   uvTransform[16];
  int32_t unlit;
  float   transmissionFactor;
  int32_t transmissionTexture;
  float   ior;
  float   anisotropyDirection[3];
  float   anisotropy;
  float   attenuationColor[3];
  float   thicknessFactor;
  int32_t thicknessTexture;
  float   attenuationDistance;
  float   clearcoatFactor;
  float   clearcoatRoughness;
  int32_t clearcoatTexture;
  int32_t clearcoatRoughnessTexture;
  uint32_t sheen;
  int32_t _pad1;
} pt_GltfShadeMaterial;

/* reference: shaders/host_device.h:215-230 */
typedef struct pt_Light {
  float   direction[3];
  float   range;
  float   color[3];
  float   intensity;
  float   position[3];
  float   innerConeCos;
  float   outerConeCos;
  int32_t type;
  float   padding[2];
} pt_Light;

/* reference: shaders/host_device.h:233-239 */
typedef struct pt_EnvAccel {
  uint32_t alias;
  float    q;
  float    pdf;
  float    aliasPdf;
} pt_EnvAccel;

/* reference: shaders/host_device.h:242-255 */
typedef struct pt_Tonemapper {
  float   brightness;
  float   contrast;
  float   saturation;
  float   vignette;
  float   avgLum;
  float   zoom;
  float   renderingRatio[2];
  int32_t autoExposure;
  float   Ywhite;
  float   key;
  int32_t dither;
} pt_Tonemapper;

/* reference: shaders/host_device.h:258-281 */
typedef struct pt_SunAndSky {
  float   rgb_unit_conversion[3];
  float   multiplier;
  float   haze;
  float   redblueshift;
  float   saturation;
  float   horizon_height;
  float   ground_color[3];
  float   horizon_blur;
  float   night_color[3];
  float   sun_disk_intensity;
  float   sun_direction[3];
  float   sun_disk_scale;
  float   sun_glow_intensity;
  int32_t y_is_up;
  int32_t physically_scaled_sun;
  int32_t in_use;
} pt_SunAndSky;

/* ---- flat scene description (replaces descriptor sets 0 and 2) ---------------------------- */

/* One glTF primitive == one BLAS in the reference (src/accelstruct.cpp:110-127,
 * nvh::GltfPrimMesh fields used at src/scene.cpp:205-252).  Indices are relative
 * to vertexOffset, exactly as the reference's per-primitive index buffers are. */
typedef struct pt_PrimMesh {
  uint32_t vertexOffset;
  uint32_t vertexCount;
  uint32_t firstIndex;
  uint32_t indexCount;
  int32_t  materialIndex;
} pt_PrimMesh;

/* One glTF node == one TLAS instance in the reference (src/accelstruct.cpp:137-159). */
typedef struct pt_Node {
  float   worldMatrix[16]; /* column-major object->world */
  int32_t primMesh;        /* == instanceCustomIndex */
} pt_Node;

/* Sampler state as it reaches the shader (VkFilter / VkSamplerAddressMode numeric values;
 * reference: src/scene.cpp:447-482,561-571). Only the mag filter matters: every tap is LOD 0. */
/* the reference's two Renderer implementations (src/sample_example.hpp:136-137), see pt_set_variant */
enum { PT_VARIANT_RAYQUERY = 0, PT_VARIANT_RTX = 1 };
/* layout of the acceleration structure, see pt_set_accel_mode */
enum { PT_ACCEL_FLAT = 0, PT_ACCEL_TWO_LEVEL = 1 };
/* functions of the fp32 transcendental contract (pt_fpmath.h), for pt_fpmath_eval */
enum { PT_FN_SIN = 0, PT_FN_COS, PT_FN_TAN, PT_FN_ASIN, PT_FN_ACOS, PT_FN_ATAN2, PT_FN_EXP, PT_FN_LOG, PT_FN_POW };
enum { PT_FILTER_NEAREST = 0, PT_FILTER_LINEAR = 1 };
enum { PT_WRAP_REPEAT = 0, PT_WRAP_MIRRORED_REPEAT = 1, PT_WRAP_CLAMP_TO_EDGE = 2 };

typedef struct pt_TextureDesc {
  const uint8_t* rgba8; /* width*height*4, row 0 first (R8G8B8A8_UNORM, src/scene.cpp:493) */
  int32_t        width;
  int32_t        height;
  int32_t        magFilter;
  int32_t        minFilter;
  int32_t        wrapS;
  int32_t        wrapT;
} pt_TextureDesc;

typedef struct pt_SceneDesc {
  const pt_VertexAttributes*  vertices;
  uint32_t                    numVertices;
  const uint32_t*             indices;
  uint32_t                    numIndices;
  const pt_PrimMesh*          primMeshes;
  uint32_t                    numPrimMeshes;
  const pt_Node*              nodes;
  uint32_t                    numNodes;
  const pt_GltfShadeMaterial* materials;
  uint32_t                    numMaterials;
  const pt_Light*             lights; /* may be NULL when numLights == 0 */
  uint32_t                    numLights;
  const pt_TextureDesc*       textures; /* may be NULL when numTextures == 0 */
  uint32_t                    numTextures;
} pt_SceneDesc;

/* Result of pt_pick: the fields of nvvk::RayPickerKHR::PickResult the reference consumes (src/sample_example.cpp:493-510). */
typedef struct pt_PickResult {
  float    worldRayOrigin[3];
  float    hitT;               /* distance along the ray; undefined when nothing is hit */
  float    worldRayDirection[3];
  int32_t  primitiveID;        /* triangle inside the prim-mesh */
  uint32_t instanceID;         /* node (TLAS instance) index, 0xffffffff: nothing hit */
  int32_t  instanceCustomIndex;/* prim-mesh index */
  float    baryCoord[3];       /* (1 - u - v, u, v) */
} pt_PickResult;

/* Counters the measurement contract needs; all totals since pt_reset_stats. */
typedef struct pt_Stats {
  uint64_t samples;          /* pixel-samples rendered */
  uint64_t closestRays;      /* rays through the closest-hit traversal kernel */
  uint64_t shadowRays;       /* rays through the shadow traversal */
  uint64_t shadedHits;       /* surface hits shaded */
  uint64_t misses;           /* environment lookups on miss */
  uint64_t alphaTests;       /* stochastic alpha evaluations */
  uint64_t neeLookups;       /* environment NEE samples */
  uint64_t nodesVisited;     /* BVH nodes popped (device layout; only in PT_STATS builds, else 0) */
  uint64_t trisTested;       /* triangle tests (only in PT_STATS builds, else 0) */
  double   msGenerate;       /* accumulated kernel time by stage, HIP events on the render stream */
  double   msTraceClosest;
  double   msShade;
  double   msTraceShadow;
  double   msAccumulate;
  uint64_t launchesTraceClosest;
  uint32_t numTriangles;     /* world-space triangles in the BVH */
  uint32_t numBvhNodes;
  double   msBuildAccel;     /* last pt_build_accel */
  uint64_t bytesScene;       /* resident HBM bytes of scene+BVH+textures+env */
  uint64_t bytesAccel;       /* of which the acceleration structure (nodes + leaf records + any-hit records) */
  uint32_t numBlas;          /* PT_ACCEL_TWO_LEVEL: bottom-level structures (prim-meshes instantiated), else 0 */
  uint32_t numTlasNodes;     /* PT_ACCEL_TWO_LEVEL: nodes of the instance hierarchy, else 0 */
  double   msTail;           /* kernel time of the fused late bounces (closest + shade + shadow of small queues in one launch) */
  uint32_t batchFrames;      /* frames traced as one wavefront after the path-state budget was applied (pt_resize) */
  uint32_t framesInFlight;   /* frame slots (a full batch each) overlapped on separate streams after the budget was applied; the one-frame display slots are not counted */
  uint64_t tailClosestRays;  /* the part of closestRays / shadowRays / shadedHits / misses / alphaTests that the fused late-bounce kernel processed */
  uint64_t tailShadowRays;
  uint64_t tailShadedHits;
  uint64_t tailMisses;
  uint64_t tailAlphaTests;
  uint64_t launchesTail;     /* launches of the fused late-bounce kernel (with profiling enabled, like launchesTraceClosest) */
  uint64_t numMergedTriangles; /* two-level mode: triangles of the prim-meshes instantiated once, kept in ONE world-space bottom-level structure
                                  (counted as one of numBlas); equal to numTriangles when the scene has no repeated mesh -- the flat kernels then run on it */
  double   msTraceFused;     /* kernel time of the fused trace stage: shadow rays of bounce b + closest-hit rays of bounce b + 1 in one launch
                                (msTraceClosest / msTraceShadow then hold bounce 0's closest-hit stage and the last staged bounce's shadow stage) */
  uint64_t launchesTraceFused;
} pt_Stats;

/* pt_measure_peaks: ceilings measured on the device */
typedef struct pt_Peaks {
  double  valuWaveInstrPerSec; /* wave64 VALU instructions issued per second, whole chip */
  double  hbmCopyBytesPerSec;  /* read + written bytes per second of a streaming copy */
  double  hbmReadBytesPerSec;  /* bytes per second of a streaming read */
  int32_t computeUnits;
  int32_t clockMHz;
} pt_Peaks;

#ifdef __cplusplus
}
static_assert(sizeof(pt_RtxState) == 48, "RtxState");
static_assert(sizeof(pt_SceneCamera) == 140, "SceneCamera");
static_assert(sizeof(pt_VertexAttributes) == 32, "VertexAttributes");
static_assert(sizeof(pt_GltfShadeMaterial) == 216, "GltfShadeMaterial");
static_assert(sizeof(pt_Light) == 64, "Light");
static_assert(sizeof(pt_EnvAccel) == 16, "EnvAccel");
static_assert(sizeof(pt_Tonemapper) == 48, "Tonemapper");
static_assert(sizeof(pt_SunAndSky) == 96, "SunAndSky");
static_assert(sizeof(pt_PrimMesh) == 20, "PrimMesh");
static_assert(sizeof(pt_Node) == 68, "Node");
#endif

#endif /* PT_TYPES_H */
