// Device-resident data layout of the gfx950 path tracer (see DESIGN.md "Data layout in HBM").
#pragma once
#include "../../include/pt_types.h"
#include "pt_math.h"
#include <cstring>

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
#define BVH_ALPHA 0x40000000u  // leaf reference: the triangle is non-opaque; inner reference: the subtree holds non-opaque triangles
#define BVH_SLOT_MASK 0x3fffffffu
#define BVH_NONE 0xffffffffu

// Any-hit inputs of a triangle, in leaf order: what the reference's HitTest gathers through
// InstanceData -> indices -> 3 x VertexAttributes (shaders/traceray_rq.glsl:62-79), flattened to one 32-byte record
// so that an opacity evaluation is 3 dependent loads deep instead of 5.
struct AlphaRec {
  float    uv0[2], uv1[2], uv2[2];  // raw texcoords (handedness bit left in place, Appendix C-8)
  uint32_t material;
  uint32_t _pad;
};
// The alpha-relevant part of a material + its base-colour texture descriptor (80 B, one per material, cache resident).
struct AlphaMat {
  float    factorA, cutoff;
  int32_t  mode, tex;             // alphaMode, pbrBaseColorTexture (-1: none)
  float    m[8];                  // uvTransform columns 0 and 1 (the two the (u,v) result needs)
  uint32_t texOffset;
  int32_t  texW, texH, texMag;    // texWrap = wrapS | wrapT << 8 | pot << 16 | ALPHA_FAST_TAP (REPEAT x REPEAT, both sizes 2^k)
  int32_t  texWrap;
  uint32_t mapOffset;             // first word of this material's opacity map in DeviceScene::alphaMaps, ALPHA_NO_MAP: none
  uint32_t _pad[2];               // 80 bytes: five aligned quads, fetched together (pt_surface.h opacity_eval)
};
static_assert(sizeof(AlphaMat) == 80, "AlphaMat is read as five 16-byte quads");
#define ALPHA_FAST_TAP (1 << 24)
#define ALPHA_TILED (1 << 25)  // the texture is stored block-linear (TexRec::tiled)
#define ALPHA_NO_MAP 0xffffffffu
// Opacity map: a conservative 2-bit classification of every ALPHA_MAP_BLOCK^2 block of base texels (+ a one-texel
// apron, so that it bounds every bilinear tap whose base texel lies in the block) under the material's factor /
// cutoff / mode: the any-hit evaluation of most candidates is then one word fetch instead of four texel taps.
// Exactness: a block is classified only when every texel it can blend decides the same way with a 1e-5 relative
// margin (fp32 filtering error is < 1e-6); everything else is ALPHA_ST_UNKNOWN and evaluated in full.
#define ALPHA_MAP_SHIFT 2
#define ALPHA_MAP_BLOCK (1 << ALPHA_MAP_SHIFT)
#define ALPHA_ST_UNKNOWN 0u
#define ALPHA_ST_ZERO 1u    // opacity <= 0 everywhere
#define ALPHA_ST_ONE 2u     // opacity >= 1 everywhere

// Wide node (PT_BVH_WIDTH = 4 or 8 children), collapsed on device from the binary LBVH by surface area.  The
// traversal is bound by dependent memory round trips, not ALU, so fewer / fatter steps win: one node fetch
// (W/4 * 7 aligned 16-byte loads, all in flight together) decides W children.  SoA inside the node.
#ifndef PT_BVH_WIDTH
#define PT_BVH_WIDTH 4
#endif
#define PT_WIDE_Q (PT_BVH_WIDTH / 4)
struct WideNode {
  float4 minx[PT_WIDE_Q], miny[PT_WIDE_Q], minz[PT_WIDE_Q];
  float4 maxx[PT_WIDE_Q], maxy[PT_WIDE_Q], maxz[PT_WIDE_Q];
  uint4  child[PT_WIDE_Q];   // bit31: leaf -> TriRec slot; BVH_NONE: empty slot (its box is inverted)
  uint4  pad[PT_WIDE_Q];     // pads the node to 128 B (W=4) / 256 B (W=8)
};

// The same node in 80 bytes = five 16-byte requests instead of seven (PT_TUNE cnodes=1; read by the persistent trace kernels of the flat
// structure only, converted from the WideNode array after the build -- same node numbering).  The child boxes sit on a per-node grid: origin
// p (the lower corner of the union of the children's boxes), one power-of-two step per axis (exponent byte e: step 2^(e-127), 2047 steps cover
// the extent), the planes as fp16 INTEGERS 0 .. 2047 -- lower planes rounded down, upper planes rounded up, so the decoded box encloses the
// fp32 box.  A plane's ray parameter is q * (step * idir) + (p * idir + n): one v_fma_mix_f32 (the fp16 operand is read out of a register half).
struct CompactNode {
  float    px, py, pz;
  uint32_t exps;    // ex | ey << 8 | ez << 16
  uint4    ax[3];   // per axis: x = lo0 | lo1 << 16, y = lo2 | lo3 << 16, z = hi0 | hi1 << 16, w = hi2 | hi3 << 16
  uint4    child;   // as WideNode::child
};
#define CN_GRID_MAX 2047

// ---- two-level acceleration structure (PT_ACCEL_TWO_LEVEL; reference: src/accelstruct.cpp:110-162) -----------------
// One BLAS per prim-mesh in OBJECT space (its WideNodes and leaf records are shared by every instance of the mesh) and one TLAS over
// the instances' world boxes.  DeviceScene::wide / tris / alphaRecs then hold the concatenated BLASes (child references and leaf slots
// are global indices into them), DeviceScene::tlas the instance hierarchy.  A BLAS leaf record keeps the three OBJECT-space vertex
// positions (p0w.xyz, e1n.xyz, e2p.xyz; p0w.w = primitive index): the triangle test transforms them with the instance matrix exactly as
// trace contract T1 does and runs in world space, so hits are bit-identical to the flat structure; only the box tests happen in object space.
// The prim-meshes a scene instantiates exactly once share ONE bottom-level structure in world space (pt_capi.hip build_two_level): its leaf
// records are the flat structure's (world-space edge form, instance and primitive in the w lanes), its TLAS leaf carries this instance id, and
// a lane inside it keeps its world-space ray constants -- no ray transform on entry, no vertex transform per triangle.
#define PT_INST_MERGED 0xfffffffeu
#define PT_INST_BLOCK_SHIFT 8
struct TlasLeaf {  // 32 B, one per TLAS leaf (= non-empty instance), TLAS leaf order
  uint32_t inst;      // instance (glTF node) index
  uint32_t nodeBase;  // root WideNode of the instance's BLAS
  uint32_t wflags;    // world index of the instance's first triangle | TRI_* flags << 29
  uint32_t _pad0;
  float    padC0, padC1;  // object-space box padding for a ray with origin o: padC1 * max|o| + padC0 (covers the rounding of the ray transform)
  uint32_t _pad1[2];
};

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
  int32_t  pot;  // bit0: w is a power of two, bit1: h is
  int32_t  tiled;  // bit 0: block-linear storage (see tex_index), else row-major; bits 8-9: layers - 1, bits 10-11: this texture's layer of an
                   // INTERLEAVED group (round 5): the textures a material samples with one (u, v) -- normal, emissive, metallic-roughness, base colour --
                   // stored texel by texel next to each other when they have the same size and sampler, so that the 2 x 2 footprints of a shading's
                   // taps share cache lines (texel (ix, iy) of layer l: offset + tex_index(...) x layers + l)
};
// Texel (ix, iy) of a w-texel-wide image -> index into its storage.  Block-linear images (w % 8 == 0, h % 4 == 0) keep every 8 x 4-texel tile in one
// 128-byte line, so the 2 x 2 footprint of a bilinear tap is ONE line two times out of three instead of always two (rows w texels apart).
// A storage order only: the texel VALUES and the filter arithmetic are untouched (k_shade is bound by the lines it pulls through L2).
#define PT_TEX_TILE_W 8
#define PT_TEX_TILE_H 4
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t tex_index(int32_t w, int32_t ix, int32_t iy, bool tiled)
{
  // one multiply for both orders, selects instead of a branch per texel: index = row x stride + rest, with (row, stride, rest) =
  // (iy / 4, 4 w, tile column x 32 + offset in the tile) block-linear and (iy, w, ix) row-major.  Image sides are < 2^16: a 24-bit multiply does.
  const uint32_t ux = uint32_t(ix), uy = uint32_t(iy), uw = uint32_t(w);
  const uint32_t row = tiled ? (uy >> 2) : uy, stride = tiled ? (uw << 2) : uw;
  const uint32_t rest = tiled ? (((ux >> 3) << 5) | ((uy & 3u) << 3) | (ux & 7u)) : ux;
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(row, stride) + rest;
#else
  return row * stride + rest;
#endif
}

// A texture's descriptor in 16 bytes (pt_surface.h resolve_material), so that a material's four descriptors cost 16 registers, not 32:
// x = texel offset, y = w | h << 16 (sides < 2^16, pt_set_scene checks), z = bit 0 NEAREST, bits 1-2 wrapS, bits 3-4 wrapT, bits 5-6 pot, bit 7 tiled,
// bits 8-9 layers - 1, bits 10-11 layer (TexRec::tiled).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint4 tex_desc_pack(const TexRec& t)
{
  uint4 d;
  d.x = t.offset;
  d.y = uint32_t(t.w) | (uint32_t(t.h) << 16);
  d.z = (t.mag == 0 ? 1u : 0u) | (uint32_t(t.wrapS) << 1) | (uint32_t(t.wrapT) << 3) | (uint32_t(t.pot & 3) << 5) | ((t.tiled & 1) ? 128u : 0u) | (uint32_t(t.tiled) & 0xf00u);
  d.w = 0;
  return d;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline TexRec tex_desc_unpack(const uint4& d)
{
  TexRec t;
  t.offset = d.x;
  t.w      = int32_t(d.y & 0xffffu);
  t.h      = int32_t(d.y >> 16);
  t.mag    = (d.z & 1u) ? 0 : 1;  // PT_FILTER_NEAREST = 0, PT_FILTER_LINEAR = 1 (include/pt_types.h)
  t.wrapS  = int32_t((d.z >> 1) & 3u);
  t.wrapT  = int32_t((d.z >> 3) & 3u);
  t.pot    = int32_t((d.z >> 5) & 3u);
  t.tiled  = int32_t(((d.z >> 7) & 1u) | (d.z & 0xf00u));
  return t;
}

// One aligned 128-byte line per material for k_shade (round 5): quads 0-2 the "hot" part of the 216-byte pt_GltfShadeMaterial -- base colour factor |
// emissive factor, normal-texture scale | roughness, metallic, ior, flags -- and quads 3-6 the 16-byte descriptors of its normal / emissive /
// metallic-roughness / base-colour textures (texture 0 for an absent one).  A material whose remaining fields are the importer's defaults (no KHR
// transmission / clearcoat / sheen / anisotropy / volume, identity texture transform, lit: MAT_SIMPLE) is shaded from the line alone -- seven requests
// in one cache line instead of fourteen into the 216-byte record plus four descriptors; any other material still reads its full record.
// The values are the record's own floats and the shading code is the same function on a record rebuilt in registers: identical results.
#define PT_MAT_LINE_QUADS 8
#define MAT_HAS_NORMAL 1u
#define MAT_HAS_EMISSIVE 2u
#define MAT_HAS_MR 4u
#define MAT_HAS_BASE 8u
#define MAT_SIMPLE 16u
inline bool mat_is_simple(const pt_GltfShadeMaterial& m)
{
  const float id[8] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f};  // the two rows of uvTransform the shading reads (pt_surface.h resolve_material)
  for(int k = 0; k < 8; ++k)
    if(std::memcmp(&m.uvTransform[k], &id[k], 4) != 0)
      return false;
  const float one = 1.0f, zero = 0.0f, big = 3.4028235e38f;
  auto eq = [](float a, float b) { return std::memcmp(&a, &b, 4) == 0; };
  return m.unlit == 0 && eq(m.transmissionFactor, zero) && m.transmissionTexture == -1 && eq(m.anisotropy, zero) && eq(m.anisotropyDirection[0], zero) &&
         eq(m.anisotropyDirection[1], one) && eq(m.anisotropyDirection[2], zero) && eq(m.attenuationColor[0], one) && eq(m.attenuationColor[1], one) &&
         eq(m.attenuationColor[2], one) && eq(m.thicknessFactor, zero) && eq(m.attenuationDistance, big) && eq(m.clearcoatFactor, zero) &&
         eq(m.clearcoatRoughness, zero) && m.clearcoatTexture == -1 && m.clearcoatRoughnessTexture == -1 && m.sheen == 0u;
}
// rec[k]: the record through which the material's normal / emissive / metallic-roughness / base-colour texture is read (its plain storage or its layer of
// an interleaved group; the scene's record 0 for an absent one)
inline void mat_line_pack(const pt_GltfShadeMaterial& m, const TexRec rec[4], uint4 out[PT_MAT_LINE_QUADS])
{
  auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  const int      ids[4] = {m.normalTexture, m.emissiveTexture, m.pbrMetallicRoughnessTexture, m.pbrBaseColorTexture};
  const uint32_t flags  = (ids[0] > -1 ? MAT_HAS_NORMAL : 0u) | (ids[1] > -1 ? MAT_HAS_EMISSIVE : 0u) | (ids[2] > -1 ? MAT_HAS_MR : 0u) | (ids[3] > -1 ? MAT_HAS_BASE : 0u) |
                         (mat_is_simple(m) ? MAT_SIMPLE : 0u);
  out[0] = uint4{bits(m.pbrBaseColorFactor[0]), bits(m.pbrBaseColorFactor[1]), bits(m.pbrBaseColorFactor[2]), bits(m.pbrBaseColorFactor[3])};
  out[1] = uint4{bits(m.emissiveFactor[0]), bits(m.emissiveFactor[1]), bits(m.emissiveFactor[2]), bits(m.normalTextureScale)};
  out[2] = uint4{bits(m.pbrRoughnessFactor), bits(m.pbrMetallicFactor), bits(m.ior), flags};
  for(int k = 0; k < 4; ++k)
    out[3 + k] = tex_desc_pack(rec[k]);
  out[7] = uint4{0u, 0u, 0u, 0u};
}

#define PT_SHADE_REC_QUADS 8
struct DeviceScene {
  const float4*               vertices;  // pt_VertexAttributes as 2 x float4
  const uint32_t*             indices;
  const InstanceRec*          instances;
  const pt_GltfShadeMaterial* materials;
  const pt_Light*             lights;
  const TexRec*               texRecs;
  const uint4*                matLines;  // per material one 128-byte line (mat_line_pack): hot fields + the descriptors of its four common textures
  const uint32_t*             texels;  // RGBA8 pool
  const BvhNode*              bvh;   // binary LBVH (build product; traversed only when PT_BVH_WIDTH == 2)
  const WideNode*             wide;  // collapsed wide BVH
  const CompactNode*          cnodes;  // its nodes in the compact form (nullptr: none)
  const float4*               shadeTris;  // flat structure only (else nullptr): per leaf slot ONE 128-byte line (PT_SHADE_REC_QUADS float4): the six float4 of the
                                          // triangle's three pt_VertexAttributes, then (instance, primitive) -- everything k_shade needs of the hit
                                          // triangle in one aligned line instead of a 48-byte TriRec (1.4 lines) + 96 bytes at a 96-byte stride (1.7 lines).  Was: next to
                                          // each other (96 B): k_shade reads them with the hit's slot, together with the triangle record, instead of
                                          // after instance -> index triple -> three vertices (two dependent round trips fewer per shading)
  const TriRec*               tris;
  const AlphaRec*             alphaRecs;  // leaf order, parallel to tris
  const AlphaMat*             alphaMats;  // one per material
  const uint32_t*             alphaMaps;  // opacity maps, 16 blocks per word
  const float4*               env;  // RGBA32F lat-long
  const pt_EnvAccel*          envAccel;
  int32_t                     envW, envH;
  uint32_t                    numTris;
  uint32_t                    numInstances;
  pt_SceneCamera              camera;
  pt_SunAndSky                sunsky;
  float                       boundsMin[3];     // world bounds of the triangles (ray-sort keys: origin cell)
  float                       boundsInvExt[3];  // 1 / extent per axis (0 for a flat axis)
  // two-level mode (null / 0 otherwise)
  const WideNode*             tlas;         // instance hierarchy; its leaf references index tlasLeaves
  const CompactNode*          ctlas;        // its nodes in the compact form (nullptr: none); DeviceScene::cnodes then covers the bottom-level structures
  const TlasLeaf*             tlasLeaves;
  const uint32_t*             instTriBase;  // InstanceRec::triBase of every instance, compact (world triangle index -> instance)
  const uint32_t*             instBlock;    // [(numTris >> PT_INST_BLOCK_SHIFT) + 2]: entry e = the last instance whose triBase <= e << PT_INST_BLOCK_SHIFT, so that
                                            // the search for a world triangle's instance starts inside a handful of candidates (nullptr: search all)
  uint32_t                    twoLevel;
  uint32_t                    allOpaque;    // 1: no instance with triangles lacks TRI_OPAQUE (scene without MASK / BLEND materials, or pt_use_any_hit(0)):
                                            // no candidate ever draws, so a shadow ray may stop at the first hit it finds (TerminateOnFirstHit,
                                            // shaders/traceray_rq.glsl:157) -- which hit commits first cannot change its result
};

// ---- wavefront path state (SoA of float4, one slot per local pixel) --------------------------------
// The state is streamed through once per stage (11 GB per batch); the scene's working set (textures, structure, environment: a few hundred MB) is
// what the caches should keep.  The arrays are thin proxies that carry the cache policy of every path-state access in their TYPE, so that the
// access sites read like plain arrays and one function body (shade_path, finish_bounce_core, store_hit) serves kernels with different policies:
//   PT_STATE_PLAIN     plain loads / stores (the host build of the shading source, tests/cpp/trace_host.cpp)
//   PT_STATE_NT        non-temporal (`nt`) loads and stores: the staged kernels (+2 % at 96 steps against plain, profiles/r05d_*)
// (sc1 buffer loads / write-through sc1 stores -- what data handed from wavefront to wavefront INSIDE one launch needs, MI355X_MICROARCH.md "inter-workgroup
// visibility" -- measured 3.5 % slower than nt as the policy of every kernel, sc1 stores alone 5.6 %: they drop the line from L2; profiles/r05d_*.  The
// in-launch scheduler that needed them, k_wave, lost to k_tail: profiles/r05_kwave_experiment.txt, branch kwave-experiment.)
#define PT_STATE_PLAIN 0
#define PT_STATE_NT 1
#ifndef PT_STATE_POLICY
#define PT_STATE_POLICY PT_STATE_NT
#endif
template <int POL>
struct StateArrayT {  // the plain form: a pointer
  float4* p;
  PT_DEV float4& operator[](size_t i) const { return p[i]; }
  PT_DEV operator float4*() const { return p; }
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef float pt_nt_f4 __attribute__((ext_vector_type(4)));
struct StateF1RefNt {
  float* p;
  PT_DEV operator float() const { return __builtin_nontemporal_load(p); }
  PT_DEV StateF1RefNt& operator=(float v) { __builtin_nontemporal_store(v, p); return *this; }
  PT_DEV StateF1RefNt& operator+=(float v) { __builtin_nontemporal_store(__builtin_nontemporal_load(p) + v, p); return *this; }
};
struct StateF4RefNt {
  float4*      p;
  StateF1RefNt x, y, z, w;
  PT_DEV explicit StateF4RefNt(float4* q) : p(q), x{&q->x}, y{&q->y}, z{&q->z}, w{&q->w} {}
  PT_DEV operator float4() const
  {
    const pt_nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const pt_nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  PT_DEV StateF4RefNt& operator=(const float4& v)
  {
    pt_nt_f4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<pt_nt_f4*>(p));
    return *this;
  }
};
template <>
struct StateArrayT<PT_STATE_NT> {
  float4* p;
  PT_DEV StateF4RefNt operator[](size_t i) const { return StateF4RefNt(p + i); }
  PT_DEV operator float4*() const { return p; }
};
#endif
typedef StateArrayT<PT_STATE_POLICY> StateArray;
template <int POL>
struct PathStateT {
  StateArrayT<POL> rayO;    // origin.xyz, -
  StateArrayT<POL> rayD;    // direction.xyz, bits(seed)
  StateArrayT<POL> thr;     // throughput.xyz, rrPcont
  StateArrayT<POL> rad;     // radiance.xyz, -
  StateArrayT<POL> absorb;  // absorption.xyz, lightDist
  StateArrayT<POL> neeDir;  // lightDir.xyz, visible (1/0)
  StateArrayT<POL> neeRad;  // vcontrib.radiance.xyz, -
  StateArrayT<POL> hit;     // t, bits(tri slot in leaf order | 0xffffffff miss), u, v
  StateArrayT<POL> sum;     // per-frame sample sum (maxSamples > 1)
};
typedef PathStateT<PT_STATE_POLICY> PathState;

struct Counters {
  unsigned long long closestRays, shadowRays, shadedHits, misses, alphaTests, neeLookups, nodesVisited, trisTested;
  unsigned int       stackOverflow, _pad;
  unsigned long long tailClosestRays, tailShadowRays, tailShadedHits, tailMisses, tailAlphaTests;  // the part of the totals above that k_tail traced / shaded
};

struct FrameParams {
  pt_RtxState st;
  int32_t     width, height;  // full image size
  int32_t     tilesX, tilesY;
  int32_t     rank, nranks;
  uint32_t    numLocalTiles;
  uint32_t    numSlots;  // numLocalTiles * 1024 (path slots of ONE frame)
  uint32_t    batch;     // consecutive frames (st.frame, st.frame + 1, ...) traced together: path slot = f * numSlots + pixel slot
  int32_t     sample;    // index of the sample inside this frame
  int32_t     variant;   // PT_VARIANT_RAYQUERY / PT_VARIANT_RTX (include/pt_api.h)
  int32_t     regen;     // 1: k_generate only builds the queue; the packet kernel of bounce 0 computes the camera rays itself (pt_render.hip plan_frame decides)
};
