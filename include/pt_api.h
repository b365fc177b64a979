/*
 * pt_api.h -- C ABI of libptmi.so, the MI355X (gfx950) path-tracing backend that sits where the
 * reference's Vulkan renderers sit (reference: src/renderer.h:30-48; the two implementations it
 * replaces are src/rayquery.cpp:40-109 and src/rtx_pipeline.cpp:45-276).
 *
 * Conventions
 *  - every entry point returns pt_Status (0 == ok, negative == error); the text of the last error
 *    is available from pt_last_error().  Nothing here aborts the process.  (The reference has no
 *    error convention at all: its methods are void and VkResults are asserted,
 *    src/rtx_pipeline.cpp:209,237.)
 *  - a context is externally synchronised: any thread may call, never two at once.
 *  - host arrays passed to pt_set_* are copied before the call returns.
 *  - pt_render_frame is asynchronous on the context's HIP stream; pt_read_accum, pt_tonemap,
 *    pt_get_stats and pt_synchronize wait for it.
 *  - there is no CPU fallback: without a gfx950 device pt_create fails with PT_ERR_NO_DEVICE.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>
#include "pt_types.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* libptmi.so is built with -fvisibility=hidden: exactly these entry points are exported */
#endif

typedef struct pt_context pt_context;

typedef enum pt_Status {
  PT_OK            = 0,
  PT_ERR_INVALID   = -1, /* bad argument / inconsistent scene description */
  PT_ERR_NO_DEVICE = -2, /* no HIP device with that ordinal (or not gfx950) */
  PT_ERR_HIP       = -3, /* a HIP runtime call failed; see pt_last_error */
  PT_ERR_STATE     = -4, /* call order violated (e.g. render before build_accel) */
  PT_ERR_OOM       = -5,
  PT_ERR_UNAVAILABLE = -6 /* an optional runtime library is missing (librccl.so for the pt_comm_* / pt_gather_* calls); see pt_comm_last_error */
} pt_Status;


/* ---- lifetime ------------------------------------------------------------------------------ */

/* replaces Renderer::setup(device, physicalDevice, familyIndex, allocator)  [src/renderer.h:33-36,
 * src/rayquery.cpp:40-46]: binds the context to one GPU and creates its stream. */
int pt_create(int device_ordinal, pt_context** out_ctx);

/* replaces Renderer::destroy() [src/renderer.h:37, src/rayquery.cpp:51-58] plus the destroy() of
 * Scene / AccelStructure / HdrSampling / RenderOutput, whose device memory the context owns. */
int pt_destroy(pt_context* ctx);

/* "HIP" -- replaces Renderer::name() [src/renderer.h:43; "RQ" src/rayquery.hpp:51, "Rtx" src/rtx_pipeline.hpp:57] */
const char* pt_renderer_name(void);

const char* pt_last_error(const pt_context* ctx); /* ctx may be NULL: last pt_create error */

/* ---- scene (descriptor set 2 of the reference, shaders/layouts.glsl:42-46) ------------------- */

/* replaces the uploads done by Scene::load [src/scene.cpp:56-118]: createMaterialBuffer :339-382,
 * createLightBuffer :304-333, createTextureImages :488-580, createVertexBuffer :190-274,
 * createInstanceDataBuffer :161-176.  The arrays are the ones those functions upload. */
int pt_set_scene(pt_context* ctx, const pt_SceneDesc* scene);

/* replaces AccelStructure::create [src/accelstruct.cpp:55-65] (BLAS per prim-mesh :110-127, TLAS
 * instance per node with opaque / cull-disable flags :132-162): builds the device LBVH over the
 * world-space triangles of every node. */
int pt_build_accel(pt_context* ctx);

/* Layout of the acceleration structure pt_build_accel creates.  PT_ACCEL_FLAT (default): one hierarchy over the world-space triangles of
 * every node (an instanced mesh is stored once per instance).  PT_ACCEL_TWO_LEVEL: the reference's own shape [src/accelstruct.cpp:110-127
 * createBottomLevelAS: one BLAS per prim-mesh; :132-162 createTopLevelAS: one TLAS instance per node carrying node.worldMatrix] -- every
 * prim-mesh is built once in object space and shared by its instances, the top level holds the instances' world boxes.  Rendered pixels are
 * bit-identical in both modes (the triangle test runs on the same world-space vertices).  Rebuilds if a structure exists. */
int pt_set_accel_mode(pt_context* ctx, int mode);

/* New world matrices for the nodes of the current scene (same count, same primMesh per node), e.g. an animated or edited scene: the
 * counterpart of rebuilding the reference's TLAS from node.worldMatrix [src/accelstruct.cpp:137-161] without touching the BLASes
 * (nvvk::RaytracingBuilderKHR::buildTlas with update = true).  PT_ACCEL_TWO_LEVEL refits: instance boxes + instance hierarchy only.
 * PT_ACCEL_FLAT rebuilds the whole structure.  Frames already handed to pt_render_frame keep the old transforms. */
int pt_update_instances(pt_context* ctx, const pt_Node* nodes, uint32_t num_nodes);

/* replaces Scene::updateCamera's UBO upload [src/scene.cpp:629-668] */
int pt_set_camera(pt_context* ctx, const pt_SceneCamera* cam);

/* ---- environment (descriptor set 3, shaders/layouts.glsl:48-50) ------------------------------ */

/* replaces HdrSampling::loadEnvironment after stbi_loadf [src/hdr_sampling.cpp:56-99]: uploads the
 * RGBA32F lat-long image (row 0 = +Y pole) and the alias table built by createEnvironmentAccel
 * [:187-248]; returns getIntegral()/getAverage() [src/hdr_sampling.hpp:44-45].  The caller sets
 * fireflyClampThreshold = 4 * integral like src/sample_example.cpp:110. */
int pt_set_env(pt_context* ctx, const float* rgba32f, int width, int height, float* out_integral, float* out_average);

/* replaces stbi_loadf(file, &w, &h, &comp, STBI_rgb_alpha) in HdrSampling::loadEnvironment [src/hdr_sampling.cpp:56-66] for Radiance
 * RGBE files (.hdr): row 0 first, RGBA32F with alpha 1, value = mantissa * 2^(exponent - 136) like stb_image; flat and run-length
 * encoded scanlines.  Host only (no GPU).  The image is malloc'ed: free it with pt_hdr_free (the reference: stbi_image_free), then hand
 * it to pt_set_env.  On failure returns PT_ERR_INVALID and writes a message to `err` (may be NULL). */
int  pt_hdr_load(const char* path, float** out_rgba32f, int* out_width, int* out_height, char* err, size_t err_len);
void pt_hdr_free(float* rgba32f);

/* replaces the SunAndSky UBO update in SampleExample::updateUniformBuffer [src/sample_example.cpp:168-178] */
int pt_set_sunsky(pt_context* ctx, const pt_SunAndSky* ss);

/* ---- output (descriptor set 1) --------------------------------------------------------------- */

/* replaces Renderer::create(size, ...) [src/renderer.h:42] + RenderOutput::update(size)
 * [src/render_output.cpp:88-99]: (re)allocates the RGBA32F accumulation tiles and the path state. */
int pt_resize(pt_context* ctx, int width, int height);

/* Image-tile sharding for multi-GPU runs (no reference counterpart).  Rank r of n
 * renders the PT_TILE x PT_TILE tiles with (tx + ty) % n == r; seeds use the global pixel index so
 * the pixels are bit-identical to a 1-GPU render.  Default (0, 1).  Call before pt_resize. */
int pt_set_shard(pt_context* ctx, int rank, int nranks);

/* Selects which of the reference's two Renderer implementations the frames reproduce (SampleExample::RndMethod,
 * src/sample_example.hpp:136-137).  PT_VARIANT_RAYQUERY (default): the compute / ray-query path [src/rayquery.cpp,
 * shaders/pathtrace.comp].  PT_VARIANT_RTX: the ray-tracing-pipeline path [src/rtx_pipeline.cpp, shaders/pathtrace.rgen +
 * .rchit / .rahit / .rmiss], which differs in two observable ways: the per-pixel seed is tea(y * W + x, frame) without the
 * maxSamples factor (pathtrace.rgen:72 vs pathtrace.comp:97), and the stochastic alpha tests of a SHADOW ray draw from a copy
 * of the path's seed, so they do not advance it (traceray_rtx.glsl:54-55 with pathtrace.rahit:44,112).  Everything else
 * (samplePixel, PathTrace, the BSDFs) is the same pathtrace.glsl. */
/* (enum PT_VARIANT_* lives in pt_types.h) */
int pt_set_variant(pt_context* ctx, int variant);

/* replaces RtxPipeline::useAnyHit(enable) [src/rtx_pipeline.cpp:269-276, GUI: src/sample_gui.cpp:140-148].  enable == 0 removes the
 * any-hit stage: every triangle is treated as opaque (no stochastic alpha test, no random draw), which "can be faster, but the scene must
 * be fully opaque".  Default 1.  Rebuilds the acceleration structure if one exists (the reference re-creates its pipeline). */
int pt_use_any_hit(pt_context* ctx, int enable);

/* replaces Renderer::setPushContants(state) + Renderer::run(cmdBuf, size, profiler, descSets)
 * [src/renderer.h:38-45; src/rayquery.cpp:97-109 -> shaders/pathtrace.comp:87-134]: renders
 * state->maxSamples samples for every local pixel and folds them into the accumulation buffer with
 * the reference's running mean (pathtrace.comp:122-133).  state->size must equal the pt_resize size. */
int pt_render_frame(pt_context* ctx, const pt_RtxState* state);

int pt_synchronize(pt_context* ctx);

/* Reads the linear RGBA32F accumulation image, row-major width*height*4 floats (alpha == 1).
 * With nranks > 1 only locally owned pixels are valid unless pt_scatter_shards was called. */
int pt_read_accum(pt_context* ctx, float* rgba32f_out);

/* Checkpoint restore (no reference counterpart; SURVEY.md section 5 "Checkpoint / resume": the progressive state of the reference is the
 * accumulation image plus RtxState.frame, src/sample_example.cpp:183-207).  Replaces the accumulation image by a row-major
 * width*height*4 float image previously obtained from pt_read_accum; continuing with frame = N reproduces an uninterrupted run bit for bit
 * (the running mean of pathtrace.comp:122-133 only needs the image and the frame index). */
int pt_write_accum(pt_context* ctx, const float* rgba32f_in);

/* replaces RenderOutput::genMipmap + RenderOutput::run [src/render_output.cpp:174-193 -> shaders/post.frag:98-147]: tonemaps the
 * accumulation image into row-major RGBA8.  Tonemapper.autoExposure bit 0 takes the image average from the 1x1 level of the
 * vkCmdBlitImage(LINEAR) mip chain the reference generates, bit 1 selects toneLocalExposure (post.frag:72-96) on that chain. */
int pt_tonemap(pt_context* ctx, const pt_Tonemapper* tm, uint8_t* rgba8_out);

/* The display pass while the viewer de-scales [src/sample_example.cpp:378,410-413; shaders/post.frag:101]: the reference renders
 * (W / level) x (H / level) pixels into the top-left corner of its W x H offscreen image and magnifies them with Tonemapper.zoom =
 * 1 / level (NEAREST sampler).  Here the context is resized to the reduced size and this call produces the disp_width x disp_height
 * RGBA8 viewport from it (texels of the offscreen image outside the rendered corner count as zero; the reference keeps stale data
 * there).  With disp == accumulation size and zoom == 1 it is pt_tonemap. */
int pt_tonemap_zoom(pt_context* ctx, const pt_Tonemapper* tm, int disp_width, int disp_height, uint8_t* rgba8_out);

/* The display pass without a host stall (the reference's display loop keeps several frames in flight: src/main.cpp:201-264 only waits for a
 * free swapchain image in prepareFrame() :213 and queues the frame with submitFrame() :261 -- the host never waits for the frame it just
 * recorded).
 * pt_tonemap_begin enqueues pt_tonemap_zoom's work behind the frames rendered so far and returns; frames rendered afterwards overlap it (their
 * accumulate step waits until the pass has read the accumulation image).  pt_tonemap_end blocks until the OLDEST image begun is in host memory
 * and copies it to rgba8_out (the size given to its pt_tonemap_begin).  At most PT_DISPLAY_RING (8) images may be pending
 * (PT_ERR_STATE beyond, and for pt_tonemap_end with none pending); pt_tonemap_pending returns how many are.  The images are bit-identical to
 * what pt_tonemap_zoom returns at the same point of the frame sequence.  A traversal-stack overflow is reported by the next synchronising
 * call (pt_synchronize, pt_tonemap, pt_read_accum), not by pt_tonemap_end. */
#define PT_DISPLAY_RING 8
int pt_tonemap_begin(pt_context* ctx, const pt_Tonemapper* tm, int disp_width, int disp_height);
int pt_tonemap_end(pt_context* ctx, uint8_t* rgba8_out);
int pt_tonemap_pending(pt_context* ctx);

/* Device-side view of the local shard for the RCCL gather: pointer to [maxTilesPerRank][PT_TILE*PT_TILE][4]
 * floats (owned tiles first, in increasing global tile id; padding zero). */
int pt_local_shard(pt_context* ctx, void** device_ptr, size_t* bytes, int* num_local_tiles, int* max_tiles_per_rank);

/* Rank 0 after the gather: gathered_dev holds nranks consecutive shards as returned by
 * pt_local_shard on each rank; places every tile at its global position so that pt_read_accum /
 * pt_tonemap see the full image. */
int pt_scatter_shards(pt_context* ctx, const void* gathered_dev, int nranks);

/* The one collective of the path, natively over RCCL / xGMI (no reference counterpart): every rank's shard goes to `root`
 * in one grouped operation (nranks-1 ncclRecv on the root, one ncclSend per peer: each shard on its own point-to-point link), then the root
 * places the tiles.  librccl.so is opened on first use.
 *   one process per GPU : rank 0 calls pt_comm_get_unique_id and distributes the 128 bytes; every rank pt_comm_init_rank; after rendering
 *                         every rank pt_gather_shards(ctx, comm, 0); rank 0 pt_gather_finish(ctx), then pt_read_accum / pt_tonemap.
 *   one process, N GPUs : pt_comm_init_all; pt_comm_group_begin(); pt_gather_shards(ctx[i], comm[i], 0) for every i; pt_comm_group_end();
 *                         pt_gather_finish(ctx[0]).
 * The communicator's rank / size must equal pt_set_shard's.  Without librccl.so every call of this group returns PT_ERR_UNAVAILABLE
 * (single-GPU hosts never need the library; libptmi.so is built without the RCCL headers).  Only the root allocates the gather buffer;
 * pt_gather_finish on a context that did not enqueue a gather as root returns PT_ERR_STATE. */
typedef struct pt_comm pt_comm;
#define PT_COMM_ID_BYTES 128
int pt_comm_get_unique_id(unsigned char id_out[PT_COMM_ID_BYTES]);
int pt_comm_init_rank(int nranks, const unsigned char id[PT_COMM_ID_BYTES], int rank, int device_ordinal, pt_comm** out_comm);
int pt_comm_init_all(int ndev, const int* device_ordinals, pt_comm** out_comms);
int pt_comm_destroy(pt_comm* comm);
/* ranks of the communicator as RCCL reports them (ncclCommCount): what a launcher prints to show that N processes really met */
int pt_comm_count(pt_comm* comm, int* out_nranks);
/* ncclGetVersion of the RCCL that was opened (e.g. 22707), 0 when the library does not say; a preflight of a multi-GPU run prints it */
int pt_comm_version(int* out_version);
/* why the last pt_comm_* call of this thread's process failed (e.g. "librccl.so not found: ..."); never NULL */
const char* pt_comm_last_error(void);
int pt_comm_group_begin(void);
int pt_comm_group_end(void);
int pt_gather_shards(pt_context* ctx, pt_comm* comm, int root);
int pt_gather_finish(pt_context* ctx);

/* ---- measurement ----------------------------------------------------------------------------- */

/* enable != 0: bracket every kernel with HIP events on the render stream and accumulate per-stage
 * milliseconds into pt_Stats (replaces the nvvk::ProfilerVK sections "Render"/"Tonemap",
 * src/sample_example.cpp:404, src/main.cpp:212-257). */
int pt_set_profiling(pt_context* ctx, int enable);
int pt_get_stats(pt_context* ctx, pt_Stats* out);
int pt_reset_stats(pt_context* ctx);

/* ---- host-side helpers restating the reference's scene packing -------------------------------- */

/* shaders/compress.glsl:111-139 (host flavour used at src/scene.cpp:224-225) */
uint32_t pt_compress_unit_vec(const float v[3]);

/* src/scene.cpp:219-242: positions[3n], normals[3n], tangents[4n] (w = handedness), uvs[2n],
 * colors[4n] (float 0..1) -> VertexAttributes[n] */
int pt_pack_vertices(uint32_t n, const float* positions, const float* normals, const float* tangents,
                     const float* uvs, const float* colors, pt_VertexAttributes* out);

/* src/scene.cpp:629-640 with glm::lookAt / perspectiveRH_ZO(fov, aspect, 0.001, 1e5), proj[1][1] *= -1 */
int pt_camera_lookat(const float eye[3], const float center[3], const float up[3], float fov_degrees,
                     float aspect, pt_SceneCamera* out);

/* src/hdr_sampling.cpp:107-248 on the host; pt_set_env calls this internally. */
int pt_build_env_accel(const float* rgba32f, int width, int height, pt_EnvAccel* out_accel, float* out_integral,
                       float* out_average);

/* src/scene.cpp:447-482,561-571: glTF sampler codes -> filter / wrap enums of pt_TextureDesc
 * (has_sampler == 0: LINEAR / REPEAT; unknown filter code: NEAREST; unknown wrap: REPEAT). */
int pt_sampler_from_gltf(int has_sampler, int gltf_mag, int gltf_min, int gltf_wrapS, int gltf_wrapT, pt_TextureDesc* io);

/* replaces the ray picker of SampleExample::screenPicking [src/sample_example.cpp:468-511 -> nvvk::RayPickerKHR]: shoots one ray through
 * the normalised window position (pick_x, pick_y in [0,1], origin top-left like the reference's cursor position) with the given inverse view
 * and inverse projection matrices (column-major, as in pt_SceneCamera) and returns the nearest triangle -- every triangle counts, without
 * face culling or alpha test, like the picker's flag-less traceRayEXT.  Synchronous. */
int pt_pick(pt_context* ctx, float pick_x, float pick_y, const float view_inverse[16], const float proj_inverse[16], pt_PickResult* out);

/* Measures, on this device, the two ceilings the measurement contract prices kernels against (no reference counterpart): VALU issue
 * (independent wave64 v_fma_f32 at 8 waves per SIMD on every CU, wave-instructions per second) and HBM streaming (float4 copy and
 * read-only over 1 GiB buffers, bytes per second).  Synchronous, a few tens of milliseconds. */
int pt_measure_peaks(pt_context* ctx, pt_Peaks* out);

/* Evaluates one function of the fp32 transcendental contract (include/pt_fpmath.h; enum PT_FN_* in pt_types.h) on the device for n
 * arguments (b is the second argument of atan2(a, b) / pow(a, b), otherwise ignored and may be NULL).  No reference counterpart: GLSL
 * leaves these functions' accuracy to the driver; the contract fixes one IEEE operation sequence and this entry point lets a test hold
 * the gfx950 evaluation bit for bit to the host evaluation of the same header.  Synchronous. */
int pt_fpmath_eval(pt_context* ctx, int fn, uint64_t n, const float* a, const float* b, float* out);

/* ---- glTF import (host only, no GPU) ------------------------------------------------------------------------------------
 * replaces Scene::load -> loadGltfScene (tinygltf) + nvh::GltfScene::importMaterials / importDrawableNodes + the create*Buffer
 * packing [src/scene.cpp:56-155, 190-382, 488-580] for .gltf and .glb files: the result is the flat pt_SceneDesc pt_set_scene
 * takes (textures are (sampler, image) pairs decoded to RGBA8; PNG and Huffman-coded JPEG, sequential or progressive), plus the first camera of the file or a
 * fit to the bounding box [src/scene.cpp:281-298].  The scene owns every array the description points to until pt_gltf_free.
 * On failure returns PT_ERR_INVALID and writes a message to `err` (may be NULL). */
typedef struct pt_GltfScene pt_GltfScene;
int                 pt_gltf_load(const char* path, pt_GltfScene** out_scene, char* err, size_t err_len);
const pt_SceneDesc* pt_gltf_desc(const pt_GltfScene* scene);
int                 pt_gltf_camera(const pt_GltfScene* scene, float eye[3], float center[3], float up[3], float* fov_degrees);
int                 pt_gltf_bounds(const pt_GltfScene* scene, float bbox_min[3], float bbox_max[3]);
void                pt_gltf_free(pt_GltfScene* scene);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif

#endif /* PT_API_H */
