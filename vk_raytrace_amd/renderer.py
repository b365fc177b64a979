"""Host-side mirror of the reference's renderer interface for the gfx950 backend.

``Renderer`` has the reference's method set and call order (reference: src/renderer.h:30-48);
``HipRenderer`` is the implementation that takes the place of ``RayQuery`` (src/rayquery.cpp:40-109)
and ``RtxPipeline`` (src/rtx_pipeline.cpp:45-276).  ``SampleExample`` is the headless part of the
reference's orchestrator that the hot path needs (src/sample_example.cpp:77-82 setup loop,
:90-111 load, :183-207 frame counter, :322-337 createRender, :390-429 renderScene, :362-384 drawPost).

Vulkan handles do not exist on the target: what reached the shaders through descriptor sets
(TLAS, output image, scene buffers, environment) is handed over with explicit ``set_*`` calls, and
the command buffer is the context's own HIP stream.
"""
import ctypes as C
import os

import numpy as np

from . import capi
from . import host_device as hd


class Renderer:
    """reference: src/renderer.h:30-48"""

    def __init__(self):
        self.m_state = hd.RtxState()

    def setup(self, device, physicalDevice=None, familyIndex=0, allocator=None):
        raise NotImplementedError

    def destroy(self):
        raise NotImplementedError

    def run(self, cmdBuf, size, profiler, extraDescSets):
        raise NotImplementedError

    def create(self, size, extraDescSetsLayout, scene=None):
        raise NotImplementedError

    def name(self):
        raise NotImplementedError

    def setPushContants(self, state):  # (sic) the reference's spelling, src/renderer.h:44
        self.m_state = state


class HipRenderer(Renderer):
    """libptmi.so behind the reference's Renderer interface."""

    def __init__(self):
        super().__init__()
        self._lib = capi.lib()
        self._ctx = None
        self._display_sizes = []  # viewports of the images begun with tonemap_begin and not collected yet
        self._keep = None
        self.size = (0, 0)
        self.env_integral = 1.0
        self.env_average = 1.0

    # -- Renderer interface ------------------------------------------------------------------------
    def setup(self, device=0, physicalDevice=None, familyIndex=0, allocator=None):
        """Renderer::setup: bind to one GPU (`device` is the HIP ordinal)."""
        if self._ctx is not None:
            return
        self._display_sizes = []  # a new context has no display image pending
        ctx = C.c_void_p()
        rc = self._lib.pt_create(int(device), C.byref(ctx))
        if rc != capi.PT_OK:
            raise capi.PtError(rc, self._lib.pt_last_error(None).decode())
        self._ctx = ctx

    def destroy(self):
        if self._ctx is not None:
            self._lib.pt_destroy(self._ctx)
            self._ctx = None
        self._display_sizes = []  # the C side's display ring went with the context

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def name(self):
        return self._lib.pt_renderer_name().decode()

    def create(self, size, extraDescSetsLayout=None, scene=None):
        """Renderer::create(size, layouts, scene): allocate the output for `size` = (width, height);
        if a scene is given it is uploaded and its acceleration structure built."""
        if scene is not None:
            self.set_scene(scene)
        self._check(self._lib.pt_resize(self._ctx, int(size[0]), int(size[1])))
        self.size = (int(size[0]), int(size[1]))

    def run(self, cmdBuf=None, size=None, profiler=None, extraDescSets=None):
        """Renderer::run: one frame with the state given to setPushContants (asynchronous)."""
        if size is not None and tuple(size) != self.size:
            self.create(size)
        self.m_state.size[0], self.m_state.size[1] = self.size
        self._check(self._lib.pt_render_frame(self._ctx, C.byref(self.m_state)))

    # -- what the descriptor sets carried -------------------------------------------------------------
    def set_scene(self, scene):
        d, keep = scene.desc()
        self._check(self._lib.pt_set_scene(self._ctx, C.byref(d)))
        self._check(self._lib.pt_build_accel(self._ctx))
        self._keep = keep

    def set_env(self, env_rgba32f):
        env = np.ascontiguousarray(env_rgba32f, np.float32)
        assert env.ndim == 3 and env.shape[2] == 4
        i, a = C.c_float(), C.c_float()
        self._check(self._lib.pt_set_env(self._ctx, env.ctypes.data, env.shape[1], env.shape[0], C.byref(i), C.byref(a)))
        self.env_integral, self.env_average = i.value, a.value
        return self.env_integral, self.env_average

    def set_camera(self, cam: hd.SceneCamera):
        self._check(self._lib.pt_set_camera(self._ctx, C.byref(cam)))

    def set_sunsky(self, ss: hd.SunAndSky):
        self._check(self._lib.pt_set_sunsky(self._ctx, C.byref(ss)))

    def useAnyHit(self, enable):
        """RtxPipeline::useAnyHit (src/rtx_pipeline.cpp:269-276)"""
        self._check(self._lib.pt_use_any_hit(self._ctx, int(bool(enable))))

    def set_accel_mode(self, mode):
        """capi.PT_ACCEL_FLAT (default: one hierarchy over all world-space triangles) or capi.PT_ACCEL_TWO_LEVEL (the reference's
        BLAS per prim-mesh + TLAS instance per node, src/accelstruct.cpp:110-162)."""
        self._check(self._lib.pt_set_accel_mode(self._ctx, int(mode)))

    def update_instances(self, nodes):
        """New world matrices for the scene's nodes (hd.node_dtype array or scene.Scene): TLAS refit in two-level mode, rebuild otherwise."""
        arr = np.ascontiguousarray(nodes.node_array() if hasattr(nodes, "node_array") else nodes, hd.node_dtype)
        self._check(self._lib.pt_update_instances(self._ctx, arr.ctypes.data, len(arr)))

    def set_variant(self, variant):
        """capi.PT_VARIANT_RAYQUERY (the reference's RayQuery renderer, default) or capi.PT_VARIANT_RTX (its RtxPipeline)."""
        self._check(self._lib.pt_set_variant(self._ctx, int(variant)))

    def set_shard(self, rank, nranks):
        self._check(self._lib.pt_set_shard(self._ctx, rank, nranks))

    # -- output --------------------------------------------------------------------------------------
    def synchronize(self):
        self._check(self._lib.pt_synchronize(self._ctx))

    def read_accum(self):
        w, h = self.size
        out = np.empty((h, w, 4), np.float32)
        self._check(self._lib.pt_read_accum(self._ctx, out.ctypes.data))
        return out

    def write_accum(self, img):
        """Checkpoint restore: replaces the accumulation image (continue with RtxState.frame = frames already in it)."""
        w, h = self.size
        img = np.ascontiguousarray(img, np.float32)
        assert img.shape == (h, w, 4)
        self._check(self._lib.pt_write_accum(self._ctx, img.ctypes.data))

    def pick(self, x, y, cam: hd.SceneCamera):
        """SampleExample::screenPicking (src/sample_example.cpp:468-511): nearest triangle under the normalised window position."""
        out = hd.PickResult()
        vi = (C.c_float * 16)(*cam.viewInverse)
        pi = (C.c_float * 16)(*cam.projInverse)
        self._check(self._lib.pt_pick(self._ctx, float(x), float(y), vi, pi, C.byref(out)))
        return out

    def tonemap(self, tm: hd.Tonemapper, display_size=None):
        """RenderOutput::genMipmap + run.  display_size (W, H): the viewport while de-scaling (Tonemapper.zoom = 1 / level)."""
        w, h = display_size or self.size
        out = np.empty((h, w, 4), np.uint8)
        if display_size is None:
            self._check(self._lib.pt_tonemap(self._ctx, C.byref(tm), out.ctypes.data))
        else:
            self._check(self._lib.pt_tonemap_zoom(self._ctx, C.byref(tm), w, h, out.ctypes.data))
        return out

    def tonemap_begin(self, tm: hd.Tonemapper, display_size=None):
        """pt_tonemap_begin: the display pass enqueued behind the frames rendered so far; later frames overlap it.  tonemap_end() returns the
        oldest image begun (at most 4 may be pending)."""
        w, h = display_size or self.size
        self._check(self._lib.pt_tonemap_begin(self._ctx, C.byref(tm), w, h))  # raises before the shadow list grows: the two stay in step
        self._display_sizes.append((w, h))

    def tonemap_pending(self):
        return int(self._lib.pt_tonemap_pending(self._ctx))

    def tonemap_end(self):
        if not self._display_sizes:
            self._check(self._lib.pt_tonemap_end(self._ctx, np.empty(4, np.uint8).ctypes.data))  # reports the error
        if self.tonemap_pending() != len(self._display_sizes):  # the C side is the truth; never size a buffer from a stale shadow entry
            self._display_sizes = []
            raise capi.PtError(capi.PT_ERR_STATE, "tonemap_end: the display ring and its Python shadow disagree (context re-created?)")
        w, h = self._display_sizes[0]
        out = np.empty((h, w, 4), np.uint8)
        self._check(self._lib.pt_tonemap_end(self._ctx, out.ctypes.data))
        self._display_sizes.pop(0)
        return out

    def measure_peaks(self):
        """pt_measure_peaks: VALU issue and HBM streaming ceilings measured on this device (dict)"""
        p = hd.Peaks()
        self._check(self._lib.pt_measure_peaks(self._ctx, C.byref(p)))
        return {k: getattr(p, k) for k, _ in hd.Peaks._fields_}

    def local_shard(self):
        ptr, nbytes, nloc, nmax = C.c_void_p(), C.c_size_t(), C.c_int(), C.c_int()
        self._check(self._lib.pt_local_shard(self._ctx, C.byref(ptr), C.byref(nbytes), C.byref(nloc), C.byref(nmax)))
        return ptr.value, nbytes.value, nloc.value, nmax.value

    def scatter_shards(self, gathered_device_ptr, nranks):
        self._check(self._lib.pt_scatter_shards(self._ctx, C.c_void_p(gathered_device_ptr), nranks))

    # -- measurement -----------------------------------------------------------------------------------
    def set_profiling(self, enable):
        self._check(self._lib.pt_set_profiling(self._ctx, int(enable)))

    def stats(self):
        s = hd.Stats()
        self._check(self._lib.pt_get_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._check(self._lib.pt_reset_stats(self._ctx))

    def _check(self, rc):
        if rc != capi.PT_OK:
            raise capi.PtError(rc, self._lib.pt_last_error(self._ctx).decode())


class SampleExample:
    """The headless slice of the reference's orchestrator (src/sample_example.{hpp,cpp}): scene / environment loading, the frame counter with its
    camera-change detection, the de-scaling state machine of the mouse callbacks, renderScene and drawPost."""

    def __init__(self, device=0, renderer=None):
        self.m_rtxState = hd.default_rtx_state()       # sample_example.hpp:162-174
        self.m_sunAndSky = hd.default_sun_and_sky()    # sample_example.hpp:176-193
        self.m_tonemapper = hd.default_tonemapper()    # render_output.hpp:37-49
        self.m_maxFrames = 100000                      # sample_example.hpp:195
        self.m_descaling = False                       # sample_example.hpp:197
        self.m_descalingLevel = 1                      # sample_example.hpp:198
        self.m_framesInFlight = 0                      # display images the host lets the GPU run behind (drawPost); the reference's swapchain ring
        self.m_pRender = renderer if renderer is not None else HipRenderer()
        self.m_pRender.setup(device)                   # sample_example.cpp:77-82
        self.m_scene = None
        self.m_size = (0, 0)                           # m_renderRegion.extent
        self._refCamMatrix = None                      # updateFrame's `static glm::mat4 refCamMatrix` / `static float fov`
        self._refFov = 0.0
        self._inputs = {"lmb": False, "mmb": False, "rmb": False}  # AppBaseVk::m_inputs
        self.resetFrame()

    # sample_example.cpp:90-98 loadScene + main.cpp:186-189.  A path goes through libptmi's own importer (pt_gltf_load), like Scene::load.
    def loadScene(self, scene):
        if isinstance(scene, (str, bytes, os.PathLike)):
            from .scene import GltfFileScene
            scene = GltfFileScene(os.fsdecode(scene))
        if scene.vertices is None:
            scene.finalize(capi.pack_vertices)
        self.m_scene = scene
        self.m_pRender.set_scene(scene)
        self.resetFrame()

    # sample_example.cpp:103-111 (the image arrives decoded: stbi_loadf has no counterpart here)
    def loadEnvironmentHdr(self, env):
        """SampleExample::loadEnvironmentHdr (sample_example.cpp:102-112): `env` is the path of a Radiance .hdr file (decoded by
        pt_hdr_load, the stbi_loadf of hdr_sampling.cpp:64) or an (h, w, 4) float32 array."""
        env_rgba32f = capi.load_hdr(env) if isinstance(env, (str, bytes, os.PathLike)) else env
        integral, _ = self.m_pRender.set_env(env_rgba32f)
        self.m_rtxState.fireflyClampThreshold = integral * 4.0  # "magic", sample_example.cpp:110
        self.resetFrame()

    def setRenderRegion(self, width, height):
        self.m_size = (int(width), int(height))
        self.m_pRender.create(self.m_size)
        self.resetFrame()

    def _render_size(self):
        """sample_example.cpp:410-413: the size actually rendered (integer division by the de-scaling level while a mouse button is down)"""
        if self.m_descaling:
            return (max(1, self.m_size[0] // self.m_descalingLevel), max(1, self.m_size[1] // self.m_descalingLevel))
        return self.m_size

    # sample_example.cpp:168-178 (the aspect ratio is the render region's, whatever the de-scaling)
    def updateUniformBuffer(self):
        cam = capi.camera_lookat(self.m_scene.camera, self.m_size[0] / self.m_size[1], nb_lights=len(self.m_scene.lights))
        self.m_pRender.set_camera(cam)
        self.m_pRender.set_sunsky(self.m_sunAndSky)

    # sample_example.cpp:183-199: "If the camera matrix has changed, resets the frame otherwise, increments frame."  CameraManip.getMatrix() is
    # the view matrix of (eye, center, up); comparing the three vectors it is computed from is the same predicate.
    def updateFrame(self):
        c = self.m_scene.camera if self.m_scene is not None else None
        m = None if c is None else (tuple(float(x) for x in c.eye), tuple(float(x) for x in c.center), tuple(float(x) for x in c.up))
        f = 0.0 if c is None else float(c.fov)
        if self._refCamMatrix != m or self._refFov != f:
            self.resetFrame()
            self._refCamMatrix, self._refFov = m, f
        if self.m_rtxState.frame < self.m_maxFrames:
            self.m_rtxState.frame += 1

    def resetFrame(self):  # sample_example.cpp:204-207
        self.m_rtxState.frame = -1

    # sample_example.cpp:528-541 / :546-557 -- the de-scaling state machine of the window callbacks
    def onMouseMotion(self, x=0, y=0):
        if self._inputs["lmb"] or self._inputs["rmb"] or self._inputs["mmb"]:
            self.m_descaling = True

    def onMouseButton(self, button, pressed):
        """button: "lmb" | "mmb" | "rmb"; pressed: GLFW_PRESS (True) / GLFW_RELEASE (False)"""
        self._inputs[button] = bool(pressed)
        if not (self._inputs["lmb"] or self._inputs["rmb"] or self._inputs["mmb"]) and not pressed and self.m_descaling:
            self.m_descaling = False
            self.resetFrame()

    # sample_example.cpp:390-429
    def renderScene(self):
        if self.m_rtxState.frame >= self.m_maxFrames:
            return
        size = self._render_size()
        self.m_rtxState.size[0], self.m_rtxState.size[1] = size
        self.m_pRender.setPushContants(self.m_rtxState)
        self.m_pRender.run(None, size, None, None)

    # sample_example.cpp:362-384 -> RenderOutput::run; :378 `zoom = m_descaling ? 1.0f / m_descalingLevel : 1.0f`
    def drawPost(self):
        """The display pass (SampleExample::drawPost, src/sample_example.cpp:396-431).  m_framesInFlight = 0: the RGBA8 image of the frame just
        rendered (the host waits for it).  m_framesInFlight = k > 0: the pass is enqueued and the image of the frame rendered k calls ago is
        returned (None for the first k calls) -- the host never waits for the frame it just issued, like the reference's loop around
        prepareFrame / submitFrame (src/main.cpp:213,261); flushDisplay() returns what is still in flight."""
        size = None
        if self.m_descaling:
            self.m_tonemapper.zoom = 1.0 / self.m_descalingLevel
            size = self.m_size
        else:
            self.m_tonemapper.zoom = 1.0
        if self.m_framesInFlight <= 0:
            return self.m_pRender.tonemap(self.m_tonemapper, display_size=size)
        self.m_pRender.tonemap_begin(self.m_tonemapper, display_size=size)
        if self.m_pRender.tonemap_pending() > min(self.m_framesInFlight, capi.PT_DISPLAY_RING - 1):
            return self.m_pRender.tonemap_end()
        return None

    def flushDisplay(self):
        return [self.m_pRender.tonemap_end() for _ in range(self.m_pRender.tonemap_pending())]

    def render(self, frames):
        """`frames` iterations of the reference's main loop body (src/main.cpp:201-264)."""
        self.updateUniformBuffer()
        for _ in range(frames):
            self.updateFrame()
            self.renderScene()
        self.m_pRender.synchronize()
        return self.m_pRender.read_accum()

    def destroy(self):
        self.m_pRender.destroy()
