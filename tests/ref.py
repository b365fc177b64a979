"""ctypes binding of oracle/_ref/libref.so -- the reference's OWN shader sources (shaders/pathtrace.comp and everything it includes,
shaders/post.frag) and host sources (src/hdr_sampling.cpp, the host branch of shaders/compress.glsl) compiled as C++ by the committed
recipe oracle/ref_glue/ (see its README).  Test infrastructure only: it pins the oracle, nothing else may load it.

The library exists only where /root/reference exists (this container); the GPU box receives the prebuilt file with the snapshot.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from tests import orc
from vk_raytrace_amd import host_device as hd

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_ROOT, "oracle", "_ref", "libref.so")
REFERENCE = os.environ.get("PT_REFERENCE_DIR", "/root/reference")


def available():
    return os.path.exists(LIB_PATH) or os.path.isdir(os.path.join(REFERENCE, "shaders"))


def build():
    if os.path.isdir(os.path.join(REFERENCE, "shaders")):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s", "ref", f"REF={REFERENCE}"])
    return LIB_PATH


class Hooks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("query", C.c_void_p), ("tri_info", C.c_void_p), ("instance", C.c_void_p), ("sample_texture", C.c_void_p),
                ("sample_env", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        P = C.c_void_p
        L.ref_bind.argtypes = [C.POINTER(hd.SceneDesc), P, C.c_int, C.c_int, C.POINTER(Hooks)]
        L.ref_set_camera.argtypes = [C.POINTER(hd.SceneCamera)]
        L.ref_set_sunsky.argtypes = [C.POINTER(hd.SunAndSky)]
        L.ref_render_frame.argtypes = [C.POINTER(hd.RtxState), P, P, C.c_uint64, C.c_int]
        L.ref_render_frames.argtypes = [C.POINTER(hd.RtxState), C.c_int, C.c_int, P, P, C.c_uint64, C.c_int]
        L.ref_tea.argtypes = [C.c_uint32, C.c_uint32]
        L.ref_tea.restype = C.c_uint32
        L.ref_pcg_stream.argtypes = [C.c_uint32, C.c_uint32, P, P, P]
        L.ref_pcg3d.argtypes = [P]
        L.ref_compress_unit_vec.argtypes = [P]
        L.ref_compress_unit_vec.restype = C.c_uint32
        L.ref_decompress_unit_vec.argtypes = [C.c_uint32, P]
        L.ref_offset_ray.argtypes = [P, P, P]
        L.ref_spherical_uv.argtypes = [P, P]
        L.ref_coordinate_system.argtypes = [P, P, P]
        L.ref_temperature.argtypes = [C.c_float, P]
        L.ref_sun_and_sky.argtypes = [C.POINTER(hd.SunAndSky), P, P]
        L.ref_range_attenuation.argtypes = [C.c_float, C.c_float]
        L.ref_range_attenuation.restype = C.c_float
        L.ref_spot_attenuation.argtypes = [P, P, C.c_float, C.c_float]
        L.ref_spot_attenuation.restype = C.c_float
        L.ref_bsdf_eval.argtypes = [C.c_int, P, P, P, P, C.c_float, C.c_int, P, P, P, P]
        L.ref_bsdf_sample.argtypes = [C.c_int, P, P, P, P, C.c_float, C.c_int, P, P, P, P, P]
        L.ref_rtx_bind.argtypes = [C.POINTER(hd.SceneDesc), P, C.c_int, C.c_int, C.POINTER(Hooks)]
        L.ref_rtx_set_camera.argtypes = [C.POINTER(hd.SceneCamera)]
        L.ref_rtx_set_sunsky.argtypes = [C.POINTER(hd.SunAndSky)]
        L.ref_rtx_use_any_hit.argtypes = [C.c_int]
        L.ref_rtx_render_frame.argtypes = [C.POINTER(hd.RtxState), P, P, C.c_uint64, C.c_int]
        L.ref_host_compress_unit_vec.argtypes = [P]
        L.ref_host_compress_unit_vec.restype = C.c_uint32
        L.ref_host_pack_unorm4x8.argtypes = [P]
        L.ref_host_pack_unorm4x8.restype = C.c_uint32
        L.ref_host_round_even.argtypes = [C.c_float]
        L.ref_host_round_even.restype = C.c_float
        L.ref_env_accel.argtypes = [P, C.c_int, C.c_int, P, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ref_mip_chain.argtypes = [P, C.c_int, C.c_int, C.c_int, P, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_tonemap.argtypes = [C.POINTER(hd.Tonemapper), P, C.c_int, C.c_int, P]
        _lib = L
    return _lib


def _fn_addr(cdll, name):
    return C.cast(getattr(cdll, name), C.c_void_p).value


class Reference:
    """pathtrace.comp dispatched on the CPU.  What the Vulkan driver would supply (triangle candidates in the order of the trace contract,
    instance matrices, bilinear sampling) is bound to an Oracle instance holding the same scene; everything else is the reference's code."""

    def __init__(self, scene, env, oracle=None, rtx=False, any_hit=True):
        """rtx: the RtxPipeline flavour (pathtrace.rgen + .rchit / .rahit / .rmiss through an emulated vkCmdTraceRaysKHR) instead of the
        ray-query compute shader; any_hit: RtxPipeline::useAnyHit (the hooks' oracle must be in the same mode: opaque flags come from it)"""
        self.L = lib()
        self.rtx = rtx
        self.o = oracle or orc.Oracle()
        self.own = oracle is None
        if self.own:
            self.o.use_any_hit(any_hit)
            self.o.set_scene(scene)
            self.o.set_env(env)
        OL = self.o.L
        OL.orc_env_accel.restype = C.c_void_p
        OL.orc_env_accel.argtypes = [C.c_void_p]
        self.desc, self.keep = scene.desc()
        env = np.ascontiguousarray(env, np.float32)
        self.hooks = Hooks(self.o.ctx, _fn_addr(OL, "orc_hook_query"), _fn_addr(OL, "orc_hook_tri_info"), _fn_addr(OL, "orc_hook_instance"),
                           _fn_addr(OL, "orc_hook_sample_texture"), _fn_addr(OL, "orc_hook_sample_env"))
        (self.L.ref_rtx_bind if rtx else self.L.ref_bind)(C.byref(self.desc), OL.orc_env_accel(self.o.ctx), env.shape[1], env.shape[0], C.byref(self.hooks))
        if rtx:
            self.L.ref_rtx_use_any_hit(int(bool(any_hit)))

    def set_camera(self, cam):
        (self.L.ref_rtx_set_camera if self.rtx else self.L.ref_set_camera)(C.byref(cam))

    def set_sunsky(self, ss):
        (self.L.ref_rtx_set_sunsky if self.rtx else self.L.ref_set_sunsky)(C.byref(ss))

    def render(self, state, frames, accum=None, first_frame=0, pixel_ids=None, threads=0):
        W, H = state.size[0], state.size[1]
        if accum is None:
            accum = np.zeros((H, W, 4), np.float32)
        ids = None if pixel_ids is None else np.ascontiguousarray(pixel_ids, np.uint32)
        for f in range(first_frame, first_frame + frames):
            state.frame = f
            (self.L.ref_rtx_render_frame if self.rtx else self.L.ref_render_frame)(C.byref(state), accum.ctypes.data, None if ids is None else ids.ctypes.data,
                                                                                    0 if ids is None else len(ids), threads)
        return accum


    def render_frames(self, state, first_frame, nframes, accum, pixel_ids, threads=0):
        """pathtrace.comp for `nframes` consecutive frames of the listed pixels inside one OpenMP team (bench.py's cpu_baseline leg)."""
        assert not self.rtx
        ids = np.ascontiguousarray(pixel_ids, np.uint32)
        self.L.ref_render_frames(C.byref(state), int(first_frame), int(nframes), accum.ctypes.data, ids.ctypes.data, len(ids), threads)
        return accum


def render_reference(cfg, frames, pixel_ids=None):
    """Same call shape as tests.common.render_oracle."""
    r = Reference(cfg.scene, cfg.env, rtx=cfg.variant == 1, any_hit=cfg.any_hit)
    r.set_camera(cfg.camera)
    r.set_sunsky(cfg.sunsky)
    return r.render(cfg.state(r.o.integral), frames, pixel_ids=pixel_ids)
