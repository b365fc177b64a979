"""ctypes binding of libptmi.so (include/pt_api.h).

The library is built in-tree by ``vk_raytrace_amd/csrc/Makefile`` (hipcc, gfx950 only).  There is
no fallback of any kind: if the shared object is missing this module raises at import of the
symbols, and without a gfx950 device ``pt_create`` fails with PT_ERR_NO_DEVICE.
"""
import ctypes as C
import os

import numpy as np

from . import host_device as hd

PT_VARIANT_RAYQUERY, PT_VARIANT_RTX = 0, 1
PT_ACCEL_FLAT, PT_ACCEL_TWO_LEVEL = 0, 1
PT_DISPLAY_RING = 8  # images pt_tonemap_begin may have pending (include/pt_api.h)
PT_FN = {"sin": 0, "cos": 1, "tan": 2, "asin": 3, "acos": 4, "atan2": 5, "exp": 6, "log": 7, "pow": 8}
PT_OK, PT_ERR_INVALID, PT_ERR_NO_DEVICE, PT_ERR_HIP, PT_ERR_STATE, PT_ERR_OOM, PT_ERR_UNAVAILABLE = 0, -1, -2, -3, -4, -5, -6

# PT_LIB: developer override used to compare builds of the same HIP library (tools/build_variants.sh); never a fallback
LIB_PATH = os.environ.get("PT_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libptmi.so")

# every symbol include/pt_api.h declares: (name, restype, argtypes)
_P = C.c_void_p
API = [
    ("pt_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("pt_destroy", C.c_int, [_P]),
    ("pt_renderer_name", C.c_char_p, []),
    ("pt_last_error", C.c_char_p, [_P]),
    ("pt_set_scene", C.c_int, [_P, C.POINTER(hd.SceneDesc)]),
    ("pt_build_accel", C.c_int, [_P]),
    ("pt_set_accel_mode", C.c_int, [_P, C.c_int]),
    ("pt_update_instances", C.c_int, [_P, _P, C.c_uint32]),
    ("pt_set_camera", C.c_int, [_P, C.POINTER(hd.SceneCamera)]),
    ("pt_set_env", C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("pt_hdr_load", C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    ("pt_hdr_free", None, [C.POINTER(C.c_float)]),
    ("pt_set_sunsky", C.c_int, [_P, C.POINTER(hd.SunAndSky)]),
    ("pt_resize", C.c_int, [_P, C.c_int, C.c_int]),
    ("pt_set_shard", C.c_int, [_P, C.c_int, C.c_int]),
    ("pt_set_variant", C.c_int, [_P, C.c_int]),
    ("pt_use_any_hit", C.c_int, [_P, C.c_int]),
    ("pt_render_frame", C.c_int, [_P, C.POINTER(hd.RtxState)]),
    ("pt_synchronize", C.c_int, [_P]),
    ("pt_read_accum", C.c_int, [_P, _P]),
    ("pt_write_accum", C.c_int, [_P, _P]),
    ("pt_tonemap", C.c_int, [_P, C.POINTER(hd.Tonemapper), _P]),
    ("pt_tonemap_zoom", C.c_int, [_P, C.POINTER(hd.Tonemapper), C.c_int, C.c_int, _P]),
    ("pt_tonemap_begin", C.c_int, [_P, C.POINTER(hd.Tonemapper), C.c_int, C.c_int]),
    ("pt_tonemap_end", C.c_int, [_P, _P]),
    ("pt_tonemap_pending", C.c_int, [_P]),
    ("pt_local_shard", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("pt_scatter_shards", C.c_int, [_P, _P, C.c_int]),
    ("pt_comm_get_unique_id", C.c_int, [_P]),
    ("pt_comm_init_rank", C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    ("pt_comm_init_all", C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    ("pt_comm_destroy", C.c_int, [_P]),
    ("pt_comm_count", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("pt_comm_version", C.c_int, [C.POINTER(C.c_int)]),
    ("pt_comm_last_error", C.c_char_p, []),
    ("pt_comm_group_begin", C.c_int, []),
    ("pt_comm_group_end", C.c_int, []),
    ("pt_gather_shards", C.c_int, [_P, _P, C.c_int]),
    ("pt_gather_finish", C.c_int, [_P]),
    ("pt_pick", C.c_int, [_P, C.c_float, C.c_float, _P, _P, C.POINTER(hd.PickResult)]),
    ("pt_fpmath_eval", C.c_int, [_P, C.c_int, C.c_uint64, _P, _P, _P]),
    ("pt_measure_peaks", C.c_int, [_P, C.POINTER(hd.Peaks)]),
    ("pt_set_profiling", C.c_int, [_P, C.c_int]),
    ("pt_get_stats", C.c_int, [_P, C.POINTER(hd.Stats)]),
    ("pt_reset_stats", C.c_int, [_P]),
    ("pt_compress_unit_vec", C.c_uint32, [_P]),
    ("pt_pack_vertices", C.c_int, [C.c_uint32, _P, _P, _P, _P, _P, _P]),
    ("pt_camera_lookat", C.c_int, [_P, _P, _P, C.c_float, C.c_float, C.POINTER(hd.SceneCamera)]),
    ("pt_build_env_accel", C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("pt_sampler_from_gltf", C.c_int, [C.c_int] * 5 + [C.POINTER(hd.TextureDesc)]),
    ("pt_gltf_load", C.c_int, [C.c_char_p, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    ("pt_gltf_desc", C.POINTER(hd.SceneDesc), [_P]),
    ("pt_gltf_camera", C.c_int, [_P, _P, _P, _P, C.POINTER(C.c_float)]),
    ("pt_gltf_bounds", C.c_int, [_P, _P, _P]),
    ("pt_gltf_free", None, [_P]),
]

_lib = None


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libptmi error {code}: {msg}")
        self.code = code


def lib():
    """Loads libptmi.so; raises if it has not been built (there is no Python/CPU fallback)."""
    global _lib
    if _lib is None:
        # libptmi overlaps several frames on separate HIP streams; HIP maps streams onto 4 hardware queues unless told
        # otherwise, which serialises them.  Must be set before the HIP runtime initialises.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C vk_raytrace_amd/csrc` "
                              "(or __graft_entry__.build()); the HIP library is the only implementation")
        # RTLD_DEEPBIND: libptmi.so resolves hip* against ITS OWN dependency (/opt/rocm's libamdhip64), whatever else the process has loaded.
        # PyTorch-ROCm wheels bundle a second HIP / HSA runtime; without deep binding the import order decides which runtime libptmi talks to,
        # and two initialised runtimes in one process do not coexist ("No HIP GPUs are available", RCCL "unhandled cuda error").  With it the
        # library, and the RCCL it opens the same way (csrc/pt_comm.cpp), always share one runtime; a process that uses libptmi must simply not
        # initialise torch.cuda (bench.py's ranks use torch.distributed over gloo for the control plane only).
        L = C.CDLL(LIB_PATH, mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND)
        for name, res, args in API:
            fn = getattr(L, name)  # AttributeError if the ABI and the header ever disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def pack_vertices(pos, nrm, tan, uv, col):
    """pt_pack_vertices (reference: src/scene.cpp:219-242)."""
    n = len(pos)
    out = np.zeros(n, hd.vertex_dtype)
    a = [np.ascontiguousarray(x, np.float32) for x in (pos, nrm, tan, uv, col)]
    rc = lib().pt_pack_vertices(n, *[x.ctypes.data for x in a], out.ctypes.data)
    if rc != PT_OK:
        raise PtError(rc, "pt_pack_vertices")
    return out


def camera_lookat(cam, aspect, nb_lights=0):
    """pt_camera_lookat (reference: src/scene.cpp:629-640) from a scene.Camera."""
    out = hd.SceneCamera()
    e, c, u = (np.asarray(v, np.float32) for v in (cam.eye, cam.center, cam.up))
    rc = lib().pt_camera_lookat(e.ctypes.data, c.ctypes.data, u.ctypes.data, cam.fov, aspect, C.byref(out))
    if rc != PT_OK:
        raise PtError(rc, "pt_camera_lookat")
    out.aperture = cam.aperture
    if cam.focal_dist is not None:
        out.focalDist = cam.focal_dist
    out.nbLights = nb_lights
    return out


def build_env_accel(env):
    env = np.ascontiguousarray(env, np.float32)
    h, w = env.shape[:2]
    acc = np.zeros(w * h, hd.envaccel_dtype)
    i, a = C.c_float(), C.c_float()
    rc = lib().pt_build_env_accel(env.ctypes.data, w, h, acc.ctypes.data, C.byref(i), C.byref(a))
    if rc != PT_OK:
        raise PtError(rc, "pt_build_env_accel")
    return acc, i.value, a.value


def load_hdr(path):
    """pt_hdr_load: a Radiance .hdr file as an (h, w, 4) float32 array (reference: stbi_loadf in src/hdr_sampling.cpp:64)."""
    p, w, h = C.POINTER(C.c_float)(), C.c_int(), C.c_int()
    err = C.create_string_buffer(256)
    rc = lib().pt_hdr_load(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h), err, 256)
    if rc != PT_OK:
        raise PtError(rc, err.value.decode())
    try:
        return np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
    finally:
        lib().pt_hdr_free(p)
