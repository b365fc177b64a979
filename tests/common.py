"""Shared helpers of the parity tests: one configuration rendered by the CPU oracle and by the HIP
path through the C ABI, on identical inputs."""
import ctypes as C

import numpy as np

from tests import orc
from vk_raytrace_amd import capi, host_device as hd, synth


class Config:
    def __init__(self, scene, env, width, height, depth=10, pbr=0, sunsky=None, debug=0, max_samples=1, hdr_multiplier=1.0, firefly=None, variant=0, any_hit=True):
        self.scene = scene
        if scene.vertices is None:
            scene.finalize(capi.pack_vertices)
        self.env = np.ascontiguousarray(env, np.float32)
        self.width, self.height = width, height
        self.depth, self.pbr, self.debug, self.max_samples = depth, pbr, debug, max_samples
        self.sunsky = sunsky if sunsky is not None else hd.default_sun_and_sky()
        self.hdr_multiplier = hdr_multiplier
        self.firefly = firefly
        self.variant = variant  # capi.PT_VARIANT_RAYQUERY / PT_VARIANT_RTX
        self.any_hit = any_hit  # RtxPipeline::useAnyHit
        self.camera = capi.camera_lookat(scene.camera, width / height, nb_lights=len(scene.lights))

    def state(self, integral):
        st = hd.default_rtx_state()
        st.size[0], st.size[1] = self.width, self.height
        st.maxDepth, st.pbrMode, st.debugging_mode, st.maxSamples = self.depth, self.pbr, self.debug, self.max_samples
        st.hdrMultiplier = self.hdr_multiplier
        st.fireflyClampThreshold = 4.0 * integral if self.firefly is None else self.firefly  # src/sample_example.cpp:110
        return st


def render_oracle(cfg, frames, use_bvh=True, threads=0, return_obj=False, math_mode=0):
    orc.set_math_mode(math_mode)
    o = orc.Oracle(threads)
    o.set_use_bvh(use_bvh)
    o.set_variant(cfg.variant)
    o.use_any_hit(cfg.any_hit)
    o.set_scene(cfg.scene)
    integral, _ = o.set_env(cfg.env)
    o.set_camera(cfg.camera)
    o.set_sunsky(cfg.sunsky)
    acc = o.render(cfg.state(integral), frames)
    orc.set_math_mode(0)
    if return_obj:
        return acc, o
    o.close()
    return acc


def render_hip(cfg, frames, device=0, shard=None, return_obj=False, accel=None):
    """accel: capi.PT_ACCEL_FLAT / PT_ACCEL_TWO_LEVEL (None: the context's default, i.e. flat unless PT_TUNE says accel=two)"""
    from vk_raytrace_amd.renderer import HipRenderer
    r = HipRenderer()
    r.setup(device)
    if accel is not None:
        r.set_accel_mode(accel)
    if shard is not None:
        r.set_shard(*shard)
    r.set_scene(cfg.scene)
    integral, _ = r.set_env(cfg.env)
    r.set_camera(cfg.camera)
    r.set_sunsky(cfg.sunsky)
    r.set_variant(cfg.variant)
    r.useAnyHit(cfg.any_hit)
    r.create((cfg.width, cfg.height))
    st = cfg.state(integral)
    for f in range(frames):
        st.frame = f
        r.setPushContants(st)
        r.run(None, (cfg.width, cfg.height), None, None)
    acc = r.read_accum()
    if return_obj:
        return acc, r
    r.destroy()
    return acc


def l2(a, b):
    """BASELINE.md parity metric: sqrt(mean over pixels and RGB of (a-b)^2) on the linear accumulation buffer."""
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))


def mismatch_fraction(a, b, rtol=1e-3, atol=1e-4):
    d = np.abs(a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64))
    bad = d > (atol + rtol * np.abs(b[..., :3]))
    return float(np.mean(np.any(bad, axis=-1)))


def noise_floor(cfg, frames):
    """Mismatch fraction between two renders of the oracle that differ only in how libm rounds its
    transcendental functions: glibc fp32 (practically always correctly rounded) vs a model of the GPU's ocml
    (correctly rounded result moved by 1 ulp in ~30 % of the calls, cf. profiles/r01_ocml_ulps.txt).  This is
    the difference one must expect between the CPU oracle and ANY correct GPU implementation."""
    a = render_oracle(cfg, frames, math_mode=0)
    b = render_oracle(cfg, frames, math_mode=2)
    return mismatch_fraction(b, a), a
