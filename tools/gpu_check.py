"""Diagnostic driver for a GPU box: renders a few configurations with the HIP path and the CPU oracle
and prints per-configuration difference statistics (used during bring-up; the asserting versions live
in tests/test_gpu_*.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.common import Config, render_oracle, render_hip, l2, mismatch_fraction
from vk_raytrace_amd import synth, host_device as hd

def report(name, a, b):
    d = np.abs(a[..., :3] - b[..., :3])
    print(f"{name:34s} exact={np.array_equal(a, b)} max={d.max():.3e} l2={l2(a,b):.3e} mismatch={mismatch_fraction(a,b):.5f} "
          f"mean_o={b[...,:3].mean():.4f} mean_h={a[...,:3].mean():.4f} nan={np.isnan(a).sum()}", flush=True)

env_small = synth.procedural_sky(256, 128)
quad = Config(synth.quad_scene(), synth.constant_env(), 256, 256)
report("quad f0", render_hip(quad, 1), render_oracle(quad, 1))
names = {1: "basecolor", 2: "normal", 3: "metallic", 4: "emissive", 5: "alpha", 6: "roughness", 7: "texcoord", 8: "tangent"}
for lights in (False,):
    for dbg, nm in names.items():
        c = Config(synth.feature_box(tex_size=64, lights=lights), env_small, 320, 240, debug=dbg)
        report(f"fbox aov {nm}", render_hip(c, 1), render_oracle(c, 1))
for pbr in (0, 1):
    c = Config(synth.feature_box(tex_size=64), env_small, 320, 240, pbr=pbr)
    for frames in (1, 8):
        t = time.time(); h = render_hip(c, frames); th = time.time() - t
        t = time.time(); o = render_oracle(c, frames); to = time.time() - t
        report(f"fbox pbr{pbr} frames{frames} ({th:.2f}s/{to:.2f}s)", h, o)
c = Config(synth.feature_box(tex_size=64, lights=True), env_small, 320, 240)
report("fbox lights f4", render_hip(c, 4), render_oracle(c, 4))
ss = hd.default_sun_and_sky(); ss.in_use = 1
c = Config(synth.feature_box(tex_size=64), env_small, 320, 240, sunsky=ss)
report("fbox sunsky f4", render_hip(c, 4), render_oracle(c, 4))
c = Config(synth.feature_box(tex_size=64), env_small, 320, 240, max_samples=3)
report("fbox maxSamples3 f2", render_hip(c, 2), render_oracle(c, 2))
for dbg in (9, 10, 11):
    c = Config(synth.feature_box(tex_size=64), env_small, 160, 120, debug=dbg, depth=3)
    report(f"fbox dbg{dbg} depth3", render_hip(c, 1), render_oracle(c, 1))
# stats
h, r = render_hip(Config(synth.feature_box(tex_size=64), env_small, 320, 240), 2, return_obj=True)
print(r.stats())
