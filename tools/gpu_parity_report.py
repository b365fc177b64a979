"""GPU box: the parity figures of BASELINE.md section 5 in one table -- per-pixel L2 (sqrt(mean over
pixels and RGB of (a-b)^2) on the linear accumulation buffer) of the HIP path against the CPU oracle at equal spp and seeds,
next to the libm noise floor (oracle vs oracle with an ocml-like 1-ulp error model) and the Monte-Carlo noise of the image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.common import Config, render_hip, render_oracle, l2, mismatch_fraction
from vk_raytrace_amd import synth, workloads

rows = []
def report(name, cfg, frames):
    h = render_hip(cfg, frames)
    o = render_oracle(cfg, frames)
    o2 = render_oracle(cfg, frames, math_mode=2)
    ref = render_oracle(cfg, 4 * frames)  # 4x the samples: a proxy for the converged image
    rms = float(np.sqrt(np.mean(o[..., :3].astype(np.float64) ** 2)))
    rows.append((name, frames, l2(h, o), l2(o2, o), l2(o, ref), rms, mismatch_fraction(h, o), mismatch_fraction(o2, o), float(np.abs(h[..., :3] - o[..., :3]).max())))

c1 = workloads.c1_quad()
report("C1 quad 256x256 frame 0", Config(c1.scene, c1.env, 256, 256, depth=10), 1)
env = synth.procedural_sky(256, 128)
report("feature box 320x240 Disney depth 10", Config(synth.feature_box(tex_size=64), env, 320, 240), 8)
c2 = workloads.c2_helmet(0.25)
report("C2 helmet-like 256x256 glTF-PBR depth 4", Config(c2.scene, c2.env, 256, 256, depth=4, pbr=1), 8)
sp = synth.sponza_like(tex_size=256)
report("C3 sponza-like (269 k tris) 480x270 depth 8 Disney", Config(sp, synth.procedural_sky(512, 256), 480, 270, depth=8), 8)
report("C3 sponza-like, depth 2", Config(sp, synth.procedural_sky(512, 256), 480, 270, depth=2), 8)
print("| scene | spp | L2 HIP vs oracle | L2 oracle(ocml-like libm) vs oracle | L2 oracle vs 4x spp (MC noise) | image RMS | mismatching pixels HIP / libm model | max abs diff |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {r[1]} | {r[2]:.3e} | {r[3]:.3e} | {r[4]:.3e} | {r[5]:.3f} | {r[6]:.2e} / {r[7]:.2e} | {r[8]:.3e} |")
