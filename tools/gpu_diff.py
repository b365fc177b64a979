"""GPU box: distribution of HIP-vs-oracle differences on the reduced C3 scene."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.common import Config, render_oracle, render_hip
from vk_raytrace_amd import workloads, host_device as hd
wl = workloads.c3_sponza(480, 270, 8, tex_size=128, env_w=512)
for frames, depth in ((1, 1), (1, 2), (1, 8), (8, 8)):
    cfg = Config(wl.scene, wl.env, wl.width, wl.height, depth=depth, pbr=0)
    h, o = render_hip(cfg, frames), render_oracle(cfg, frames)
    d = np.abs(h[..., :3] - o[..., :3]).max(-1)
    s = np.abs(o[..., :3]).max(-1)
    rel = d / (s + 1e-4)
    print(f"frames {frames} depth {depth}: ", " ".join(f">{t:g}:{(rel > t).mean():.5f}" for t in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1)), "max abs", d.max())
    bad = np.argwhere(rel > 1e-3)[:6]
    for y, x in bad:
        print("   px", x, y, "hip", h[y, x, :3], "orc", o[y, x, :3])
