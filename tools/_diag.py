import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.common import Config, render_hip, render_oracle
from vk_raytrace_amd import synth
env = synth.procedural_sky(256, 128)
for ms, w, h, frames in ((3, 200, 150, 2), (1, 200, 150, 2), (2, 320, 240, 4), (3, 200, 150, 1)):
    cfg = Config(synth.feature_box(tex_size=64), env, w, h, max_samples=ms)
    res = {}
    for tune in ("wave=0,tail=65536", "wave=1,tail=65536", "wave=1,tail=65536"):
        os.environ["PT_TUNE"] = tune
        img, r = render_hip(cfg, frames, return_obj=True)
        st = r.stats(); r.destroy()
        res.setdefault(tune, []).append((img, st))
    a = res["wave=0,tail=65536"][0]
    for k, (img, st) in enumerate(res["wave=1,tail=65536"]):
        bad = np.any(img.view(np.uint32) != a[0].view(np.uint32), axis=-1)
        ys, xs = np.nonzero(bad)
        print(f"ms={ms} {w}x{h} frames={frames} run{k}: differing pixels {bad.sum()} first {list(zip(xs[:6], ys[:6]))}",
              {k2: (st[k2], a[1][k2]) for k2 in ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests") if st[k2] != a[1][k2]})
