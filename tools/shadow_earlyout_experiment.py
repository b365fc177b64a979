"""CPU experiment (no GPU): how many shadow rays of the C3 stand-in an exact early-out could serve.  A shadow ray may stop at the FIRST certain hit it finds
(instead of searching for the nearest one) only if no non-opaque candidate lies in front of that hit -- otherwise the draws consumed in front of the nearest
certain hit (trace contract T6) need the full front-to-back walk.  Counts, for shadow rays from primary hit points towards the sun and towards uniformly
random sky directions: occluded rays, and occluded rays with no draw at all (the ones an early-out helps).  Uses the product's settle logic on the host
(tests/cpp/trace_host.cpp th_settle).
   python tools/shadow_earlyout_experiment.py [rays]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_trace_host as T  # noqa: E402
from vk_raytrace_amd import workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
wl = workloads.c3_sponza(tex_size=256)
tr = T.TracedScene(wl.scene)
rng = np.random.default_rng(3)
cam = wl.scene.camera
eye = np.array(cam.eye, np.float64)
fwd = np.array(cam.center, np.float64) - eye
fwd /= np.linalg.norm(fwd)
right = np.cross(fwd, np.array(cam.up, np.float64)); right /= np.linalg.norm(right)
up = np.cross(right, fwd)
th = np.tan(np.radians(cam.fov) / 2)
px = rng.uniform(-1, 1, (n, 2)) * (th * 16 / 9, th)
d0 = fwd + px[:, :1] * right + px[:, 1:] * up
d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
o0 = np.repeat(eye[None], n, 0)
w, tuv, _, _ = tr.settle(0, 0, 0, o0, d0, np.zeros(n, np.uint32))
hit = w != T.NONE
p1 = (o0 + tuv[:, :1].astype(np.float64) * d0)[hit]
sun = np.array([np.cos(np.radians(45)) * np.sin(np.radians(30)), np.sin(np.radians(45)), np.cos(np.radians(45)) * np.cos(np.radians(30))])
sky = rng.normal(0, 1, (len(p1), 3)); sky[:, 1] = np.abs(sky[:, 1]); sky /= np.linalg.norm(sky, axis=1, keepdims=True)
print(f"C3 stand-in, {tr.n} triangles; {len(p1)} shadow rays per class (origins: primary hit points)")
for name, d in (("towards the sun", np.repeat(sun[None], len(p1), 0)), ("towards random sky directions", sky)):
    o = p1 + d * 1e-3
    seeds = rng.integers(0, 2 ** 32, len(o), dtype=np.uint64).astype(np.uint32)
    w, tuv, sd, dr = tr.settle(1, 0, 0, o, d, seeds, tmax=np.full(len(o), 1e32, np.float32))
    occ = w == 1
    nodraw = occ & (dr == 0)
    print(f"  {name:32s} occluded {occ.mean():.3f}   occluded with no draw in front (early-out applies) {nodraw.mean():.3f}   "
          f"occluded but draws in front (needs the full walk) {(occ & (dr > 0)).mean():.3f}   unoccluded {1 - occ.mean():.3f} (full walk by definition)")
tr.close()
