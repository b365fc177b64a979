"""CPU experiment (no GPU): how many shadow rays of the C3 stand-in an exact early-out could serve.  A shadow ray may stop at the FIRST certain hit it finds
(instead of searching for the nearest one) only if no non-opaque candidate lies in front of that hit -- otherwise the draws consumed in front of the nearest
certain hit (trace contract T6) need the full front-to-back walk.  Counts, for shadow rays from primary hit points towards the sun and towards uniformly
random sky directions: occluded rays, and occluded rays with no draw at all (the ones an early-out helps).  Uses the product's settle logic on the host
(tests/cpp/trace_host.cpp th_settle).
Round 4: the early-out itself (pt_machine.h EARLY: after an opaque hit with nothing non-opaque seen, only BVH_ALPHA-tagged references are entered; a
non-opaque candidate in front of the hit starts the plain walk over) run ray by ray on the trace machine of the persistent kernels, with and without
it: node steps, triangle steps, walks that started over; results (verdict + RNG state) must be identical.
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
    import ctypes as C
    for f in ("th_take_inner_steps", "th_take_leaf_steps", "th_take_restarts"):
        getattr(tr.L, f).restype = C.c_uint64
    res = {}
    for early in (0, 1):
        tr.L.th_set_shadow_early(early)
        tr.L.th_take_inner_steps(); tr.L.th_take_leaf_steps(); tr.L.th_take_restarts()
        wm, _, sdm, drm = tr.settle(1, 0, 2, o, d, seeds, tmax=np.full(len(o), 1e32, np.float32))
        res[early] = (tr.L.th_take_inner_steps(), tr.L.th_take_leaf_steps(), tr.L.th_take_restarts(), wm, sdm)
        assert np.array_equal(wm, w) and np.array_equal(sdm, sd), "the machine walk differs from the definition"
    (n0, l0, _, _, _), (n1, l1, r1, _, _) = res[0], res[1]
    print(f"      trace machine, per ray: node steps {n0 / len(o):.2f} -> {n1 / len(o):.2f} ({n1 / n0:.3f}x), triangle steps {l0 / len(o):.2f} -> {l1 / len(o):.2f} ({l1 / l0:.3f}x), "
          f"walks started over {r1 / len(o):.3f} of the rays; verdicts and RNG states identical")
tr.L.th_set_shadow_early(1)
tr.close()
