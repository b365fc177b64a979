"""CPU experiment (no GPU): how deep does the traversal stack of the persistent trace kernels get?  The product's lane machine (pt_machine.h, compiled
for the host by tests/cpp/trace_host.cpp) walks camera, bounce and shadow rays over the C3 stand-in's flat structure; per ray the deepest stack level
used.  The kernels keep the first STACK_LDS = 24 levels in LDS and the rest in a private array (the 160 B of "scratch" in profiles/*_kernel_usage.txt):
this says how often that array is touched at all.
   python tools/stack_depth_experiment.py [rays] [c3|c5]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_trace_host as T  # noqa: E402
from vk_raytrace_amd import workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
which = sys.argv[2] if len(sys.argv) > 2 else "c3"
wl = workloads.c3_sponza(tex_size=64) if which == "c3" else workloads.c5_bistro(tex_size=32)
tr = T.TracedScene(wl.scene)
L = tr.L
L.th_take_sp_hist.argtypes = [C.c_void_p]
rng = np.random.default_rng(1)
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
seeds = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
w, tuv, _, _ = tr.settle(0, 0, 0, o0, d0, seeds)
hit = w != T.NONE
p1 = (o0 + tuv[:, :1].astype(np.float64) * d0)[hit]
d1 = rng.normal(0, 1, (len(p1), 3)); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
o1 = p1 + d1 * 1e-3
ds = rng.normal(0, 1, (len(p1), 3)); ds[:, 1] = np.abs(ds[:, 1]); ds /= np.linalg.norm(ds, axis=1, keepdims=True)
os_ = p1 + ds * 1e-3
tmax = np.full(len(p1), 1e32, np.float32)
print(f"{which} stand-in, {tr.n} triangles; per ray the deepest level of the traversal stack (entries = 32-bit child references: leaf bit + alpha bit + up to 22 bits of slot)")
for name, kind, o, d, tm in (("camera rays", 0, o0, d0, None), ("bounce rays (random direction from a surface point)", 0, o1, d1, None), ("shadow rays (upper hemisphere, unbounded)", 1, os_, ds, tmax)):
    o32, d32 = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
    hist = np.zeros(65, np.uint64)
    L.th_take_sp_hist(hist.ctypes.data)
    tr.settle(kind, 0, 2, o32, d32, seeds[:len(o32)], tm)
    L.th_take_sp_hist(hist.ctypes.data)
    tot = hist.sum()
    cum = np.cumsum(hist) / tot
    deepest = int(np.nonzero(hist)[0].max())
    print(f"  {name}: {int(tot)} walks; mean {float((hist * np.arange(65)).sum() / tot):.1f}, p99 {int(np.searchsorted(cum, 0.99))}, p99.99 {int(np.searchsorted(cum, 0.9999))}, deepest {deepest}; "
          f"walks that go beyond 16 levels {float(hist[17:].sum() / tot) * 100:.4f} %, beyond 24 (the LDS part) {float(hist[25:].sum() / tot) * 100:.5f} %")
tr.close()
