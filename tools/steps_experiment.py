"""CPU experiment (no GPU): dependent memory round trips per ray of a closest-hit walk over the C3 stand-in's flat structure -- the quantity that
bounds the trace stages (a stage lasts as long as its slowest ray, a chain of node / triangle fetches) -- for the product's
4-wide nodes with one triangle per step (what the kernels do today), with the hit leaf children of a node fetched together, and for 8-wide nodes.
Uses the binary tree the device builder produces (host emulation) and the test harness tests/cpp/trace_host.cpp (th_step_model).
   python tools/steps_experiment.py [rays]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_trace_host as T  # noqa: E402
from vk_raytrace_amd import workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
wl = workloads.c3_sponza(tex_size=64)
tr = T.TracedScene(wl.scene)
L = tr.L
L.th_step_model.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
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
seeds = np.zeros(n, np.uint32)
w, tuv, _, _ = tr.settle(0, 0, 0, o0, d0, seeds)
hit = w != T.NONE
p1 = (o0 + tuv[:, :1].astype(np.float64) * d0)[hit]
d1 = rng.normal(0, 1, (len(p1), 3)); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
o1 = p1 + d1 * 1e-3
sun = np.array([np.cos(np.radians(45)) * np.sin(np.radians(30)), np.sin(np.radians(45)), np.cos(np.radians(45)) * np.cos(np.radians(30))])
ds = np.repeat(sun[None], len(p1), 0)
os_ = p1 + ds * 1e-3
classes = [("primary (camera)", o0, d0), ("bounce (random direction from a surface point)", o1, d1), ("shadow (towards the sun from a surface point)", os_, ds)]
print(f"C3 stand-in, {tr.n} triangles; {n} rays per class")
for name, o, d in classes:
    o32, d32 = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
    print(name)
    base = None
    for width, batch, label in ((4, 0, "4-wide, one triangle per step (today)"), (4, 1, "4-wide, a node's hit leaves in one step"), (8, 0, "8-wide, one triangle per step"), (8, 1, "8-wide, a node's hit leaves in one step")):
        out = np.zeros((len(o32), 3), np.uint32)
        L.th_step_model(tr.h, width, batch, len(o32), o32.ctypes.data, d32.ctypes.data, None, out.ctypes.data)
        st = out[:, 0].astype(np.float64)
        base = base or st.mean()
        print(f"   {label:42s} steps/ray mean {st.mean():6.1f} ({st.mean() / base:4.2f}x)  p99 {np.percentile(st, 99):6.0f}  max {st.max():5.0f}   nodes {out[:, 1].mean():6.1f}  triangles {out[:, 2].mean():5.1f}")
tr.close()
