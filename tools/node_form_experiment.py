"""CPU experiment (no GPU): node visits per ray of the persistent kernels' walk (pt_machine.h on the host) over the C3 stand-in's flat structure with the
fp32 nodes, the 80-byte compact nodes (fp16 grid planes) and the experimental 64-byte nodes (8-bit planes) -- looser boxes mean more visits --
and the sixteen-byte requests per ray that follow (7 / 5 / 4 per node visit + 3 per triangle test).
   python tools/node_form_experiment.py [rays]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402
import tests.test_trace_host as T  # noqa: E402
from vk_raytrace_amd import workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
wl = workloads.c3_sponza(tex_size=64)
rng = np.random.default_rng(2)
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
base = T.TracedScene(wl.scene)
base.L.th_take_inner_steps.restype = C.c_ulonglong
w, tuv, _, _ = base.settle(0, 0, 0, o0, d0, np.zeros(n, np.uint32))
hit = w != T.NONE
p1 = (o0 + tuv[:, :1].astype(np.float64) * d0)[hit]
d1 = rng.normal(0, 1, (len(p1), 3)); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
o1 = p1 + d1 * 1e-3
seeds = np.zeros(len(o1), np.uint32)
print(f"C3 stand-in, {base.n} triangles; {len(o1)} bounce rays (random direction from a primary hit point), walk of the persistent kernels on the host")
tri_tests = 6.3  # per ray, tools/steps_experiment.py
ref = None
for form, name, req in ((0, "fp32 nodes (128 B, 7 requests)", 7), (1, "compact nodes (80 B, fp16 planes, 5 requests)", 5), (2, "EXPERIMENT 64-byte nodes (8-bit planes, 4 requests)", 4)):
    base.L.th_set_compact_nodes(form)
    tr = T.TracedScene(wl.scene)
    base.L.th_set_compact_nodes(0)
    tr.L.th_take_inner_steps.restype = C.c_ulonglong
    tr.L.th_take_inner_steps()
    got = tr.settle(0, 0, 2, o1, d1, seeds)
    steps = tr.L.th_take_inner_steps() / len(o1)
    if ref is None:
        ref = got
    same = (got[0] == ref[0]).all()
    print(f"  {name:52s} node visits / ray {steps:6.2f}   requests / ray {steps * req + tri_tests * 3:6.1f}   hits identical: {same}")
    tr.close()
base.close()
