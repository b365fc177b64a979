"""CPU experiment (no GPU): node visits / triangle tests per ray on a 4-wide collapse of a binned-SAH BVH against the same with spatial splits of
opaque triangles (tools/sbvh_experiment.cpp), on the bench scene (C3 stand-in) or the instanced C5 stand-in, for camera rays and two diffuse bounces.
   python tools/sbvh_experiment.py [c3|c5] [camera rays, default 40000] [reference budget, default 0.3] [references per leaf, default 1]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vk_raytrace_amd import capi, workloads

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
nrays = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
max_leaf = int(sys.argv[4]) if len(sys.argv) > 4 else 1
wl = workloads.c3_sponza(1920, 1080, 8, tex_size=64) if which == "c3" else workloads.c5_bistro(tex_size=64)
sc = wl.scene
sc.finalize(capi.pack_vertices)
pos = np.ascontiguousarray(sc.vertices["position"], np.float64)
idx = np.asarray(sc.indices, np.int64)
tris, opaque = [], []
for m, pm in sc.nodes:
    vo, vc, fi, ic, mi = sc.prim_meshes[pm]
    mat = sc.materials[max(mi, 0)]
    op = int(mat["alphaMode"]) == 0 or (float(mat["pbrBaseColorFactor"][3]) == 1.0 and int(mat["pbrBaseColorTexture"]) < 0)  # src/accelstruct.cpp:144-146
    M = np.asarray(m, np.float64).reshape(4, 4)
    p = pos[vo + idx[fi:fi + ic]]
    w = p @ M[:3, :3].T + M[:3, 3]
    tris.append(w.reshape(-1, 9))
    opaque.append(np.full(len(w) // 3, 1 if op else 0, np.uint32))
tris = np.concatenate(tris).astype(np.float32)
opaque = np.concatenate(opaque)
# camera rays: pinhole through a regular sub-grid of the image
cam = sc.camera
eye, center, up = (np.asarray(v, np.float64) for v in (cam.eye, cam.center, cam.up))
f = center - eye; f /= np.linalg.norm(f)
r = np.cross(f, up); r /= np.linalg.norm(r)
u = np.cross(r, f)
aspect = wl.width / wl.height
th = np.tan(np.radians(cam.fov) / 2)
ny = int(np.sqrt(nrays / aspect)); nx = int(ny * aspect)
ys, xs = np.meshgrid((np.arange(ny) + 0.5) / ny * 2 - 1, (np.arange(nx) + 0.5) / nx * 2 - 1, indexing="ij")
d = f[None, None, :] + xs[..., None] * th * aspect * r[None, None, :] - ys[..., None] * th * u[None, None, :]
d /= np.linalg.norm(d, axis=-1, keepdims=True)
rays = np.concatenate([np.broadcast_to(eye, d.shape), d], -1).reshape(-1, 6).astype(np.float32)
with tempfile.TemporaryDirectory() as tmp:
    exe = os.path.join(tmp, "sbvh")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(os.path.dirname(os.path.abspath(__file__)), "sbvh_experiment.cpp"), "-o", exe])
    with open(os.path.join(tmp, "t.bin"), "wb") as fh:
        fh.write(np.uint32(len(tris)).tobytes())
        rec = np.zeros(len(tris), np.dtype([("v", np.float32, 9), ("o", np.uint32)]))
        rec["v"], rec["o"] = tris, opaque
        fh.write(rec.tobytes())
    with open(os.path.join(tmp, "r.bin"), "wb") as fh:
        fh.write(np.uint32(len(rays)).tobytes()); fh.write(rays.tobytes())
    print(wl.name)
    sys.stdout.flush()
    subprocess.check_call([exe, os.path.join(tmp, "t.bin"), os.path.join(tmp, "r.bin"), str(budget), str(max_leaf)])
