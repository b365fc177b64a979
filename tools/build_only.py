"""GPU box: uploads a BASELINE workload's scene and builds its acceleration structure (nothing is rendered); prints the build time.
usage: [PT_TUNE=build=sahdev|sah|ploc|lbvh] python tools/build_only.py [c3|c5] [repeats]   (run under rocprofv3 --kernel-trace --stats for the kernel split)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk_raytrace_amd import capi, workloads
from vk_raytrace_amd.renderer import HipRenderer

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = workloads.c3_sponza(tex_size=64) if which == "c3" else workloads.c5_bistro(tex_size=64)
wl.scene.finalize(capi.pack_vertices)
r = HipRenderer()
r.setup(0)
for i in range(reps):
    t = time.perf_counter()
    r.set_scene(wl.scene)
    dt = time.perf_counter() - t
    print(f"{which} {wl.scene.num_triangles} tris: pt_set_scene + pt_build_accel {dt * 1e3:.1f} ms, msBuildAccel {r.stats()['msBuildAccel']:.1f} ms", flush=True)
r.destroy()
