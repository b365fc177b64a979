"""CPU experiment (no GPU): how often do the walks (flat, two-level) and brute force disagree on the candidates of a ray -- under the contract's
fp32 Moeller-Trumbore (T2) and under a candidate replacement that re-evaluates in double precision whenever the fp32 verdict could be an artefact of
rounding (tests/cpp/trace_host.cpp -DTH_ROBUST_T2)?  The product's traversal source runs on the host (tests/test_trace_host.py's harness).
   python tools/t2_robust_experiment.py [rays per scene]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_trace_host as T  # noqa: E402
from vk_raytrace_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
robust = os.path.join(ROOT, "tests", "cpp", "_build", "libtracehost_robust.so")
T.harness()  # builds the contract's flavour
lib_dir = os.path.dirname(capi.LIB_PATH)
subprocess.check_call(["g++", "-std=c++17", "-O2", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-DSTACK_LDS=24", "-DTH_ROBUST_T2", "-Wno-attributes",
                       "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "vk_raytrace_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), T.SRC,
                       "-L" + lib_dir, "-l:libptmi.so", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", robust])
certified = os.path.join(ROOT, "tests", "cpp", "_build", "libtracehost_certified.so")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-DSTACK_LDS=24", "-DTH_CERTIFIED_T2", "-Wno-attributes",
                       "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "vk_raytrace_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), T.SRC,
                       "-L" + lib_dir, "-l:libptmi.so", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", certified])
names = ["camera", "surface", "axis", "far"]
for flavour, path in (("fp32 T2 (the contract)", T.OUT), ("fp32 T2 + fp64 when ambiguous", robust), ("fp32 T2, accepted only when certified to 2^-7 (no fp64)", certified)):
    T.OUT = path
    tot = np.zeros(4, int)
    bad = np.zeros((2, 4), int)
    cand = 0
    t2 = np.zeros((3, 2), np.int64)  # per mode (brute force, flat walk, two-level walk): triangle tests, re-evaluated in double
    stats = C.CDLL(path).th_t2_stats if path in (robust, certified) else None
    t0 = time.time()
    for seed in range(20, 26):
        sc, flags, off = T.instanced_scene(seed, n_nodes=220, far=False)
        tr = T.Traced(sc, flags)
        rng = np.random.default_rng(900 + seed)
        org, dirs = T.rays_for(tr, rng, off, n)
        ref_w, ref_t = tr.candidates(0, org, dirs, max_cand=4)
        cand += int((ref_w != T.NONE).sum())
        if stats:
            st = (C.c_ulonglong * 2)(); stats(st); t2[0] += (st[0], st[1])
        k = n // 4
        for mode in (1, 2):
            w, t = tr.candidates(mode, org, dirs, max_cand=4)
            if stats:
                st = (C.c_ulonglong * 2)(); stats(st); t2[mode] += (st[0], st[1])
            rows = np.nonzero(((w != ref_w) | (t.view(np.uint32) != ref_t.view(np.uint32))).any(1))[0]
            for r in rows:
                bad[mode - 1][min(3, r // k)] += 1
        tot += np.array([k, k, k, n - 3 * k])
        tr.close()
    print(f"{flavour}: {tot.sum()} rays, {cand} brute-force candidates, {time.time() - t0:.0f} s")
    if path in (robust, certified):
        what = "re-evaluated in double" if path == robust else "fp32 accepts turned into misses"
        for i, nm in enumerate(("brute force", "flat walk", "two-level walk")):
            print(f"   {nm:15s} triangle tests {t2[i][0]:12d}, {what} {t2[i][1]:10d} ({100.0 * t2[i][1] / max(1, t2[i][0]):.4f} %)")
    for i, nm in enumerate(names):
        print(f"   {nm:8s} rays {tot[i]:7d}   rays on which the flat walk differs from brute force: {bad[0][i]:4d}   two-level walk: {bad[1][i]:4d}")

# what certification would remove from a render: the bench scene's own rays (camera rays, bounce rays between surface points, shadow rays to the sun)
T.OUT = certified
from vk_raytrace_amd import workloads  # noqa: E402
wl = workloads.c3_sponza(tex_size=64)
tr = T.TracedScene(wl.scene)
L = C.CDLL(certified)
L.th_t2_accepts.restype = C.c_uint64
rng = np.random.default_rng(5)
cam = wl.scene.camera
eye = np.array(cam.eye, np.float64); fwd = np.array(cam.center, np.float64) - eye; fwd /= np.linalg.norm(fwd)
right = np.cross(fwd, np.array(cam.up, np.float64)); right /= np.linalg.norm(right); up = np.cross(right, fwd)
th = np.tan(np.radians(cam.fov) / 2)
m = 60000
px = rng.uniform(-1, 1, (m, 2)) * (th * 16 / 9, th)
d0 = fwd + px[:, :1] * right + px[:, 1:] * up; d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
o0 = np.repeat(eye[None], m, 0)
st = (C.c_ulonglong * 2)(); L.th_t2_stats(st); L.th_t2_accepts()
w, tuv, _, _ = tr.settle(0, 0, 0, o0, d0, np.zeros(m, np.uint32))
L.th_t2_stats(st); acc = L.th_t2_accepts()
print(f"bench scene (C3 stand-in), certified T2: camera rays: {st[0]} tests, {acc} fp32 accepts, {st[1]} of them not certifiable ({100.0 * st[1] / max(1, acc):.4f} %)")
hit = w != T.NONE
p1 = (o0 + tuv[:, :1].astype(np.float64) * d0)[hit]
dd = rng.normal(0, 1, (len(p1), 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
nrm = np.array([np.cross(tr.world_tri(int(x))[0][3:6], tr.world_tri(int(x))[0][6:9]) for x in w[hit]])
nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
nrm *= np.where((nrm * d0[hit]).sum(1, keepdims=True) > 0, -1.0, 1.0)          # the side the camera ray came from
dd *= np.where((dd * nrm).sum(1, keepdims=True) < 0, -1.0, 1.0)                 # leave on that side (what a reflected bounce ray does)
tr.settle(0, 0, 0, p1 + nrm * 1e-4, dd, np.zeros(len(p1), np.uint32))
L.th_t2_stats(st); acc = L.th_t2_accepts()
print(f"                                          bounce rays: {st[0]} tests, {acc} fp32 accepts, {st[1]} not certifiable ({100.0 * st[1] / max(1, acc):.4f} %)")
tr.close()
