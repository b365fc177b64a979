"""The product's traversal source on the CPU: vk_raytrace_amd/csrc/pt_trace.h (traverse<MODE, TWO>, the fused slab test, the two-level
walk with its per-instance box padding, tri_test) compiled for the host by tests/cpp/trace_host.cpp and held, ray by ray, to a brute-force
loop over every world triangle with the same triangle test.

Claim under test ("trace contract"): the walk reports exactly the candidates brute force reports -- every triangle T2
accepts along the ray, in key order (t, world index) -- for the flat structure and for the two-level structure, on instanced scenes with
scaled / rotated / mirrored / far-translated instances, for camera rays, rays between surface points (bounce rays start a few ulps off a
surface), axis-parallel rays and rays from far outside the scene.  A candidate on which a walk and brute force disagree is acceptable only
if fp32's verdict on one of the triangles involved is an artefact of cancellation: an ACCIDENTAL hit (Moeller-Trumbore accepting a triangle
the ray misses in double precision) or a hit distance that is off by more than the box tolerance (nearly edge-on triangle) -- such a
candidate is found or not depending on the shape of the boxes around it and on the order of the walk (DESIGN.md section 3 documents the
one case seen on the GPU).  Even the flat walk differs from brute force in such cases; what must never happen is a walk losing a
well-conditioned hit.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from vk_raytrace_amd import capi, host_device as hd, synth
from vk_raytrace_amd.scene import Scene, translate, scale, rotate_x, rotate_y, rotate_z

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "trace_host.cpp")
OUT = os.path.join(ROOT, "tests", "cpp", "_build", "libtracehost.so")
NONE = 0xFFFFFFFF
OPAQUE, NOCULL = 1, 2


class InstIn(C.Structure):
    _fields_ = [("vertexOffset", C.c_uint32), ("firstIndex", C.c_uint32), ("triCount", C.c_uint32), ("flags", C.c_uint32), ("primMesh", C.c_int32), ("worldMatrix", C.c_float * 16)]


FLAVOUR = ""  # "" = the product's defaults; other keys of FLAVOURS build the harness with measurement flags
FLAVOURS = {"": []}


def harness():
    capi.lib()  # libptmi.so must exist: the harness links its test hooks (device-builder emulation, two_level_pad)
    global OUT
    OUT = os.path.join(ROOT, "tests", "cpp", "_build", "libtracehost%s.so" % ("_" + FLAVOUR if FLAVOUR else ""))
    deps = [SRC] + [os.path.join(ROOT, "vk_raytrace_amd", "csrc", f) for f in ("pt_trace.h", "pt_machine.h", "pt_settle.h", "pt_shade.h", "pt_surface.h", "pt_device.h", "pt_math.h", "pt_cnode.h")] + [capi.LIB_PATH]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        lib_dir = os.path.dirname(capi.LIB_PATH)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-DSTACK_LDS=24", "-Wno-attributes"] + FLAVOURS[FLAVOUR] + [
                               "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "vk_raytrace_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), SRC,
                               "-L" + lib_dir, "-l:libptmi.so", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", OUT])
    L = C.CDLL(OUT)
    L.th_create.restype = C.c_void_p
    L.th_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.th_destroy.argtypes = [C.c_void_p]
    L.th_compact_in_use.argtypes = [C.c_void_p, C.c_int]
    L.th_num_tris.restype = C.c_uint32
    L.th_num_tris.argtypes = [C.c_void_p]
    L.th_sizes.argtypes = [C.c_void_p, C.c_void_p]
    L.th_world_tri.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.th_candidates.restype = C.c_uint32
    L.th_candidates.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p]
    return L


class Traced:
    def __init__(self, scene: Scene, flags, merge_singles=True):
        """flags: per node TRI_OPAQUE | TRI_NOCULL bits.  merge_singles: the two-level structure keeps the prim-meshes instantiated once in one
        world-space structure (the product's default, PT_TUNE mergeSingles) or gives every prim-mesh its own object-space BLAS"""
        self.L = harness()
        self.L.th_set_merge_singles(1 if merge_singles else 0)
        if scene.vertices is None:
            scene.finalize(capi.pack_vertices)
        v = np.ascontiguousarray(scene.vertices)
        idx = np.ascontiguousarray(scene.indices, np.uint32)
        inst = (InstIn * len(scene.nodes))()
        for i, (m, pm) in enumerate(scene.nodes):
            vo, vc, fi, ic, _ = scene.prim_meshes[pm]
            inst[i] = InstIn(vo, fi, ic // 3, int(flags[i]), pm, (C.c_float * 16)(*np.asarray(m, np.float32).T.reshape(16)))
        bound = np.zeros(len(scene.prim_meshes), np.float32)
        for p, (vo, vc, fi, ic, _) in enumerate(scene.prim_meshes):
            bound[p] = np.abs(v["position"][vo:vo + vc]).max() if vc else 0.0
        self.h = self.L.th_create(v.ctypes.data, len(v), idx.ctypes.data, len(idx), inst, len(inst), bound.ctypes.data, len(bound))
        self.L.th_set_merge_singles(1)
        assert self.h, "th_create failed"
        self.keep = (v, idx, inst, bound)
        self.n = self.L.th_num_tris(self.h)

    def candidates(self, mode, org, dirs, tmax=1e32, max_cand=6):
        org, dirs = np.ascontiguousarray(org, np.float32), np.ascontiguousarray(dirs, np.float32)
        w = np.zeros((len(org), max_cand), np.uint32)
        t = np.zeros((len(org), max_cand), np.float32)
        over = self.L.th_candidates(self.h, mode, len(org), org.ctypes.data, dirs.ctypes.data, tmax, max_cand, w.ctypes.data, t.ctypes.data)
        assert over == 0, "traversal stack overflow"
        return w, t

    def world_tri(self, w):
        out = np.zeros(9, np.float32)
        fl = C.c_uint32()
        self.L.th_world_tri(self.h, int(w), out.ctypes.data, C.byref(fl))
        return out.astype(np.float64), fl.value

    def sizes(self):
        out = np.zeros(4, np.uint32)
        self.L.th_sizes(self.h, out.ctypes.data)
        return out

    def close(self):
        self.L.th_destroy(self.h)


def ill_conditioned(tr, o, d, t32):
    """Moeller-Trumbore in double precision on the fp32 inputs.  True when fp32's verdict on this triangle is an artefact of cancellation:
    the ray misses the triangle in exact arithmetic (an ACCIDENTAL hit), or the fp32 hit distance is off by more than the box tests'
    tolerance (the triangle is nearly edge-on: det ~ 0), so that pruning against it -- or it against another candidate -- depends on the
    order in which a walk meets them."""
    tri, _ = tr
    p0, e1, e2 = tri[0:3], tri[3:6], tri[6:9]
    o, d = o.astype(np.float64), d.astype(np.float64)
    pv = np.cross(d, e2)
    det = e1 @ pv
    if det == 0.0:
        return True
    tv = o - p0
    u = (tv @ pv) / det
    qv = np.cross(tv, e1)
    v = (d @ qv) / det
    t = (e2 @ qv) / det
    eps = 1e-9
    if u < -eps or v < -eps or u + v > 1 + eps:
        return True
    return abs(float(t32) - t) > 4e-7 * abs(t) + 1e-30


def instanced_scene(seed, n_nodes=160, far=False):
    rng = np.random.default_rng(seed)
    sc = Scene(f"trace{seed}")
    m = sc.add_material()
    meshes = [synth.uv_sphere(0.5, 16, 8), synth.box((0.8, 0.9, 0.7), sub=3), synth.revolve(0.2 + 0.1 * np.sin(np.linspace(0, 3, 9)), np.linspace(0, 1, 9), 12),
              synth.cards(rng, 40, (0, 0.5, 0), (0.8, 1.0, 0.8), 0.2), synth.grid(6, 6, (-1, 0, 1), (2, 0, 0), (0, 0, -2))]
    pms = [sc.add_prim_mesh(p, n, uv, i, m, tangents=t) for (p, n, uv, i, t) in meshes]
    tri = sc.add_prim_mesh([(-1, -1, 0), (1, -1, 0), (0, 1, 0)], [(0, 0, 1)] * 3, [(0, 0), (1, 0), (0.5, 1)], [0, 1, 2], m)   # single-leaf BLAS
    hole = sc.add_prim_mesh(np.zeros((3, 3)), [(0, 0, 1)] * 3, np.zeros((3, 2)), np.zeros(0, np.uint32), m)                      # no triangles
    flags = []
    off = np.array([3000.0, -1500.0, 800.0]) if far else np.zeros(3)
    for i in range(n_nodes):
        s = 10.0 ** rng.uniform(-0.7, 0.7, 3) if i % 3 else np.full(3, 10.0 ** rng.uniform(-0.5, 0.5))
        if i % 7 == 0:
            s[rng.integers(3)] *= -1.0   # mirrored
        t = rng.uniform(-6, 6, 3) + off
        mtx = translate(*t) @ rotate_y(rng.uniform(0, 6.3)) @ rotate_x(rng.uniform(0, 6.3)) @ rotate_z(rng.uniform(0, 6.3)) @ scale(*s)
        if i % 11 == 0:
            mtx = translate(*t)          # axis-aligned instances: boxes whose faces are parallel to axis-parallel rays
        pm = (pms + [tri, hole])[i % 7]
        sc.add_node(pm, mtx)
        flags.append(OPAQUE | (NOCULL if i % 2 else 0))
    # two instances of the same mesh exactly on top of each other: ties in t across instances
    mtx = translate(*(np.array([0.5, 0.5, 0.5]) + off))
    sc.add_node(pms[1], mtx); flags.append(OPAQUE | NOCULL)
    sc.add_node(pms[1], mtx); flags.append(OPAQUE | NOCULL)
    # prim-meshes with ONE instance each (the two-level structure keeps these in its merged world-space structure): rotated + non-uniformly
    # scaled, mirrored, and one overlapping the coincident pair above
    once = [sc.add_prim_mesh(p, n, uv, i, m, tangents=t) for (p, n, uv, i, t) in (synth.uv_sphere(0.7, 12, 6), synth.box((1.1, 0.6, 0.9), sub=2), synth.grid(4, 4, (-1, 0, 1), (2, 0, 0), (0, 0, -2)))]
    sc.add_node(once[0], translate(*(np.array([-2.0, 1.0, 3.0]) + off)) @ rotate_y(0.7) @ rotate_x(1.9) @ scale(1.5, 0.4, 2.2)); flags.append(OPAQUE)
    sc.add_node(once[1], translate(*(np.array([0.6, 0.4, 0.5]) + off)) @ rotate_z(0.3) @ scale(-1.0, 1.0, 1.0)); flags.append(OPAQUE | NOCULL)
    sc.add_node(once[2], translate(*(np.array([1.0, -2.0, -1.0]) + off)) @ rotate_x(0.4) @ scale(3.0, 1.0, 3.0)); flags.append(OPAQUE | NOCULL)
    return sc, np.array(flags), off


def rays_for(tr: Traced, rng, off, n):
    """camera-like, surface-to-surface, axis-parallel and far-origin rays"""
    org, dirs = [], []
    # towards random triangles from a ring of eye points
    k = n // 4
    targets = rng.integers(0, tr.n, k)
    pts = []
    for w in targets:
        tri, _ = tr.world_tri(w)
        b = rng.dirichlet((1, 1, 1))
        pts.append(tri[0:3] + b[1] * tri[3:6] + b[2] * tri[6:9])
    pts = np.array(pts)
    eye = off + rng.normal(0, 1, (k, 3)) * 14.0
    org.append(eye); dirs.append(pts - eye)
    # between surface points (what a bounce ray is), started a few ulps off the surface
    a, b = pts[rng.permutation(k)], pts[rng.permutation(k)]
    org.append(a + (b - a) * 1e-6); dirs.append(b - a)
    # axis-parallel rays through the scene, some exactly through lattice-like coordinates
    o = off + np.round(rng.uniform(-7, 7, (k, 3)) * 2) / 2
    ax = np.eye(3)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1))
    org.append(o - ax * 20); dirs.append(ax)
    # from far outside
    o = off + rng.normal(0, 1, (n - 3 * k, 3)) * 3000.0
    org.append(o); dirs.append(pts[rng.integers(0, k, n - 3 * k)] - o)
    org, dirs = np.concatenate(org), np.concatenate(dirs)
    dirs = dirs / np.maximum(np.linalg.norm(dirs, axis=1, keepdims=True), 1e-30)
    return org.astype(np.float32), dirs.astype(np.float32)


def compare(tr, org, dirs, what, max_cand=6):
    ref_w, ref_t = tr.candidates(0, org, dirs, max_cand=max_cand)
    total = int((ref_w != NONE).sum())
    accidental = 0
    for mode, name in ((1, "flat"), (2, "two-level")):
        w, t = tr.candidates(mode, org, dirs, max_cand=max_cand)
        bad = np.nonzero(((w != ref_w) | (t.view(np.uint32) != ref_t.view(np.uint32))).any(1))[0]
        for r in bad:
            # the first differing position: one side reports a triangle the other skips.  Tolerated only if fp32's verdict on one of the two
            # triangles involved is an artefact (ill_conditioned); a well-conditioned hit that a walk loses is a hole in its box tests.
            c = int(np.nonzero((w[r] != ref_w[r]) | (t[r].view(np.uint32) != ref_t[r].view(np.uint32)))[0][0])
            involved = [(ref_w[r, c], ref_t[r, c]), (w[r, c], t[r, c])]
            assert any(x != NONE and ill_conditioned(tr.world_tri(x), org[r], dirs[r], tx) for x, tx in involved), \
                f"{what}, {name}: ray {r} candidate {c}: brute force {ref_w[r]} {ref_t[r]} vs walk {w[r]} {t[r]}"
            accidental += 1
    return total, accidental


@pytest.mark.parametrize("seed,merge", [(0, True), (1, True), (2, True), (3, True), (0, False), (3, False)])
def test_walks_report_brute_force_candidates(seed, merge):
    sc, flags, off = instanced_scene(seed)
    tr = Traced(sc, flags, merge_singles=merge)
    rng = np.random.default_rng(100 + seed)
    org, dirs = rays_for(tr, rng, off, 6000)
    total, accidental = compare(tr, org, dirs, f"scene {seed}")
    assert total > 8000                       # the rays do hit things (several candidates each)
    # Every disagreement was verified to be an accidental hit (compare()).  They are not rare HERE because a quarter of the rays start thousands
    # of units away, where |o - p0| ~ 5e3 leaves fp32 only ~1e-3 of absolute resolution in Moeller-Trumbore's numerators (brute force then reports
    # hits at t = 4096.0 exactly and the like; both walks, flat and two-level, never get near those triangles).  On the GPU workloads
    # (camera inside the scene) the rate is ~1e-8 per ray.  A hole in the box tests would show up as hundreds of GENUINE misses, not as these.
    assert accidental <= 12, accidental
    nflat, nblas, ntlas, nslots = tr.sizes()
    assert nblas < nflat and nslots < tr.n    # meshes are stored once in the two-level structure
    tr.close()


def test_far_from_the_origin_and_shadow_range():
    """instances around (3000, -1500, 800): fp32 resolution there is 2.4e-4, the object-space padding must absorb the rounding of the ray
    transform; bounded rays (tmax) prune the same candidates on all sides"""
    sc, flags, off = instanced_scene(11, n_nodes=90, far=True)
    tr = Traced(sc, flags)
    rng = np.random.default_rng(7)
    org, dirs = rays_for(tr, rng, off, 4000)
    total, accidental = compare(tr, org, dirs, "far scene")
    assert total > 4000 and accidental <= 12, (total, accidental)
    ref_w, ref_t = tr.candidates(0, org[:1500], dirs[:1500], tmax=9.0)
    for mode in (1, 2):
        w, t = tr.candidates(mode, org[:1500], dirs[:1500], tmax=9.0)
        same = (w == ref_w).all(1) & (t.view(np.uint32) == ref_t.view(np.uint32)).all(1)
        assert same.mean() > 0.999
        assert (ref_t[ref_w != NONE] < 9.0).all()
    tr.close()


def test_degenerate_inputs():
    """one triangle in one instance (single-leaf BLAS and TLAS), an empty scene, a scene of empty instances"""
    sc = Scene("one")
    m = sc.add_material()
    sc.add_node(sc.add_prim_mesh([(-1, -1, 0), (1, -1, 0), (0, 1, 0)], [(0, 0, 1)] * 3, [(0, 0), (1, 0), (0.5, 1)], [0, 1, 2], m), translate(0.2, 0.1, -1.0) @ rotate_y(0.4))
    tr = Traced(sc, [OPAQUE | NOCULL])
    org = np.array([[0, 0, 3], [0.2, 0.1, 3], [5, 5, 5]], np.float32)
    dirs = np.array([[0, 0, -1], [0, 0, -1], [0, 0, -1]], np.float32)
    ref = tr.candidates(0, org, dirs)
    for mode in (1, 2):
        got = tr.candidates(mode, org, dirs)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
    assert (ref[0][:2, 0] == 0).all() and ref[0][2, 0] == NONE
    tr.close()
    empty = Scene("empty")
    m = empty.add_material()
    hole = empty.add_prim_mesh(np.zeros((3, 3)), [(0, 0, 1)] * 3, np.zeros((3, 2)), np.zeros(0, np.uint32), m)
    empty.add_node(hole); empty.add_node(hole, translate(1, 2, 3))
    tr = Traced(empty, [OPAQUE, OPAQUE])
    for mode in (0, 1, 2):
        assert (tr.candidates(mode, org, dirs)[0] == NONE).all()
    tr.close()


# ---- stochastic alpha: the product's two-pass settle functions against the contract's key-ordered loop -------------------------------------
class TracedScene(Traced):
    """a full scene description (materials, textures): instance flags, alpha view, opacity maps and texel pool come from the product's own
    host code (pt_capi.hip build_scene_records)"""

    def __init__(self, scene: Scene):
        self.L = harness()
        self.L.th_create_scene.restype = C.c_void_p
        self.L.th_create_scene.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self.L.th_settle.restype = C.c_uint32
        self.L.th_settle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32] + [C.c_void_p] * 8
        if scene.vertices is None:
            scene.finalize(capi.pack_vertices)
        d, keep = scene.desc()
        err = C.create_string_buffer(256)
        self.h = self.L.th_create_scene(C.byref(d), err, 256)
        assert self.h, err.value
        self.keep = keep
        self.n = self.L.th_num_tris(self.h)

    def settle(self, kind, two, exact, org, dirs, seeds, tmax=None, variant=0):
        org, dirs = np.ascontiguousarray(org, np.float32), np.ascontiguousarray(dirs, np.float32)
        seeds = np.ascontiguousarray(seeds, np.uint32)
        n = len(org)
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float32)
        w, tuv, sd, dr = np.zeros(n, np.uint32), np.zeros((n, 3), np.float32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        over = self.L.th_settle(self.h, kind, two, exact, variant, n, org.ctypes.data, dirs.ctypes.data, tm.ctypes.data if tm is not None else None, seeds.ctypes.data,
                                w.ctypes.data, tuv.ctypes.data, sd.ctypes.data, dr.ctypes.data)
        assert over == 0
        return w, tuv, sd, dr


def scene_rays(tr, rng, n, eye_center, eye_spread):
    k = n // 2
    targets = rng.integers(0, tr.n, n)
    pts = []
    for w in targets:
        tri, _ = tr.world_tri(w)
        b = rng.dirichlet((1, 1, 1))
        pts.append(tri[0:3] + b[1] * tri[3:6] + b[2] * tri[6:9])
    pts = np.array(pts)
    eye = np.asarray(eye_center) + rng.normal(0, 1, (k, 3)) * eye_spread
    a, b = pts[:n - k], pts[rng.permutation(n)[:n - k]]
    org = np.concatenate([eye, a + (b - a) * 1e-5])
    dirs = np.concatenate([pts[:k] - eye, b - a])
    dirs = dirs / np.maximum(np.linalg.norm(dirs, axis=1, keepdims=True), 1e-30)
    return org.astype(np.float32), dirs.astype(np.float32)


def _alpha_scenes():
    yield "fuzz0", synth.fuzz_scene(0), (0, 0, 6), 3.0
    yield "fuzz1", synth.fuzz_scene(1), (0, 0, 6), 3.0
    yield "fuzz5", synth.fuzz_scene(5), (0, 0, 6), 3.0
    yield "sponza-like", synth.sponza_like(target_tris=12000, tex_size=64), (0, 3, 0), 4.0   # foliage cards: MASK with power-of-two textures -> opacity maps


@pytest.mark.parametrize("name,scene,eye,spread", list(_alpha_scenes()), ids=lambda x: x if isinstance(x, str) else None)
def test_two_pass_alpha_equals_the_key_ordered_loop(name, scene, eye, spread):
    """trace contract T5 / T6 with stochastic alpha: the product's settle functions (pass A nearest certain hit, pass B count of the zero-
    opacity candidates in front of it, draws consumed in bulk, exact fallback; opacity maps answering most evaluations) -- in the lock-step
    form k_tail and the k_*_s kernels use (traverse<>) and in the resumable per-lane form of the persistent kernels (pt_machine.h) -- must return the hit,
    the barycentrics AND the RNG state of the definition -- candidates strictly in key order, one draw per non-opaque candidate -- on the flat
    and on the two-level structure, for closest-hit rays, bounded shadow rays and the RT-pipeline flavour of the shadow ray."""
    tr = TracedScene(scene)
    rng = np.random.default_rng(4242)
    n = 6000
    org, dirs = scene_rays(tr, rng, n, eye, spread)
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    ref = tr.settle(0, 0, 1, org, dirs, seeds)                      # the definition, flat structure
    assert (ref[0] != NONE).mean() > 0.5 and ref[3].sum() > n // 20, "the scene must exercise hits and alpha draws"
    for two, exact in ((0, 0), (1, 0), (1, 1), (0, 2), (1, 2)):   # exact = 2: the trace machine of the persistent kernels (pt_machine.h)
        got = tr.settle(0, two, exact, org, dirs, seeds)
        same = (got[0] == ref[0]) & (got[1].view(np.uint32) == ref[1].view(np.uint32)).all(1) & (got[2] == ref[2])
        # (a ray may differ only through an ill-conditioned candidate, see test_walks_report_brute_force_candidates: none is expected at this size)
        assert same.all(), f"{name}: closest-hit, two={two} exact={exact}: {np.count_nonzero(~same)} rays differ, first {np.nonzero(~same)[0][:5]}"
        if exact != 1:
            assert np.array_equal(got[3], ref[3])                   # the alpha-test counter (pt_Stats.alphaTests) counts the same draws
    # the persistent kernels on the compact form of the flat structure's nodes (PT_TUNE cnodes=1: 80-byte nodes, fp16 grid planes): the boxes are
    # looser, never tighter -- same hits, barycentrics, RNG states and draw counts
    tr.L.th_set_compact_nodes(1)
    trc = TracedScene(scene)
    tr.L.th_set_compact_nodes(0)
    assert trc.L.th_compact_ok() == 1
    for two in (0, 1):
        got = trc.settle(0, two, 2, org, dirs, seeds)
        same = (got[0] == ref[0]) & (got[1].view(np.uint32) == ref[1].view(np.uint32)).all(1) & (got[2] == ref[2])
        assert same.all(), f"{name}: closest-hit on compact nodes, two={two}: {np.count_nonzero(~same)} rays differ, first {np.nonzero(~same)[0][:5]}"
        assert np.array_equal(got[3], ref[3])
    tmax = np.where(rng.random(n) < 0.3, np.float32(1e32), rng.uniform(0.3, 12.0, n)).astype(np.float32)
    # ... and bounded shadow rays on them (the machine's loop as k_trace_p / k_closest_p / k_shadow_p drive it: lane_inner, then lane_leaf)
    for variant in (0, 1):
        want = tr.settle(1, 0, 1, org, dirs, seeds, tmax, variant)
        got = trc.settle(1, 0, 2, org, dirs, seeds, tmax, variant)
        assert (got[0] == want[0]).all() and (got[2] == want[2]).all(), f"{name}: shadow rays on compact nodes, variant={variant}"
    trc.close()
    for variant in (0, 1):
        ref = tr.settle(1, 0, 1, org, dirs, seeds, tmax, variant)
        for two, exact in ((0, 0), (1, 0), (1, 1), (0, 2), (1, 2)):
            got = tr.settle(1, two, exact, org, dirs, seeds, tmax, variant)
            same = (got[0] == ref[0]) & (got[2] == ref[2])
            assert same.all(), f"{name}: shadow variant {variant}, two={two} exact={exact}: {np.count_nonzero(~same)} rays differ"
        if variant == 1:
            assert np.array_equal(ref[2], seeds)                    # RT pipeline: the any-hit shader draws from a copy of the seed
    tr.close()


@pytest.mark.parametrize("far", [False, True])
def test_compact_nodes_never_lose_a_hit(far):
    """The 80-byte form of the nodes (pt_cnode.h cn_encode, pt_trace.h wide_node_step_c) on the adversarial instanced scene -- scales 0.2 .. 5, mirrored
    and coincident instances, rays that start thousands of units away (far=True: every coordinate carries an offset of thousands, the worst case
    for the per-node grid's p * idir + n): the persistent kernels' walk on compact nodes returns the hits of the walk on the fp32 nodes."""
    sc, flags, off = instanced_scene(11, far=far)
    tr = TracedScene(sc)
    tr.L.th_set_compact_nodes(1)
    trc = TracedScene(sc)
    tr.L.th_set_compact_nodes(0)
    assert trc.L.th_compact_ok() == 1
    rng = np.random.default_rng(5)
    org, dirs = rays_for(tr, rng, off, 8000)
    seeds = np.zeros(len(org), np.uint32)
    # the property itself, plane by plane in double: every decoded box encloses the fp32 box it stands for, the child references are the same
    trc.L.th_cnode_violations.restype = C.c_ulonglong
    trc.L.th_cnode_violations.argtypes = [C.c_void_p, C.c_void_p]
    loose = C.c_double()
    assert trc.L.th_cnode_violations(trc.h, C.byref(loose)) == 0
    assert 0.0 <= loose.value <= 2.0 / 2047, loose.value  # and by at most one grid step per plane (mean growth of a child's extent, in units of the node's grid extent)
    for two in (0, 1):   # flat structure; two-level structure (TLAS + object-space BLASes with their per-instance padding + the merged structure)
        assert trc.L.th_compact_in_use(trc.h, two) == 1 and tr.L.th_compact_in_use(tr.h, two) == 0
        want = tr.settle(0, two, 2, org, dirs, seeds)
        got = trc.settle(0, two, 2, org, dirs, seeds)
        assert (want[0] != NONE).mean() > 0.3
        same = (got[0] == want[0]) & (got[1].view(np.uint32) == want[1].view(np.uint32)).all(1)
        assert same.all(), f"two={two}: {np.count_nonzero(~same)} rays differ, first {np.nonzero(~same)[0][:5]}"
    tr.close(); trc.close()


# ---- whole frames: the product's shading source on the host against the oracle ---------------------------------------------------------------
def host_render(cfg, frames, two=0, shard=None):
    """cfg: tests.common.Config.  The frames k_generate / k_tail / k_accumulate would produce, computed by the same functions (pt_shade.h,
    pt_settle.h, pt_trace.h, pt_bsdf.h, pt_surface.h, pt_sky.h) compiled for the host.  shard = (rank, nranks): only that rank's image tiles
    (pt_set_shard), the other pixels stay zero."""
    tr = TracedScene(cfg.scene)
    L = tr.L
    L.th_set_env.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.th_set_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.th_render_shard.restype = C.c_uint32
    L.th_render_shard.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    integral = C.c_float()
    assert L.th_set_env(tr.h, cfg.env.ctypes.data, cfg.env.shape[1], cfg.env.shape[0], C.byref(integral)) == 0
    L.th_set_camera(tr.h, C.byref(cfg.camera), C.byref(cfg.sunsky))
    st = cfg.state(integral.value)
    out = np.zeros((cfg.height, cfg.width, 4), np.float32)
    rank, nranks = shard if shard is not None else (0, 1)
    assert L.th_render_shard(tr.h, two, C.byref(st), cfg.variant, frames, rank, nranks, out.ctypes.data) == 0
    tr.close()
    return out


def _bits_equal(a, b):
    an, bn = np.isnan(a), np.isnan(b)
    return np.array_equal(an, bn) and np.array_equal(np.where(an, 0, a).view(np.uint32), np.where(bn, 0, b).view(np.uint32))


@pytest.mark.parametrize("two", [0, 1])
def test_host_build_of_the_shading_source_renders_the_oracles_frames(two):
    """End to end without a GPU: camera ray, traversal with stochastic alpha, shading state, materials and textures, environment NEE through the
    alias table, both BSDFs, punctual lights, sun & sky, Russian roulette, firefly clamp and the running mean -- the functions the HIP kernels
    are made of, compiled by g++ -- give the CPU oracle's accumulation images BIT FOR BIT (the oracle is in turn bit-identical to the
    reference's own shaders, tests/test_oracle_vs_ref.py).  On the GPU the same comparison is tests/test_gpu_parity.py."""
    from tests.common import Config, render_oracle
    env = synth.procedural_sky(128, 64)
    # every material feature + punctual lights, both BSDFs, several samples per frame
    for pbr in (0, 1):
        cfg = Config(synth.feature_box(tex_size=32, lights=True), env, 64, 48, depth=6, pbr=pbr, max_samples=2)
        assert _bits_equal(host_render(cfg, 2, two), render_oracle(cfg, 2)), ("feature box", pbr)
    # first-hit AOVs
    for mode in (hd.eNormal, hd.eTexcoord, hd.eBaseColor, hd.eAlpha):
        cfg = Config(synth.feature_box(tex_size=32), env, 64, 48, debug=mode)
        assert _bits_equal(host_render(cfg, 1, two), render_oracle(cfg, 1)), mode
    # adversarial geometry with MASK / BLEND soups, the RT-pipeline flavour, sun & sky instead of the environment map
    ss = hd.default_sun_and_sky(); ss.in_use = 1
    cfg = Config(synth.fuzz_scene(2), env, 96, 64, depth=5, variant=capi.PT_VARIANT_RTX, sunsky=ss)
    assert _bits_equal(host_render(cfg, 3, two), render_oracle(cfg, 3)), "fuzz scene, RTX variant, sun & sky"
    # an image that is not a multiple of the tile size, depth of field
    sc = synth.fuzz_scene(4); sc.camera.aperture = 0.05
    cfg = Config(sc, env, 50, 37, depth=4, hdr_multiplier=2.0)
    assert _bits_equal(host_render(cfg, 2, two), render_oracle(cfg, 2)), "odd size, depth of field"


def test_host_build_renders_the_c3_stand_in_like_the_oracle():
    """the bench scene (269 k triangles, alpha-tested foliage cards with opacity maps, HDR environment, Disney BSDF, depth 8) at a reduced
    resolution, four frames: host build of the product's source == oracle, bit for bit, on both acceleration structures"""
    from tests.common import Config, render_oracle
    from vk_raytrace_amd import workloads
    wl = workloads.c3_sponza(160, 90, 4, tex_size=64, env_w=256)
    cfg = Config(wl.scene, wl.env, wl.width, wl.height, depth=wl.depth, pbr=wl.pbr_mode)
    ref = render_oracle(cfg, 4)
    assert np.isfinite(ref).all() and ref[..., :3].max() > 0
    for two in (0, 1):
        assert _bits_equal(host_render(cfg, 4, two), ref), two
