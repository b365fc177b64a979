"""Two-level acceleration structure (PT_ACCEL_TWO_LEVEL: one object-space BLAS per prim-mesh + a TLAS over the instances; reference:
src/accelstruct.cpp:110-162) and instance updates (pt_update_instances: TLAS refit).

CPU part: the object-space box padding of the walk (pt_capi.hip two_level_pad / pt_trace.h enter_instance) is held to a float32 emulation
of the ray transform -- the padded box must contain the transformed ray's point for every hit the world-space triangle test can report.
GPU part: frames, AOVs, counters and picks of the two-level mode are bit-identical to the oracle (the trace contract is BVH independent),
before and after instance updates, under every launch policy.
"""
import ctypes as C
import os

import numpy as np
import pytest

from vk_raytrace_amd import capi, host_device as hd, synth
from vk_raytrace_amd.scene import Scene, Camera, translate, scale, rotate_x, rotate_y, rotate_z

F = np.float32


def _pad_record(m_rowmajor, bo):
    out = (C.c_float * 27)()
    L = capi.lib()
    L.pt_debug_two_level_pad.restype = C.c_int
    L.pt_debug_two_level_pad.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
    colmajor = np.ascontiguousarray(np.asarray(m_rowmajor, F).T.reshape(16))
    assert L.pt_debug_two_level_pad(colmajor.ctypes.data, float(bo), out) == capi.PT_OK
    a = np.array(out[:], F)
    return a[:12].reshape(3, 4), a[12:24].reshape(3, 4), float(a[24]), float(a[25]), int(a[26])


def _xform_point_f32(rows, p):
    """csrc/pt_math.h xform_point in float32, same association: ((r.x * x + r.y * y) + r.z * z) + r.w"""
    p = np.asarray(p, F)
    return np.stack([((rows[i, 0] * p[..., 0] + rows[i, 1] * p[..., 1]) + rows[i, 2] * p[..., 2]) + rows[i, 3] for i in range(3)], -1).astype(F)


def _xform_dir_f32(rows, d):
    d = np.asarray(d, F)
    return np.stack([(rows[i, 0] * d[..., 0] + rows[i, 1] * d[..., 1]) + rows[i, 2] * d[..., 2] for i in range(3)], -1).astype(F)


@pytest.mark.parametrize("seed", range(6))
def test_object_space_padding_bounds_the_ray_transform(seed):
    """For random instances (scales 1e-2..1e2, any rotation, mirrored or not, translations up to 1e4) and random rays that hit a world
    triangle: the point the OBJECT-space ray reaches at the hit parameter lies within eps = padC1 * max|o| + padC0 of the object-space
    triangle's bounding box.  (The walk grows every BLAS box by eps, so the subtree holding the triangle cannot be culled.)"""
    rng = np.random.default_rng(1000 + seed)
    worst = 0.0
    for _ in range(40):
        s = 10.0 ** rng.uniform(-2, 2, 3) * rng.choice([-1.0, 1.0], 3)
        tr = rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(0, 4)
        m = translate(*tr) @ rotate_y(rng.uniform(0, 6.3)) @ rotate_x(rng.uniform(0, 6.3)) @ rotate_z(rng.uniform(0, 6.3)) @ scale(*s)
        obj_scale = 10.0 ** rng.uniform(-1, 2)
        n = 400
        tri = (rng.uniform(-1, 1, (n, 1, 3)) * obj_scale + rng.normal(0, 0.05 * obj_scale, (n, 3, 3))).astype(F)  # object-space triangles
        bo = float(np.abs(tri).max())
        o2w, w2o, c0, c1, _ = _pad_record(m, bo)
        assert c0 > 0 and c1 > 0
        world = _xform_point_f32(o2w, tri)                       # trace contract T1 (float32)
        # a point on the world triangle and a ray towards it
        b = rng.dirichlet((1, 1, 1), n)
        pw = np.einsum("nk,nkc->nc", b, world.astype(np.float64))
        far = rng.random(n) < 0.3
        o = (pw + rng.normal(0, 1, (n, 3)) * np.where(far, 1e4, 10.0 ** rng.uniform(-1, 2, n))[:, None]).astype(F)
        d = pw - o.astype(np.float64)
        tlen = np.linalg.norm(d, axis=1, keepdims=True)
        d = (d / tlen).astype(F)
        t = np.einsum("nc,nc->n", pw - o.astype(np.float64), d.astype(np.float64))  # parameter of the closest approach: the hit (pw up to 1e-7 |pw|)
        oo, dd = _xform_point_f32(w2o, o), _xform_dir_f32(w2o, d)
        reached = oo.astype(np.float64) + t[:, None] * dd.astype(np.float64)     # where the object-space ray is at the hit parameter
        lo, hi = tri.min(1).astype(np.float64), tri.max(1).astype(np.float64)
        # distance outside the object-space triangle box (the leaf padding of tri_box, 4e-6 |coordinate|, comes on top of eps and is not used here)
        out = np.maximum(np.maximum(lo - reached, reached - hi), 0.0).max(1)
        eps = c1 * np.abs(o).max(1).astype(np.float64) + c0
        worst = max(worst, float((out / eps).max()))
        assert (out <= eps).all(), (seed, float((out / eps).max()))
    assert worst < 0.25, worst  # the bound keeps a >= 4x margin over everything this experiment produces (measured: ~0.03)


def test_pad_record_matches_set_scene_rules():
    """pt_debug_two_level_pad builds the record the way pt_set_scene does: rows of the column-major matrix, inverse in double rounded once,
    TRI_FLIP (4) for a mirroring matrix."""
    m = translate(3, -2, 5) @ rotate_y(0.7) @ scale(2.0, 0.5, -1.5)
    o2w, w2o, c0, c1, flags = _pad_record(m, 1.0)
    assert np.array_equal(o2w, m[:3].astype(F))
    assert np.allclose(w2o, np.linalg.inv(m.astype(np.float64))[:3], rtol=1e-6, atol=1e-6)
    assert flags == 4
    assert _pad_record(translate(1, 2, 3), 1.0)[4] == 0
    # the padding scales with the distance from the origin and with the mesh extent
    near, far = _pad_record(translate(1, 0, 0), 1.0), _pad_record(translate(1e4, 0, 0), 1.0)
    assert far[2] > 100 * near[2] and far[3] == near[3]
    assert _pad_record(translate(1, 0, 0), 100.0)[2] > 10 * near[2]


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def env_small():
    return synth.procedural_sky(256, 128)


@pytest.fixture(params=["tail=0", "tail=65536", "tail=65536,mergeSingles=0"])
def tail_policy(request):
    """staged kernels vs the fused tail kernel (see tests/test_gpu_parity.py); mergeSingles=0: every prim-mesh gets its own object-space BLAS
    (by default the ones instantiated once share a world-space structure, and a scene made only of those runs the flat kernels on it)"""
    old = os.environ.get("PT_TUNE")
    os.environ["PT_TUNE"] = request.param
    yield request.param
    os.environ["PT_TUNE"] = "tail=65536" if old is None else old


def _assert_identical(h, o, what=""):
    from tests.test_gpu_parity import assert_identical
    assert_identical(h, o, what)


@gpu
@pytest.mark.parametrize("seed", range(8))
def test_two_level_fuzz_scenes(env_small, seed, tail_policy):
    """Adversarial scenes (instanced boxes: translated, mirrored, scaled 40x and 0.01x, coincident triangles, MASK / BLEND soups): first-hit
    AOVs and path-traced frames of the two-level structure equal the oracle bit for bit."""
    from tests.common import Config, render_hip, render_oracle
    sc = synth.fuzz_scene(seed)
    for mode in (hd.eNormal, hd.eTexcoord, hd.eAlpha):
        cfg = Config(sc, env_small, 96, 64, debug=mode)
        assert np.array_equal(render_hip(cfg, 1, accel=capi.PT_ACCEL_TWO_LEVEL), render_oracle(cfg, 1)), (seed, mode)
    cfg = Config(sc, env_small, 96, 64, depth=5)
    _assert_identical(render_hip(cfg, 3, accel=capi.PT_ACCEL_TWO_LEVEL), render_oracle(cfg, 3), f"two-level, fuzz scene {seed}")


@gpu
@pytest.mark.parametrize("pbr", [0, 1])
def test_two_level_feature_box(env_small, pbr, tail_policy):
    """every material feature, punctual lights, both BSDFs, maxSamples > 1; counters equal the flat structure's"""
    from tests.common import Config, render_hip, render_oracle
    cfg = Config(synth.feature_box(tex_size=32, lights=True), env_small, 128, 96, depth=8, pbr=pbr, max_samples=2)
    h2, r2 = render_hip(cfg, 3, accel=capi.PT_ACCEL_TWO_LEVEL, return_obj=True)
    h1, r1 = render_hip(cfg, 3, return_obj=True)
    _assert_identical(h2, render_oracle(cfg, 3), "two-level feature box")
    assert np.array_equal(h1.view(np.uint32), h2.view(np.uint32))
    s1, s2 = r1.stats(), r2.stats()
    for k in ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups"):
        assert s1[k] == s2[k], k
    assert s2["numBlas"] > 0 and s2["numTlasNodes"] > 0 and s1["numBlas"] == 0
    r1.destroy(); r2.destroy()


def _street(n_nodes=400, seed=5):
    """instancing-heavy scene: 6 prim-meshes (one of them alpha-tested cards), hundreds of nodes"""
    rng = np.random.default_rng(seed)
    sc = Scene("street")
    tex = sc.add_texture(synth.tex_leaf(rng, 64))
    mats = [sc.add_material(pbrBaseColorFactor=tuple(0.3 + 0.6 * rng.random(3)) + (1.0,), pbrRoughnessFactor=0.4 + 0.5 * rng.random(), pbrMetallicFactor=float(i == 2)) for i in range(4)]
    leaf = sc.add_material(pbrBaseColorTexture=tex, alphaMode=hd.ALPHA_MASK, doubleSided=1)
    meshes = [synth.uv_sphere(0.5, 24, 12), synth.box((0.8, 0.9, 0.8), sub=4), synth.revolve(0.2 + 0.1 * np.sin(np.linspace(0, 3, 17)), np.linspace(0, 1, 17), 24),
              synth.cards(rng, 150, (0, 0.8, 0), (1.0, 1.4, 1.0), 0.15), synth.box((0.3, 2.0, 0.3), sub=2)]
    pms = []
    for i, (pos, nrm, uv, idx, tan) in enumerate(meshes):
        pms.append(sc.add_prim_mesh(pos, nrm, uv, idx, leaf if i == 3 else mats[i % 4], tangents=tan))
    gpos, gnrm, guv, gidx, gtan = synth.grid(16, 16, (-30, 0, 30), (60, 0, 0), (0, 0, -60))
    sc.add_node(sc.add_prim_mesh(gpos, gnrm, guv, gidx, mats[0], tangents=gtan))
    for i in range(n_nodes):
        x, z = (rng.random(2) - 0.5) * 50
        s = 0.5 + 1.5 * rng.random()
        mirror = -1.0 if i % 7 == 0 else 1.0
        sc.add_node(pms[i % len(pms)], translate(x, 0.0, z) @ rotate_y(rng.random() * 6.28) @ scale(s * mirror, s * (0.8 + 0.6 * rng.random()), s))
    sc.camera = Camera(eye=(-20, 5, 18), center=(4, 1, -4), up=(0, 1, 0), fov=55.0)
    return sc


@gpu
def test_two_level_instanced_street(env_small):
    """hundreds of instances of six meshes: frames equal the oracle; the two-level structure stores every mesh once"""
    from tests.common import Config, render_hip, render_oracle
    sc = _street()
    cfg = Config(sc, env_small, 160, 96, depth=6)
    h2, r2 = render_hip(cfg, 4, accel=capi.PT_ACCEL_TWO_LEVEL, return_obj=True)
    _assert_identical(h2, render_oracle(cfg, 4), "two-level street")
    h1, r1 = render_hip(cfg, 4, return_obj=True)
    assert np.array_equal(h1.view(np.uint32), h2.view(np.uint32))
    s1, s2 = r1.stats(), r2.stats()
    assert s2["numBlas"] == 6 and s2["numTriangles"] == s1["numTriangles"] == sc.num_triangles
    assert s2["bytesAccel"] * 10 < s1["bytesAccel"], (s1["bytesAccel"], s2["bytesAccel"])
    r1.destroy(); r2.destroy()


@gpu
@pytest.mark.parametrize("accel", [capi.PT_ACCEL_FLAT, capi.PT_ACCEL_TWO_LEVEL])
def test_update_instances(env_small, accel, tail_policy):
    """pt_update_instances: after the nodes move (some get mirrored) the frames equal the oracle's render of the moved scene -- a TLAS
    refit in two-level mode, a rebuild in flat mode; moving back reproduces the first image."""
    from tests.common import Config, render_oracle
    from vk_raytrace_amd.renderer import HipRenderer
    sc = _street(120, seed=9)
    cfg = Config(sc, env_small, 128, 80, depth=5)
    r = HipRenderer(); r.setup(0); r.set_accel_mode(accel); r.set_scene(cfg.scene)
    integral, _ = r.set_env(cfg.env); r.set_camera(cfg.camera); r.set_sunsky(cfg.sunsky); r.create((cfg.width, cfg.height))

    def frames(n=3):
        st = cfg.state(integral)
        for f in range(n):
            st.frame = f
            r.setPushContants(st)
            r.run()
        return r.read_accum()

    first = frames()
    _assert_identical(first, render_oracle(cfg, 3), "before the update")
    original = [(m.copy(), p) for m, p in sc.nodes]
    rng = np.random.default_rng(77)
    for i in range(1, len(sc.nodes)):
        m, p = sc.nodes[i]
        sc.nodes[i] = (translate(*(rng.normal(0, 1.5, 3) * (1, 0.1, 1))) @ m @ rotate_y(rng.uniform(0, 6.3)) @ scale(1.0, 1.0, -1.0 if i % 5 == 0 else 1.0), p)
    r.update_instances(sc)
    moved = frames()
    assert not np.array_equal(moved, first)
    _assert_identical(moved, render_oracle(cfg, 3), "after the update")
    build_ms = r.stats()["msBuildAccel"]
    sc.nodes[:] = original
    r.update_instances(sc)
    assert np.array_equal(frames().view(np.uint32), first.view(np.uint32))
    # the ground is the one prim-mesh with a single instance: in two-level mode it lives in the merged world-space structure, which is rebuilt
    m0, p0 = sc.nodes[0]
    sc.nodes[0] = (translate(0.5, -0.4, 1.0) @ rotate_y(0.3) @ rotate_x(0.05) @ m0, p0)
    r.update_instances(sc)
    _assert_identical(frames(), render_oracle(cfg, 3), "after moving the singly instantiated ground")
    sc.nodes[:] = original
    r.update_instances(sc)
    assert np.array_equal(frames().view(np.uint32), first.view(np.uint32))
    # errors: node count and primMesh are fixed
    with pytest.raises(capi.PtError):
        r.update_instances(sc.node_array()[:-1])
    bad = sc.node_array(); bad[3]["primMesh"] = (bad[3]["primMesh"] + 1) % len(sc.prim_meshes)
    with pytest.raises(capi.PtError):
        r.update_instances(bad)
    assert build_ms > 0
    r.destroy()


@gpu
def test_two_level_mode_switch_any_hit_and_variants(env_small):
    """switching the mode on a live context rebuilds; useAnyHit(false), the RTX variant and the heat-map instantiations run on the two-level
    kernels and give the flat structure's bits"""
    from tests.common import Config, render_hip, render_oracle
    sc = synth.fuzz_scene(3)
    for kw in (dict(any_hit=False), dict(variant=capi.PT_VARIANT_RTX, max_samples=2)):
        cfg = Config(sc, env_small, 96, 64, depth=5, **kw)
        _assert_identical(render_hip(cfg, 2, accel=capi.PT_ACCEL_TWO_LEVEL), render_oracle(cfg, 2), str(kw))
    cfg = Config(sc, env_small, 96, 64, depth=5)
    img, r = render_hip(cfg, 2, return_obj=True)
    r.set_accel_mode(capi.PT_ACCEL_TWO_LEVEL)
    assert r.stats()["numBlas"] > 0
    st = cfg.state(r.env_integral)
    for f in range(2):
        st.frame = f
        r.setPushContants(st); r.run()
    assert np.array_equal(r.read_accum().view(np.uint32), img.view(np.uint32))
    r.set_accel_mode(capi.PT_ACCEL_FLAT)
    assert r.stats()["numBlas"] == 0
    # heat map: wall-clock colours differ run to run; the instantiation must run and produce a finite image
    st.debugging_mode = hd.eHeatmap; st.frame = 0
    r.set_accel_mode(capi.PT_ACCEL_TWO_LEVEL)
    r.setPushContants(st); r.run()
    heat = r.read_accum()
    assert np.isfinite(heat).all() and heat[..., :3].max() > 0
    r.destroy()


@gpu
def test_two_level_tiny_and_ragged_scenes(env_small, tail_policy):
    """edge cases of the two-level build: one triangle in one instance (single-leaf BLAS and TLAS), a scene whose only node has no
    triangles, a node without triangles between nodes with, two meshes of one triangle each, instances of the same mesh stacked exactly on
    top of each other (ties in t between instances: the world index decides)"""
    from tests.common import Config, render_hip, render_oracle
    two = capi.PT_ACCEL_TWO_LEVEL
    tri = ([(-1, -1, 0), (1, -1, 0), (0, 1, 0)], [(0, 0, 1)] * 3, [(0, 0), (1, 0), (0.5, 1)], [0, 1, 2])
    sc = Scene("tri")
    m = sc.add_material(pbrBaseColorFactor=(0.9, 0.2, 0.1, 1), doubleSided=1, pbrMetallicFactor=0.0)
    sc.add_node(sc.add_prim_mesh(*tri, m))
    sc.camera = Camera(eye=(0, 0, 3), center=(0, 0, 0), fov=45)
    cfg = Config(sc, env_small, 64, 64, debug=hd.eNormal)
    assert np.array_equal(render_hip(cfg, 1, accel=two), render_oracle(cfg, 1))
    cfg = Config(sc, env_small, 64, 64)
    _assert_identical(render_hip(cfg, 2, accel=two), render_oracle(cfg, 2), "one triangle")
    empty = Scene("empty")
    m = empty.add_material()
    empty.add_node(empty.add_prim_mesh(np.zeros((3, 3)), [(0, 0, 1)] * 3, np.zeros((3, 2)), np.zeros(0, np.uint32), m))
    empty.camera = Camera(eye=(0, 0, 3), center=(0, 0, 0), fov=45)
    cfg = Config(empty, env_small, 48, 32)
    _assert_identical(render_hip(cfg, 2, accel=two), render_oracle(cfg, 2), "empty scene")
    rag = Scene("ragged")
    m = rag.add_material(pbrBaseColorFactor=(0.2, 0.7, 0.9, 1), doubleSided=1)
    a = rag.add_prim_mesh(*tri, m)
    hole = rag.add_prim_mesh(np.zeros((3, 3)), [(0, 0, 1)] * 3, np.zeros((3, 2)), np.zeros(0, np.uint32), m)
    b = rag.add_prim_mesh([(-1, 1, -0.5), (1, 1, -0.5), (0, -1, -0.5)], [(0, 0, 1)] * 3, [(0, 0), (1, 0), (0.5, 1)], [0, 1, 2], m)
    rag.add_node(a, translate(-0.6, 0, 0)); rag.add_node(hole); rag.add_node(b); rag.add_node(hole, translate(1, 1, 1)); rag.add_node(a, translate(0.6, 0, 0.2))
    rag.add_node(a, translate(0.6, 0, 0.2))   # coincident with the previous instance
    rag.add_node(a, translate(0.6, 0, 0.2) @ scale(1.0, 1.0, -1.0))   # and once more, mirrored
    rag.camera = Camera(eye=(0, 0, 4), center=(0, 0, 0), fov=45)
    for mode in (hd.eNormal, hd.eTexcoord):
        cfg = Config(rag, env_small, 64, 48, debug=mode)
        assert np.array_equal(render_hip(cfg, 1, accel=two), render_oracle(cfg, 1)), mode
    cfg = Config(rag, env_small, 64, 48, depth=4)
    _assert_identical(render_hip(cfg, 3, accel=two), render_oracle(cfg, 3), "ragged scene")


@gpu
def test_two_level_forest_build_runs_and_small_meshes(env_small):
    """Round 6: all BLASes of two or more triangles are built as ONE forest per contiguous run of meshes (csrc/pt_accel.hip pt_blas_build / PtForest); one-triangle
    meshes between the runs keep the per-mesh path, meshes of <= 12 triangles start in the builder's small-node list, bigger ones open its first level.  Every
    mesh is instantiated twice (so that none is merged into the world-space structure): the image must equal the oracle's and the flat structure's, bit for bit,
    under the device SAH builder (forest) and under a per-mesh builder."""
    from tests.common import Config, render_hip, render_oracle
    rng = np.random.default_rng(77)
    sc = Scene("forest")
    mats = [sc.add_material(pbrBaseColorFactor=(0.3 + 0.1 * k, 0.8 - 0.1 * k, 0.5, 1), doubleSided=1, pbrRoughnessFactor=0.6) for k in range(3)]

    def soup(n_tris, spread):
        """n_tris random triangles around the origin"""
        c = rng.uniform(-spread, spread, (n_tris, 1, 3))
        v = (c + rng.normal(0, 0.25, (n_tris, 3, 3))).reshape(-1, 3)
        nrm = np.tile((0.0, 0.0, 1.0), (len(v), 1))
        uv = rng.uniform(0, 1, (len(v), 2))
        return v, nrm, uv, np.arange(len(v), dtype=np.uint32)

    sizes = [40, 2, 13, 1, 300, 12, 1, 1, 7, 90]   # runs: [40, 2, 13] | 1 | [300, 12] | 1 | 1 | [7, 90]
    for i, n in enumerate(sizes):
        pm = sc.add_prim_mesh(*soup(n, 0.8), mats[i % 3])
        sc.add_node(pm, translate(-3.0 + 0.7 * i, 0.4 * ((i % 3) - 1), 0.0))
        sc.add_node(pm, translate(-3.0 + 0.7 * i, -0.6 + 0.3 * (i % 2), -1.0) @ rotate_y(0.4 * i) @ scale(0.8, 1.1, 0.9))
    sc.camera = Camera(eye=(0.2, 0.3, 6.5), center=(0, 0, 0), fov=50)
    cfg = Config(sc, env_small, 192, 96, depth=5)
    want = render_oracle(cfg, 3)
    _assert_identical(render_hip(cfg, 3, accel=capi.PT_ACCEL_TWO_LEVEL), want, "forest build")
    _assert_identical(render_hip(cfg, 3), want, "flat structure")
    aov = Config(sc, env_small, 192, 96, debug=hd.eNormal)
    assert np.array_equal(render_hip(aov, 1, accel=capi.PT_ACCEL_TWO_LEVEL), render_oracle(aov, 1))
    # the per-mesh path (any builder but the device SAH one): PT_TUNE is read by pt_create into the context's own knobs
    keep = os.environ.get("PT_TUNE")
    os.environ["PT_TUNE"] = "build=lbvh"
    try:
        _assert_identical(render_hip(cfg, 3, accel=capi.PT_ACCEL_TWO_LEVEL), want, "one build per mesh")
    finally:
        if keep is None:
            del os.environ["PT_TUNE"]
        else:
            os.environ["PT_TUNE"] = keep


@gpu
def test_two_level_ray_picker(env_small):
    """pt_pick walks the two-level structure: same instance / primitive / t / barycentrics as the oracle's probe"""
    from tests import orc
    from tests.common import Config
    from vk_raytrace_amd.renderer import HipRenderer
    sc = Scene("pick2")
    m = sc.add_material(pbrBaseColorFactor=(0.8, 0.8, 0.8, 1.0))
    bpos, bnrm, buv, bidx, btan = synth.box((1, 1, 1))
    bm = sc.add_prim_mesh(bpos, bnrm, buv, bidx, m, tangents=btan)
    spos, snrm, suv, sidx, stan = synth.uv_sphere(0.6, 16, 8)
    sm = sc.add_prim_mesh(spos, snrm, suv, sidx, m, tangents=stan)
    rng = np.random.default_rng(3)
    for i in range(60):
        sc.add_node(bm if i % 2 else sm, translate(*rng.uniform(-3, 3, 3)) @ scale(*rng.uniform(0.5, 1.6, 3)))
    sc.camera = Camera(eye=(0.3, 0.4, 9.0), center=(0, 0, 0), up=(0, 1, 0), fov=50.0)
    cfg = Config(sc, env_small, 64, 48)
    r = HipRenderer(); r.setup(0); r.set_accel_mode(capi.PT_ACCEL_TWO_LEVEL); r.set_scene(cfg.scene); r.set_env(cfg.env); r.set_camera(cfg.camera); r.create((64, 48))
    o = orc.Oracle(); o.set_scene(cfg.scene)
    hits = 0
    for (x, y) in [(0.07 + 0.17 * i, 0.09 + 0.16 * j) for i in range(6) for j in range(6)]:
        p = r.pick(x, y, cfg.camera)
        org = np.array([list(p.worldRayOrigin)], np.float32); d = np.array([list(p.worldRayDirection)], np.float32)
        t, node, prim, uv, _ = o.trace_closest(org, d)
        if node[0] < 0:
            assert p.instanceID == 0xFFFFFFFF
            continue
        hits += 1
        assert (p.instanceID, p.primitiveID, p.instanceCustomIndex) == (int(node[0]), int(prim[0]), int(sc.nodes[int(node[0])][1]))
        assert p.hitT == t[0] and p.baryCoord[1] == uv[0, 0] and p.baryCoord[2] == uv[0, 1]
    assert hits >= 6, hits
    r.destroy(); o.close()


# ---- launch policy (host logic) -------------------------------------------------------------------------------------------------------------
def test_tail_depth_follows_the_observed_queue_sizes():
    """flush_pending's decision where the fused tail kernel takes over (pt_capi.hip tail_from_depth), on plain numbers: the first bounce
    whose expected queue is <= the threshold; observed alive fractions first, the last observed shrink factor beyond them, 0.3 per bounce
    before any feedback; never when the threshold is 0.  (Performance policy only: tests/test_gpu_parity.py holds every threshold to
    bit-identical images.)"""
    L = capi.lib()
    L.pt_debug_tail_from.restype = C.c_int
    L.pt_debug_tail_from.argtypes = [C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_int]

    def tail(paths, depth, below, ratios=()):
        r = np.asarray(ratios, np.float64)
        return L.pt_debug_tail_from(float(paths), depth, below, r.ctypes.data if len(r) else None, len(r))

    assert tail(66e6, 8, 0) == 8                                   # off
    assert tail(1000, 8, 65536) == 0                               # a tiny launch runs in the tail kernel as a whole
    assert tail(66e6, 8, 65536) == 6                               # no feedback: 0.3^6 * 66e6 = 48 k
    assert tail(66e6, 8, 65536, [1, 0.25, 0.0625, 0.0156]) == 5    # observed 4x shrink, continued: 66e6 / 4^5 = 64.5 k
    assert tail(1.3e6, 8, 65536, [1, 0.25, 0.0625]) == 3           # an 8-GPU shard's 5-frame piece: 81 k at bounce 2, 20 k at bounce 3
    assert tail(1.0e6, 8, 65536, [1, 0.25, 0.0625]) == 2
    assert tail(66e6, 8, 65536, [1, 0.9, 0.81]) == 8               # an enclosed scene that barely shrinks: staged kernels all the way
    assert tail(66e6, 3, 65536, [1, 0.25]) == 3                    # maxDepth reached first
    assert tail(2e6, 8, 65536, [1, 0.0]) == 1                      # everything missed at bounce 0
