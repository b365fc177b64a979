"""The 8-wide quantised acceleration structure (vk_raytrace_amd/csrc/pt_cwbvh.h) as the product's collapse builds it -- run through its host
emulation (pt_debug_cw_collapse: the same cw_* bodies k_collapse8 runs on the device) by the CPU harness tests/cpp/trace_host.cpp.

Structure invariants, checked node by node in double precision (th_check_structure): the decoded child boxes enclose everything below them
(so the walk can never lose a triangle to the quantisation), inner children are consecutive nodes, leaf triangles consecutive slots, every
triangle is referenced exactly once, empty slots are inert, the non-opaque tags are exact.  That the WALK over this structure reports
exactly the candidates of a brute-force loop is tests/test_trace_host.py."""
import numpy as np
import pytest

from tests.test_trace_host import Traced, TracedScene, instanced_scene
from vk_raytrace_amd import synth, workloads
from vk_raytrace_amd.scene import Scene


def check(tr, which=0):
    import ctypes as C
    out = np.zeros(6, np.uint32)
    tr.L.th_check_structure.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    tr.L.th_check_structure(tr.h, which, out.ctypes.data)
    return dict(zip(("violations", "nodes", "triangles", "depth", "children", "leaf_children"), (int(x) for x in out)))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_invariants_on_instanced_scenes(seed):
    sc, flags, _ = instanced_scene(seed, far=seed == 2)
    tr = Traced(sc, flags)
    flat, tlas = check(tr, 0), check(tr, 1)
    sizes = tr.sizes()
    assert flat["violations"] == 0 and tlas["violations"] == 0, (flat, tlas)
    assert flat["triangles"] == tr.n and flat["nodes"] == sizes[0]              # everything reachable, nothing allocated in vain
    assert tlas["triangles"] == sum(1 for m, pm in sc.nodes if sc.prim_meshes[pm][3] > 0) and tlas["nodes"] == sizes[2]
    assert flat["children"] / flat["nodes"] > 4.0                                 # the nodes are really wide (8 slots)
    assert flat["depth"] <= 24 and flat["nodes"] <= tr.n // 2 + 2
    tr.close()


def test_invariants_on_the_c3_stand_in_with_alpha_geometry():
    wl = workloads.c3_sponza(tex_size=16, target_tris=60_000)
    tr = TracedScene(wl.scene)
    flat = check(tr, 0)
    assert flat["violations"] == 0 and flat["triangles"] == tr.n, flat
    assert flat["depth"] <= 20
    tr.close()


def test_degenerate_inputs_still_encode():
    """one triangle, two coincident triangles, a flat (zero-extent) axis, coordinates far from the origin"""
    for pts in ([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[1e4, 1e4, 1e4], [1e4 + 1e-2, 1e4, 1e4], [1e4, 1e4 + 1e-2, 1e4]]):
        for copies in (1, 2, 9):
            sc = Scene("tiny")
            m = sc.add_material()
            p = np.array(pts, np.float32)
            pm = sc.add_prim_mesh(np.tile(p, (copies, 1)), np.tile([[0, 0, 1]], (3 * copies, 1)), np.zeros((3 * copies, 2)), np.arange(3 * copies), m)
            sc.add_node(pm)
            tr = Traced(sc, np.array([3], np.uint32))
            r = check(tr, 0)
            assert r["violations"] == 0 and r["triangles"] == copies, (pts, copies, r)
            tr.close()
