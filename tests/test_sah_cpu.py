"""The host SAH topology builder (vk_raytrace_amd/csrc/pt_sah.hip) checked on the CPU: the output must be a valid
binary tree over a permutation of the triangles in the numbering the device pipeline relies on (root = inner node 0,
a subtree over k leaves owns k-1 consecutive inner ids, parent links consistent), for ordinary, degenerate and
coincident inputs; and it must actually be a surface-area tree (cost far below a median split of shuffled input)."""
import ctypes as C

import numpy as np
import pytest

from vk_raytrace_amd import capi

LEAF = 0x80000000
NONE = 0xFFFFFFFF


def build(tri9, which="host"):
    L = capi.lib()
    fn = L.pt_debug_sah_topology if which == "host" else L.pt_debug_sahdev_topology   # "device": pt_sahdev.h's kernels' bodies emulated on the host
    fn.restype = C.c_int
    fn.argtypes = [C.c_uint32] + [C.c_void_p] * 6
    n = len(tri9)
    tri9 = np.ascontiguousarray(tri9, np.float32)
    out = [np.zeros(n, np.uint32) for _ in range(5)]
    assert fn(n, tri9.ctypes.data, *[o.ctypes.data for o in out]) == 0
    return out


def check_tree(n, vals, cl, cr, pi, pl):
    assert sorted(vals.tolist()) == list(range(n))          # leaf order is a permutation
    assert pi[0] == NONE
    seen_leaf = np.zeros(n, bool)
    seen_inner = np.zeros(n - 1, bool)

    def walk(node):                                         # returns the number of leaves below `node`
        stack, count = [node], 0
        while stack:
            k = stack.pop()
            assert not seen_inner[k]
            seen_inner[k] = True
            for c in (int(cl[k]), int(cr[k])):
                if c & LEAF:
                    i = c & ~LEAF
                    assert i < n and not seen_leaf[i] and pl[i] == k
                    seen_leaf[i] = True
                    count += 1
                else:
                    assert 0 < c < n - 1 and pi[c] == k
                    stack.append(c)
        return count
    assert walk(0) == n
    assert seen_leaf.all() and seen_inner.all()


def boxes_cost(tri9, vals, cl, cr):
    """SAH cost (sum of inner-node surface areas) of the tree."""
    p0, e1, e2 = tri9[:, 0:3], tri9[:, 3:6], tri9[:, 6:9]
    pts = np.stack([p0, p0 + e1, p0 + e2], 1)[vals]
    lo, hi = pts.min(1), pts.max(1)
    n = len(tri9)
    nlo, nhi = np.zeros((n - 1, 3)), np.zeros((n - 1, 3))
    order = []
    stack = [0]
    while stack:
        k = stack.pop()
        order.append(k)
        for c in (int(cl[k]), int(cr[k])):
            if not c & LEAF:
                stack.append(c)
    for k in reversed(order):
        b = []
        for c in (int(cl[k]), int(cr[k])):
            b.append((lo[c & ~LEAF], hi[c & ~LEAF]) if c & LEAF else (nlo[c], nhi[c]))
        nlo[k], nhi[k] = np.minimum(b[0][0], b[1][0]), np.maximum(b[0][1], b[1][1])
    d = nhi - nlo
    return float((d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]).sum())


@pytest.mark.parametrize("which", ["host", "device"])
@pytest.mark.parametrize("n", [2, 3, 7, 12, 13, 14, 64, 5000, 70000])
def test_random_soup(n, which):
    rng = np.random.default_rng(n)
    tri9 = np.concatenate([rng.uniform(-10, 10, (n, 3)), rng.normal(0, 0.3, (n, 6))], 1).astype(np.float32)
    check_tree(n, *build(tri9, which))


@pytest.mark.parametrize("which", ["host", "device"])
def test_coincident_and_degenerate(which):
    n = 300
    tri9 = np.zeros((n, 9), np.float32)                     # every triangle is the same point
    check_tree(n, *build(tri9, which))
    tri9[:, 3] = 1.0                                        # identical segments
    check_tree(n, *build(tri9, which))
    tri9[:150, 0] = np.linspace(0, 1, 150)                  # half distinct, half coincident
    check_tree(n, *build(tri9, which))


@pytest.mark.parametrize("which", ["host", "device"])
def test_non_finite_vertices_do_not_break_the_builder(which):
    rng = np.random.default_rng(2)
    n = 500
    tri9 = np.concatenate([rng.uniform(-5, 5, (n, 3)), rng.normal(0, 0.3, (n, 6))], 1).astype(np.float32)
    tri9[::7, 0] = np.nan
    tri9[3::11, 4] = np.inf
    tri9[5::13, 8] = -np.inf
    check_tree(n, *build(tri9, which))


def test_device_builder_reaches_the_host_builders_quality():
    """Same algorithm (32 bins, exact sweeps <= 12, one triangle per leaf) as data-parallel passes: the SAH cost of the tree must equal the
    host builder's up to what differs by design (degenerate ranges are halved by position instead of by nth_element)."""
    rng = np.random.default_rng(5)
    for n, spread in ((4096, 0.05), (30000, 0.3)):
        tri9 = np.concatenate([rng.uniform(-10, 10, (n, 3)), rng.normal(0, spread, (n, 6))], 1).astype(np.float32)
        h, d = build(tri9, "host"), build(tri9, "device")
        ch, cd = boxes_cost(tri9, h[0], h[1], h[2]), boxes_cost(tri9, d[0], d[1], d[2])
        assert abs(cd - ch) <= 1e-3 * ch, (n, ch, cd)
    # a mesh-like input: a tessellated sheet plus clutter
    u, v = np.meshgrid(np.linspace(0, 10, 80), np.linspace(0, 10, 80))
    p0 = np.stack([u.ravel(), np.sin(u.ravel()) * 0.5, v.ravel()], 1)
    tri9 = np.concatenate([p0, np.tile([0.13, 0, 0], (len(p0), 1)), np.tile([0, 0.02, 0.13], (len(p0), 1))], 1).astype(np.float32)
    h, d = build(tri9, "host"), build(tri9, "device")
    ch, cd = boxes_cost(tri9, h[0], h[1], h[2]), boxes_cost(tri9, d[0], d[1], d[2])
    assert abs(cd - ch) <= 1e-3 * ch, (ch, cd)


def test_surface_area_quality():
    rng = np.random.default_rng(1)
    n = 4096
    c = rng.uniform(-10, 10, (n, 3))
    tri9 = np.concatenate([c, rng.normal(0, 0.05, (n, 6))], 1).astype(np.float32)
    vals, cl, cr, pi, pl = build(tri9)
    cost = boxes_cost(tri9, vals, cl, cr)
    # a balanced tree over RANDOM order (no spatial sorting at all) for comparison
    m = np.arange(n)
    cl2, cr2 = np.zeros(n - 1, np.uint32), np.zeros(n - 1, np.uint32)

    def bal(first, count, node):
        nl = count // 2
        for side, (f, k, idn) in enumerate(((first, nl, node + 1), (first + nl, count - nl, node + nl))):
            ref = (f | LEAF) if k == 1 else idn
            (cl2 if side == 0 else cr2)[node] = ref
            if k > 1:
                bal(f, k, idn)
    bal(0, n, 0)
    assert cost < 0.05 * boxes_cost(tri9, m, cl2, cr2)
