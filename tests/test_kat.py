"""Integer / bit-level known answers checked against BOTH the CPU oracle and the product's host helpers.  Bar: bit-exact.

Two independent sources: tests/golden/kat.npz, minted by an independent numpy implementation (tests/golden/gen_kat.py), used here; and
tests/golden/ref_golden.npz, minted from the reference's own compiled code (tests/test_golden.py).  The numpy vectors additionally cover
the handedness-bit packing of src/scene.cpp:232-239, which needs nvh::GltfScene and therefore has no compiled-reference counterpart."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import orc
from vk_raytrace_amd import capi, host_device as hd

KAT = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat.npz"))


def test_tea_oracle():
    L = orc.lib()
    got = np.array([L.orc_tea(int(a), int(b)) for a, b in zip(KAT["tea_a"], KAT["tea_b"])], np.uint32)
    assert np.array_equal(got, KAT["tea_out"])


def test_pcg_stream_oracle():
    L = orc.lib()
    for i, seed in enumerate(KAT["pcg_seed"]):
        words = np.zeros(32, np.uint32)
        floats = np.zeros(32, np.float32)
        final = C.c_uint32()
        L.orc_pcg_stream(int(seed), 32, words.ctypes.data, floats.ctypes.data, C.byref(final))
        assert np.array_equal(words, KAT["pcg_words"][i])
        assert np.array_equal(floats.view(np.uint32), KAT["pcg_floats"][i].view(np.uint32))
        assert final.value == KAT["pcg_final"][i]
    assert (KAT["pcg_floats"] >= 0).all() and (KAT["pcg_floats"] < 1).all()


def test_pcg3d_oracle():
    L = orc.lib()
    for vin, vout in zip(KAT["pcg3d_in"], KAT["pcg3d_out"]):
        v = vin.copy()
        L.orc_pcg3d(v.ctypes.data)
        assert np.array_equal(v, vout)


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_compress_unit_vec(who):
    fn = orc.lib().orc_compress_unit_vec if who == "oracle" else capi.lib().pt_compress_unit_vec
    got = np.array([fn(np.ascontiguousarray(v).ctypes.data) for v in KAT["oct_in"]], np.uint32)
    assert np.array_equal(got, KAT["oct_packed"])
    inf = np.array([np.inf, 0, 0], np.float32)
    assert fn(inf.ctypes.data) == 0xFFFFFFFF


def test_decompress_unit_vec_oracle():
    L = orc.lib()
    for p, pre, src in zip(KAT["oct_packed"], KAT["oct_prenorm"], KAT["oct_in"]):
        out = np.zeros(3, np.float32)
        L.orc_decompress_unit_vec(int(p), out.ctypes.data)
        inv = np.float32(1.0) / np.sqrt((pre[0] * pre[0] + pre[1] * pre[1]) + pre[2] * pre[2], dtype=np.float32)
        assert np.array_equal(out, pre * inv)          # normalize(v) = v * (1/sqrt(dot)) in fp32
        assert np.abs(out - src).max() < 1.5e-4        # 16+16 bit octahedral precision


def test_offset_ray_oracle():
    L = orc.lib()
    for p, n, want in zip(KAT["offs_p"], KAT["offs_n"], KAT["offs_out"]):
        out = np.zeros(3, np.float32)
        L.orc_offset_ray(np.ascontiguousarray(p).ctypes.data, np.ascontiguousarray(n).ctypes.data, out.ctypes.data)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_pack_vertices(who):
    n = len(KAT["hand_v"])
    rng = np.random.default_rng(1)
    pos = rng.normal(size=(n, 3)).astype(np.float32)
    nrm = KAT["oct_in"][:n]
    tan = np.concatenate([KAT["oct_in"][n:2 * n], KAT["hand_w"][:, None]], 1).astype(np.float32)
    uv = np.stack([rng.random(n).astype(np.float32), KAT["hand_v"]], 1)
    col = KAT["unorm_in"]
    pack = orc.pack_vertices if who == "oracle" else capi.pack_vertices
    v = pack(pos, nrm, tan, uv, col)
    assert np.array_equal(v["position"], pos)
    assert np.array_equal(v["normal"], KAT["oct_packed"][:n])
    assert np.array_equal(v["tangent"], KAT["oct_packed"][n:2 * n])
    assert np.array_equal(v["texcoord"][:, 1].view(np.uint32), KAT["hand_out"])
    assert np.array_equal(v["texcoord"][:, 0], uv[:, 0])
    assert np.array_equal(v["color"], KAT["unorm_out"])
