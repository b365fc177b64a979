"""Fixtures minted from the reference itself (tests/golden/ref_golden.npz <- tests/golden/gen_ref_golden.py <- oracle/_ref/libref.so, the
reference's own shader and host sources compiled by oracle/ref_glue/).  The fixture file travels to boxes without /root/reference:

  -m "not gpu": the CPU oracle and the product's host helpers reproduce every array bit for bit;
  -m gpu      : the HIP path through the C ABI reproduces the reference's FRAMES bit for bit -- directly, without the oracle in between.
"""
import ctypes as C
import os

import numpy as np
import pytest

from tests import orc
from tests.common import render_oracle, render_hip
from tests.golden.gen_ref_golden import frame_configs
from tests.test_oracle_vs_ref import bsdf_inputs, sunsky_variants, hdr_image, tonemapper, TM_CASES
from vk_raytrace_amd import capi, host_device as hd, synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz"))
FRAMES = frame_configs()


def same(a, b, what=""):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    an, bn = np.isnan(a), np.isnan(b)
    assert a.shape == b.shape and np.array_equal(an, bn), what
    bad = np.count_nonzero(np.where(an, 0, a).view(np.uint32) != np.where(bn, 0, b).view(np.uint32))
    assert bad == 0, f"{what}: {bad} of {a.size} values differ from the reference (max abs {np.nanmax(np.abs(a - b)):.3e})"


def test_rng_and_packing_oracle():
    L = orc.lib()
    assert np.array_equal(np.array([L.orc_tea(int(a), int(b)) for a, b in zip(G["tea_a"], G["tea_b"])], np.uint32), G["tea_out"])
    for i, seed in enumerate(G["pcg_seed"]):
        w, f, s = np.zeros(32, np.uint32), np.zeros(32, np.float32), C.c_uint32()
        L.orc_pcg_stream(int(seed), 32, w.ctypes.data, f.ctypes.data, C.byref(s))
        assert np.array_equal(w, G["pcg_words"][i]) and np.array_equal(f.view(np.uint32), G["pcg_floats"][i].view(np.uint32)) and s.value == G["pcg_final"][i]
    for vin, vout in zip(G["pcg3d_in"], G["pcg3d_out"]):
        v = vin.copy()
        L.orc_pcg3d(v.ctypes.data)
        assert np.array_equal(v, vout)
    for p, n, want, uv in zip(G["offs_p"], G["offs_n"], G["offs_out"], G["spherical_uv"]):
        out, o2 = np.zeros(3, np.float32), np.zeros(2, np.float32)
        L.orc_offset_ray(np.ascontiguousarray(p).ctypes.data, np.ascontiguousarray(n).ctypes.data, out.ctypes.data)
        L.orc_spherical_uv.argtypes = [C.c_void_p] * 2
        L.orc_spherical_uv(np.ascontiguousarray(n).ctypes.data, o2.ctypes.data)
        same(out, want, "OffsetRay")
        same(o2, uv, "GetSphericalUv")


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_oct_vectors_and_unorm(who):
    assert np.array_equal(G["oct_packed"], G["oct_packed_host"])  # the reference's device and host flavours agree with each other
    fn = orc.lib().orc_compress_unit_vec if who == "oracle" else capi.lib().pt_compress_unit_vec
    assert np.array_equal(np.array([fn(np.ascontiguousarray(v).ctypes.data) for v in G["oct_in"]], np.uint32), G["oct_packed"])
    if who == "oracle":
        for p, want in zip(G["oct_packed"], G["oct_unpacked"]):
            out = np.zeros(3, np.float32)
            orc.lib().orc_decompress_unit_vec(int(p), out.ctypes.data)
            same(out, want, "decompress_unit_vec")
    n = len(G["unorm_in"])
    rng = np.random.default_rng(0)
    pack = orc.pack_vertices if who == "oracle" else capi.pack_vertices
    v = pack(rng.normal(size=(n, 3)).astype(np.float32), G["oct_in"][:n], np.concatenate([G["oct_in"][n:2 * n], np.ones((n, 1), np.float32)], 1), rng.random((n, 2)).astype(np.float32), G["unorm_in"])
    assert np.array_equal(v["color"], G["unorm_out"]) and np.array_equal(v["normal"], G["oct_packed"][:n]) and np.array_equal(v["tangent"], G["oct_packed"][n:2 * n])


def test_sun_and_sky_oracle():
    L = orc.lib()
    for k, ss in enumerate(sunsky_variants()):
        out = np.zeros_like(G[f"sky_{k}"])
        for d, o in zip(np.ascontiguousarray(G["sky_dirs"]), out):
            L.orc_sun_and_sky(C.byref(ss), d.ctypes.data, o.ctypes.data)
        same(out, G[f"sky_{k}"], f"sun_and_sky variant {k}")


@pytest.mark.parametrize("pbr", [0, 1])
def test_bsdf_oracle(pbr):
    L = orc.lib()
    P = C.c_void_p
    L.orc_bsdf_eval.argtypes = [C.c_int, P, P, P, P, C.c_float, C.c_int, P, P, P, P]
    L.orc_bsdf_sample.argtypes = [C.c_int, P, P, P, P, C.c_float, C.c_int, P, P, P, P, P]
    rows = []
    for m, N, T, B, eta, thin, V, Ld, seed in bsdf_inputs(800, 40 + pbr):
        f, pdf = np.zeros(3, np.float32), np.zeros(1, np.float32)
        L.orc_bsdf_eval(pbr, m.ctypes.data, N.ctypes.data, T.ctypes.data, B.ctypes.data, eta, thin, V.ctypes.data, Ld.ctypes.data, f.ctypes.data, pdf.ctypes.data)
        s = C.c_uint32(seed)
        l2, f2, pdf2 = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(1, np.float32)
        L.orc_bsdf_sample(pbr, m.ctypes.data, N.ctypes.data, T.ctypes.data, B.ctypes.data, eta, thin, V.ctypes.data, C.byref(s), l2.ctypes.data, f2.ctypes.data, pdf2.ctypes.data)
        rows.append(np.concatenate([f, pdf, l2, f2, pdf2, np.array([s.value], np.uint32).view(np.float32)]))
    same(np.array(rows, np.float32), G[f"bsdf_{pbr}"], f"BSDF pbrMode {pbr}")


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_env_alias_table(who):
    env = np.ascontiguousarray(synth.procedural_sky(32, 16), np.float32)
    if who == "oracle":
        acc = np.zeros(32 * 16, hd.envaccel_dtype)
        i, a = C.c_float(), C.c_float()
        orc.lib().orc_build_env_accel(env.ctypes.data, 32, 16, acc.ctypes.data, C.byref(i), C.byref(a))
        integral, average = i.value, a.value
    else:
        acc, integral, average = capi.build_env_accel(env)
    assert np.array_equal(acc.view(np.uint32).reshape(-1, 4), G["envaccel_table"])
    same(np.array([integral, average], np.float32), G["envaccel_integral_average"])


@pytest.mark.parametrize("name", sorted(FRAMES))
def test_frames_oracle(name):
    cfg, frames = FRAMES[name]
    same(render_oracle(cfg, frames), G["frame_" + name], name)


@pytest.mark.parametrize("k", range(len(TM_CASES)))
def test_post_frag_oracle(k):
    img = hdr_image(75, 41, 2 + k)
    out = np.zeros((41, 75, 4), np.float32)
    L = orc.lib()
    L.orc_tonemap_zoom.argtypes = [C.POINTER(hd.Tonemapper), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert L.orc_tonemap_zoom(C.byref(tonemapper(**TM_CASES[k])), img.ctypes.data, 75, 41, 75, 41, None, out.ctypes.data) == 0
    same(out, G[f"post_{k}"], f"post.frag case {k}")


# ---- the HIP path against the reference's frames, no oracle in between ------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FRAMES))
def test_frames_hip(name):
    cfg, frames = FRAMES[name]
    same(render_hip(cfg, frames), G["frame_" + name], name)


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(TM_CASES)))
def test_post_frag_hip(k):
    """pt_tonemap on an image restored with pt_write_accum against post.frag's fragColor quantised like the UNORM8 swapchain"""
    from vk_raytrace_amd.renderer import HipRenderer
    img = hdr_image(75, 41, 2 + k)
    r = HipRenderer()
    r.setup(0)
    r.create((75, 41))
    r.write_accum(img)
    got = r.tonemap(tonemapper(**TM_CASES[k]))
    r.destroy()
    want = np.floor(np.clip(G[f"post_{k}"], 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert np.array_equal(got, want)
