"""The CPU oracle held to the reference's OWN code, bit for bit.

oracle/_ref/libref.so is built from /root/reference by the committed recipe oracle/ref_glue/ (shaders/pathtrace.comp with everything it
includes and shaders/post.frag through a lexical GLSL->C++ rewrite; src/hdr_sampling.cpp and the host branch of shaders/compress.glsl
unmodified).  These tests need the reference tree (or a prebuilt libref.so) and skip otherwise; tests/test_golden.py holds the oracle and the
product to fixtures generated from the same library, which travel everywhere.

Bar: BIT-EXACT, floating point included -- both sides evaluate GLSL's implementation-defined pieces (built-in association, transcendental
functions, triangle candidate order, bilinear filtering) identically, so any difference is a misreading of the reference in the oracle.
"""
import ctypes as C

import numpy as np
import pytest

from tests import orc, ref
from tests.common import Config, render_oracle
from vk_raytrace_amd import capi, host_device as hd, synth, workloads

pytestmark = pytest.mark.skipif(not ref.available(), reason="needs /root/reference (or a prebuilt oracle/_ref/libref.so)")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    an, bn = np.isnan(a), np.isnan(b)
    assert np.array_equal(an, bn), what
    assert np.array_equal(bits(np.where(an, 0, a)), bits(np.where(bn, 0, b))), f"{what}: {np.count_nonzero(bits(np.where(an, 0, a)) != bits(np.where(bn, 0, b)))} values differ"


# ---- integer known answers ----------------------------------------------------------------------------------------------------------
def test_rng():
    R, O = ref.lib(), orc.lib()
    rng = np.random.default_rng(3)
    for a, b in rng.integers(0, 2 ** 32, (500, 2), dtype=np.uint64):
        assert R.ref_tea(int(a), int(b)) == O.orc_tea(int(a), int(b))
    for seed in rng.integers(0, 2 ** 32, 16, dtype=np.uint64):
        out = []
        for L, fn in ((R, "ref_pcg_stream"), (O, "orc_pcg_stream")):
            w, f, s = np.zeros(64, np.uint32), np.zeros(64, np.float32), C.c_uint32()
            getattr(L, fn)(int(seed), 64, w.ctypes.data, f.ctypes.data, C.byref(s))
            out.append((w, f.view(np.uint32), s.value))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    for v in rng.integers(0, 2 ** 32, (200, 3), dtype=np.uint64).astype(np.uint32):
        a, b = v.copy(), v.copy()
        R.ref_pcg3d(a.ctypes.data)
        O.orc_pcg3d(b.ctypes.data)
        assert np.array_equal(a, b)


def unit_vectors(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    axes = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [0.70710678, 0.70710678, 0], [0, -0.70710678, -0.70710678]], np.float32)
    return np.ascontiguousarray(np.concatenate([v, axes]), np.float32)


def test_compress_decompress_device_and_host_flavours():
    R, O, P = ref.lib(), orc.lib(), capi.lib()
    for v in unit_vectors(3000, 4):
        p = v.ctypes.data
        dev, host = R.ref_compress_unit_vec(p), R.ref_host_compress_unit_vec(p)
        assert dev == host == O.orc_compress_unit_vec(p) == P.pt_compress_unit_vec(p)
        a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
        R.ref_decompress_unit_vec(dev, a.ctypes.data)
        O.orc_decompress_unit_vec(dev, b.ctypes.data)
        same(a, b)
    inf = np.array([np.inf, 0, 0], np.float32)
    assert R.ref_compress_unit_vec(inf.ctypes.data) == R.ref_host_compress_unit_vec(inf.ctypes.data) == 0xFFFFFFFF


def test_host_pack_unorm_and_vertex_packing():
    R = ref.lib()
    R.ref_host_pack_unorm4x8.restype = C.c_uint32
    rng = np.random.default_rng(5)
    col = np.concatenate([rng.uniform(-0.2, 1.2, (500, 4)), (np.arange(0, 256)[:, None] + np.array([0.5, 0.49999, 0.50001, 0.0])) / 255.0]).astype(np.float32)
    n = len(col)
    want = np.array([R.ref_host_pack_unorm4x8(c.ctypes.data) for c in np.ascontiguousarray(col)], np.uint32)
    pos, nrm = rng.normal(size=(n, 3)).astype(np.float32), unit_vectors(n - 8, 6)
    tan = np.concatenate([unit_vectors(n - 8, 7), np.where(rng.random(n) < 0.5, -1, 1)[:, None]], 1).astype(np.float32)
    uv = rng.random((n, 2)).astype(np.float32)
    for pack in (orc.pack_vertices, capi.pack_vertices):
        assert np.array_equal(pack(pos, nrm, tan, uv, col)["color"], want)


def test_offset_ray_and_common_helpers():
    R, O = ref.lib(), orc.lib()
    O.orc_spherical_uv.argtypes = [C.c_void_p] * 2
    O.orc_coordinate_system.argtypes = [C.c_void_p] * 3
    rng = np.random.default_rng(8)
    pts = np.concatenate([rng.normal(size=(500, 3)) * 10.0 ** rng.uniform(-4, 3, (500, 1)), rng.uniform(-1 / 32, 1 / 32, (100, 3)), np.zeros((1, 3))]).astype(np.float32)
    nrm = unit_vectors(len(pts) - 8, 9)
    for p, n in zip(np.ascontiguousarray(pts), nrm):
        a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
        R.ref_offset_ray(p.ctypes.data, n.ctypes.data, a.ctypes.data)
        O.orc_offset_ray(p.ctypes.data, n.ctypes.data, b.ctypes.data)
        same(a, b, "OffsetRay")
        a, b = np.zeros(2, np.float32), np.zeros(2, np.float32)
        R.ref_spherical_uv(n.ctypes.data, a.ctypes.data)
        O.orc_spherical_uv(n.ctypes.data, b.ctypes.data)
        same(a, b, "GetSphericalUv")
        t0, b0, t1, b1 = (np.zeros(3, np.float32) for _ in range(4))
        R.ref_coordinate_system(n.ctypes.data, t0.ctypes.data, b0.ctypes.data)
        O.orc_coordinate_system(n.ctypes.data, t1.ctypes.data, b1.ctypes.data)
        same(t0, t1, "CreateCoordinateSystem")
        same(b0, b1, "CreateCoordinateSystem")


def test_heatmap_palette():
    R, O = ref.lib(), orc.lib()
    O.orc_temperature.argtypes = [C.c_float, C.c_void_p]
    for x in np.concatenate([np.linspace(-0.5, 1.5, 2001), np.random.default_rng(2).random(2000)]).astype(np.float32):
        a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
        R.ref_temperature(float(x), a.ctypes.data)
        O.orc_temperature(float(x), b.ctypes.data)
        same(a, b, "temperature")


def test_punctual_attenuation():
    R, O = ref.lib(), orc.lib()
    for L, p in ((O, "orc"), ):
        getattr(L, p + "_range_attenuation").argtypes = [C.c_float, C.c_float]
        getattr(L, p + "_range_attenuation").restype = C.c_float
        getattr(L, p + "_spot_attenuation").argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        getattr(L, p + "_spot_attenuation").restype = C.c_float
    rng = np.random.default_rng(10)
    for _ in range(2000):
        rg, d = float(rng.choice([-1.0, 0.0, rng.uniform(0.1, 50)])), float(rng.uniform(0.01, 60))
        same(np.float32(R.ref_range_attenuation(rg, d)), np.float32(O.orc_range_attenuation(rg, d)), "getRangeAttenuation")
        a, b = (np.ascontiguousarray(rng.normal(size=3), np.float32) for _ in range(2))
        oc = float(rng.uniform(-1, 1))
        ic = float(rng.uniform(oc, 1))
        same(np.float32(R.ref_spot_attenuation(a.ctypes.data, b.ctypes.data, oc, ic)), np.float32(O.orc_spot_attenuation(a.ctypes.data, b.ctypes.data, oc, ic)), "getSpotAttenuation")


def sunsky_variants():
    out = []
    for k in range(6):
        ss = hd.default_sun_and_sky()
        ss.in_use = 1
        if k == 1:
            ss.haze, ss.redblueshift, ss.saturation = 3.0, 0.3, 1.4
        if k == 2:
            ss.sun_direction[0], ss.sun_direction[1], ss.sun_direction[2] = 0.9363, -0.1, 0.3366  # below the horizon
        if k == 3:
            ss.y_is_up, ss.physically_scaled_sun = 0, 1
        if k == 4:
            ss.horizon_height, ss.horizon_blur, ss.sun_disk_scale, ss.sun_glow_intensity = 0.3, 0.5, 4.0, 2.5
        if k == 5:
            ss.multiplier, ss.sun_disk_intensity = 0.25, 0.0
        out.append(ss)
    return out


def test_sun_and_sky():
    R, O = ref.lib(), orc.lib()
    dirs = unit_vectors(1500, 12)
    for ss in sunsky_variants():
        for d in dirs:
            a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
            R.ref_sun_and_sky(C.byref(ss), d.ctypes.data, a.ctypes.data)
            O.orc_sun_and_sky(C.byref(ss), d.ctypes.data, b.ctypes.data)
            same(a, b, "sun_and_sky")


def bsdf_inputs(n, seed):
    """random shading states covering every lobe of both BSDFs (metal, transmission incl. thin-walled, clearcoat, sheen, anisotropy, subsurface)"""
    rng = np.random.default_rng(seed)
    for _ in range(n):
        m = np.zeros(22, np.float32)
        m[0:3] = rng.uniform(0.02, 1, 3)
        m[3] = 0.5
        m[4] = rng.choice([0.0, rng.uniform(0, 0.95)])
        m[5] = rng.choice([0.0, 1.0, rng.uniform()])
        m[6] = max(0.001, rng.choice([rng.uniform(), 0.001, 1.0]))
        m[7] = rng.choice([0.0, rng.uniform()])
        m[8] = rng.uniform()
        m[9] = rng.choice([0.0, rng.uniform()])
        m[10:13] = rng.uniform(0, 1, 3)
        m[13] = rng.choice([0.0, rng.uniform()])
        m[14] = max(0.001, rng.uniform())
        m[15] = rng.choice([0.0, 1.0, rng.uniform()])
        m[16] = rng.choice([1.5, 1.0, rng.uniform(1.05, 2.4)])
        aspect = np.sqrt(np.float32(1.0) - m[4] * np.float32(0.9))
        m[17], m[18] = max(0.001, m[6] / aspect), max(0.001, m[6] * aspect)
        dsp = ((m[16] - 1) / (m[16] + 1)) ** 2
        m[19:22] = dsp * (1 - m[5]) + m[0:3] * m[5]
        N = unit_vectors(1, int(rng.integers(1 << 30)))[0]
        t = np.cross(N, unit_vectors(1, int(rng.integers(1 << 30)))[0]).astype(np.float32)
        T = (t / np.linalg.norm(t)).astype(np.float32)
        B = np.cross(N, T).astype(np.float32)
        V, L = unit_vectors(2, int(rng.integers(1 << 30)))[:2]
        if rng.random() < 0.7 and np.dot(V, N) < 0:
            V = -V
        inside = rng.random() < 0.3
        eta = np.float32(m[16] if inside else 1.0 / m[16])
        yield m, N, T, B, float(eta), int(rng.random() < 0.25), np.ascontiguousarray(V), np.ascontiguousarray(L), int(rng.integers(1 << 32))


@pytest.mark.parametrize("pbr", [0, 1])
def test_bsdf_eval_and_sample(pbr):
    R, O = ref.lib(), orc.lib()
    P = C.c_void_p
    O.orc_bsdf_eval.argtypes = [C.c_int, P, P, P, P, C.c_float, C.c_int, P, P, P, P]
    O.orc_bsdf_sample.argtypes = [C.c_int, P, P, P, P, C.c_float, C.c_int, P, P, P, P, P]
    for m, N, T, B, eta, thin, V, L, seed in bsdf_inputs(6000, 20 + pbr):
        res = []
        for lib, pre in ((R, "ref"), (O, "orc")):
            f, pdf = np.zeros(3, np.float32), np.zeros(1, np.float32)
            getattr(lib, pre + "_bsdf_eval")(pbr, m.ctypes.data, N.ctypes.data, T.ctypes.data, B.ctypes.data, eta, thin, V.ctypes.data, L.ctypes.data, f.ctypes.data, pdf.ctypes.data)
            s = C.c_uint32(seed)
            l2, f2, pdf2 = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(1, np.float32)
            getattr(lib, pre + "_bsdf_sample")(pbr, m.ctypes.data, N.ctypes.data, T.ctypes.data, B.ctypes.data, eta, thin, V.ctypes.data, C.byref(s), l2.ctypes.data, f2.ctypes.data,
                                              pdf2.ctypes.data)
            res.append((f, pdf, l2, f2, pdf2, np.array([s.value], np.uint32).view(np.float32)))
        for a, b, nm in zip(res[0], res[1], ("eval f", "eval pdf", "sample L", "sample f", "sample pdf", "seed")):
            same(a, b, f"pbrMode {pbr} {nm}")


# ---- host: src/hdr_sampling.cpp ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(16, 8), (64, 32), (37, 19)])
def test_env_alias_table(shape):
    w, h = shape
    env = synth.procedural_sky(w, h) if w % 2 == 0 else np.random.default_rng(1).uniform(0, 3, (h, w, 4)).astype(np.float32)
    env = np.ascontiguousarray(env, np.float32)
    R = ref.lib()
    acc = np.zeros(w * h, hd.envaccel_dtype)
    i, a = C.c_float(), C.c_float()
    R.ref_env_accel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    assert R.ref_env_accel(env.ctypes.data, w, h, acc.ctypes.data, C.byref(i), C.byref(a)) == w * h
    acc2 = np.zeros(w * h, hd.envaccel_dtype)
    i2, a2 = C.c_float(), C.c_float()
    orc.lib().orc_build_env_accel(env.ctypes.data, w, h, acc2.ctypes.data, C.byref(i2), C.byref(a2))
    acc3, i3, a3 = capi.build_env_accel(env)
    for other, io, ao in ((acc2, i2.value, a2.value), (acc3, i3, a3)):
        assert acc.tobytes() == other.tobytes()
        assert np.float32(i.value) == np.float32(io) and np.float32(a.value) == np.float32(ao)


# ---- whole frames: pathtrace.comp dispatched over the image -----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def env_small():
    return synth.procedural_sky(128, 64)


def frames_equal(cfg, frames):
    a, b = ref.render_reference(cfg, frames), render_oracle(cfg, frames)
    same(a, b, "frame")
    return a


def test_frame_c1_quad():
    wl = workloads.c1_quad()
    frames_equal(Config(wl.scene, wl.env, 96, 96, depth=wl.depth, pbr=wl.pbr_mode), 2)


@pytest.mark.parametrize("mode", range(1, 12))
def test_frame_debug_modes(env_small, mode):
    frames_equal(Config(synth.feature_box(tex_size=32), env_small, 80, 60, debug=mode, depth=3), 1)


@pytest.mark.parametrize("pbr", [0, 1])
def test_frame_path_traced(env_small, pbr):
    img = frames_equal(Config(synth.feature_box(tex_size=32), env_small, 96, 72, pbr=pbr), 3)
    assert np.isfinite(img).all() and img[..., :3].mean() > 0.1


def test_frame_lights_sunsky_samples_dof(env_small):
    frames_equal(Config(synth.feature_box(tex_size=32, lights=True), env_small, 80, 60), 3)
    ss = hd.default_sun_and_sky()
    ss.in_use = 1
    frames_equal(Config(synth.feature_box(tex_size=32), env_small, 80, 60, sunsky=ss), 3)
    frames_equal(Config(synth.feature_box(tex_size=32), env_small, 64, 48, max_samples=3, hdr_multiplier=2.5), 2)
    sc = synth.feature_box(tex_size=32)
    sc.camera.aperture, sc.camera.focal_dist = 0.05, 3.0
    frames_equal(Config(sc, env_small, 64, 48, firefly=0.5), 2)


def test_frame_rtx_pipeline_flavour(env_small):
    """The reference's second renderer (src/rtx_pipeline.cpp): pathtrace.rgen + pathtrace.rchit / .rahit / .rmiss / pathtraceShadow.rmiss
    compiled from their own sources and driven through an emulated vkCmdTraceRaysKHR -- against the oracle's restatement of its two
    observable differences (seed without the maxSamples factor, pathtrace.rgen:72; shadow-ray any-hit draws on a copy of the seed because
    pathtrace.rahit is handed the payload of location 1, traceray_rtx.glsl:54-55).  And RtxPipeline::useAnyHit(false)."""
    sc = synth.feature_box(tex_size=32)
    rtx = Config(sc, env_small, 80, 60, depth=6, max_samples=2, variant=1)
    a = frames_equal(rtx, 3)
    rq = Config(sc, env_small, 80, 60, depth=6, max_samples=2)
    assert not np.array_equal(a, render_oracle(rq, 3))          # the flavours do differ
    frames_equal(Config(synth.fuzz_scene(1), env_small, 64, 48, depth=6, variant=1), 2)
    off = frames_equal(Config(sc, env_small, 80, 60, depth=6, variant=1, any_hit=False), 2)
    assert not np.array_equal(off, render_oracle(Config(sc, env_small, 80, 60, depth=6, variant=1), 2))
    # the ray-query flavour with every instance forced opaque (what pt_use_any_hit(0) means there) against the compute shader on the same hooks
    frames_equal(Config(sc, env_small, 64, 48, depth=6, any_hit=False), 2)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_frame_fuzz_scenes(env_small, seed):
    frames_equal(Config(synth.fuzz_scene(seed), env_small, 64, 48, depth=6, pbr=seed & 1), 2)


# ---- display pass: shaders/post.frag on the mip chain of RenderOutput::genMipmap ----------------------------------------------------------
def hdr_image(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.float32)
    img[..., :3] = (0.05 + 3.0 * rng.random((h, w, 3)) ** 3 * (1 + 20 * np.exp(-((xx - w * 0.7) ** 2 + (yy - h * 0.3) ** 2) / (0.02 * w * w)))[..., None]).astype(np.float32)
    img[..., 3] = 1.0
    return img


def tonemapper(**kw):
    tm = hd.default_tonemapper()
    for k, v in kw.items():
        if k == "renderingRatio":
            tm.renderingRatio[0], tm.renderingRatio[1] = v
        else:
            setattr(tm, k, v)
    return tm


@pytest.mark.parametrize("shape", [(64, 64), (120, 68), (67, 33), (1, 1), (5, 3)])
def test_mip_chain(shape):
    w, h = shape
    img = hdr_image(w, h, 1)
    R, O = ref.lib(), orc.lib()
    O.orc_mip_chain.argtypes = R.ref_mip_chain.argtypes
    n = R.ref_mip_chain(img.ctypes.data, w, h, -1, None, None, None)
    assert n == O.orc_mip_chain(img.ctypes.data, w, h, -1, None, None, None) == int(np.floor(np.log2(max(w, h)))) + 1
    for lod in range(n):
        res = []
        for L, fn in ((R, "ref_mip_chain"), (O, "orc_mip_chain")):
            ow, oh = C.c_int(), C.c_int()
            getattr(L, fn)(img.ctypes.data, w, h, lod, None, C.byref(ow), C.byref(oh))
            out = np.zeros((oh.value, ow.value, 4), np.float32)
            getattr(L, fn)(img.ctypes.data, w, h, lod, out.ctypes.data, C.byref(ow), C.byref(oh))
            res.append(out)
        assert res[0].shape == res[1].shape == (max(1, h >> lod), max(1, w >> lod), 4)
        same(res[0], res[1], f"mip {lod}")
    if w == h == 64:  # power-of-two: every level is the plain 2x2 box average
        assert np.allclose(res[0][0, 0], img.reshape(-1, 4).mean(0), rtol=1e-5)


TM_CASES = [dict(), dict(dither=1), dict(autoExposure=1), dict(autoExposure=3), dict(autoExposure=3, key=0.3, Ywhite=2.0, dither=1),
            dict(brightness=1.4, contrast=1.3, saturation=0.6, vignette=0.5, avgLum=2.0), dict(autoExposure=1, renderingRatio=(0.8, 0.6), vignette=0.3)]


@pytest.mark.parametrize("case", range(len(TM_CASES)))
@pytest.mark.parametrize("shape", [(96, 64), (75, 41)])
def test_post_frag(case, shape):
    w, h = shape
    img = hdr_image(w, h, 2 + case)
    tm = tonemapper(**TM_CASES[case])
    R, O = ref.lib(), orc.lib()
    O.orc_tonemap_zoom.argtypes = [C.POINTER(hd.Tonemapper), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    a, b = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    R.ref_tonemap(C.byref(tm), img.ctypes.data, w, h, a.ctypes.data)
    assert O.orc_tonemap_zoom(C.byref(tm), img.ctypes.data, w, h, w, h, None, b.ctypes.data) == 0
    same(a, b, "post.frag")
    assert np.isfinite(a).all() and a[..., :3].std() > 0.01


@pytest.mark.parametrize("level", [2, 3])
@pytest.mark.parametrize("ae", [0, 1, 3])
def test_post_frag_descaling(level, ae):
    """While navigating the reference renders W/level x H/level pixels into the corner of the offscreen image and magnifies them with
    Tonemapper.zoom = 1/level (src/sample_example.cpp:378,410-413; shaders/post.frag:101)."""
    W, H = 90, 62
    w, h = W // level, H // level
    small = hdr_image(w, h, 7)
    big = np.zeros((H, W, 4), np.float32)
    big[:h, :w] = small
    tm = tonemapper(zoom=1.0 / level, autoExposure=ae, dither=1)
    R, O = ref.lib(), orc.lib()
    O.orc_tonemap_zoom.argtypes = [C.POINTER(hd.Tonemapper), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    a, b = np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)
    R.ref_tonemap(C.byref(tm), big.ctypes.data, W, H, a.ctypes.data)
    assert O.orc_tonemap_zoom(C.byref(tm), small.ctypes.data, w, h, W, H, None, b.ctypes.data) == 0
    same(a, b, "post.frag with zoom")


def test_multi_frame_entry_points_equal_frame_by_frame():
    """ref_render_frames / orc_render_frames (several frames inside one OpenMP team: what bench.py's cpu_baseline leg times) give exactly the
    frames of the frame-by-frame entry points, and the compiled reference and the oracle agree on them."""
    sc = synth.feature_box(tex_size=16)
    env = synth.procedural_sky(64, 32)
    cfg = Config(sc, env, 96, 64)
    ids = np.arange(96 * 64, dtype=np.uint32)[::3].copy()
    o = orc.Oracle()
    o.set_scene(cfg.scene); integ, _ = o.set_env(cfg.env); o.set_camera(cfg.camera); o.set_sunsky(cfg.sunsky)
    st = cfg.state(integ)
    a = np.zeros((64, 96, 4), np.float32)
    for f in range(5):
        st.frame = f
        o.render_frame(st, a, ids)
    b = np.zeros_like(a)
    o.render_frames(st, 0, 2, b, ids)
    o.render_frames(st, 2, 3, b, ids)
    same(a, b, "orc_render_frames")
    r = ref.Reference(cfg.scene, cfg.env, oracle=o)
    r.set_camera(cfg.camera); r.set_sunsky(cfg.sunsky)
    c = np.zeros_like(a)
    r.render_frames(st, 0, 5, c, ids, threads=3)
    same(a, c, "ref_render_frames")
    o.close()
