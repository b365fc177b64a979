"""Mints tests/golden/ref_golden.npz FROM THE REFERENCE ITSELF: every array is an output of oracle/_ref/libref.so, i.e. of the
reference's own sources (shaders/*.glsl, post.frag, src/hdr_sampling.cpp, compress.glsl's host branch) compiled by the committed recipe
oracle/ref_glue/.  Needs /root/reference (this container); the fixture file travels to the GPU box, where tests/test_golden.py holds the
oracle AND the HIP path to it bit for bit.

What the Vulkan driver supplies to the shaders (triangle candidates in trace-contract order, instance matrices, bilinear texels) comes from
the oracle through RefHooks -- see oracle/ref_glue/ref_driver.h; the frames are the reference's shader arithmetic on those inputs.

Run:  python tests/golden/gen_ref_golden.py     (deterministic; rewrites ref_golden.npz)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import ref  # noqa: E402
from tests.common import Config  # noqa: E402
from tests.test_oracle_vs_ref import unit_vectors, bsdf_inputs, sunsky_variants, hdr_image, tonemapper, TM_CASES  # noqa: E402
from vk_raytrace_amd import host_device as hd, synth, workloads  # noqa: E402


def frame_configs():
    """name -> (Config, frames); small enough that the whole fixture stays under a few MB"""
    env = synth.procedural_sky(128, 64)
    out = {}
    wl = workloads.c1_quad()
    out["c1_quad"] = (Config(wl.scene, wl.env, 64, 64, depth=wl.depth, pbr=wl.pbr_mode), 2)
    for mode in (1, 2, 4, 7, 8, 9, 10, 11):
        out[f"fbox_debug{mode}"] = (Config(synth.feature_box(tex_size=32), env, 48, 36, debug=mode, depth=3), 1)
    out["fbox_disney"] = (Config(synth.feature_box(tex_size=32), env, 64, 48, pbr=0), 3)
    out["fbox_gltf"] = (Config(synth.feature_box(tex_size=32), env, 64, 48, pbr=1), 3)
    out["fbox_lights"] = (Config(synth.feature_box(tex_size=32, lights=True), env, 64, 48), 2)
    ss = hd.default_sun_and_sky()
    ss.in_use = 1
    out["fbox_sunsky"] = (Config(synth.feature_box(tex_size=32), env, 64, 48, sunsky=ss), 2)
    out["fbox_samples3"] = (Config(synth.feature_box(tex_size=32), env, 48, 36, max_samples=3, hdr_multiplier=2.5), 2)
    sc = synth.feature_box(tex_size=32)
    sc.camera.aperture, sc.camera.focal_dist = 0.05, 3.0
    out["fbox_dof_clamp"] = (Config(sc, env, 48, 36, firefly=0.5), 2)
    out["fuzz1"] = (Config(synth.fuzz_scene(1), env, 64, 48, depth=6, pbr=1), 2)
    out["fuzz2"] = (Config(synth.fuzz_scene(2), env, 64, 48, depth=6, pbr=0), 2)
    out["fbox_rtx"] = (Config(synth.feature_box(tex_size=32), env, 64, 48, depth=6, max_samples=2, variant=1), 2)       # pathtrace.rgen + hit / miss stages
    out["fbox_rtx_no_anyhit"] = (Config(synth.feature_box(tex_size=32), env, 64, 48, depth=6, variant=1, any_hit=False), 2)  # RtxPipeline::useAnyHit(false)
    out["fbox_rq_no_anyhit"] = (Config(synth.feature_box(tex_size=32), env, 64, 48, depth=6, any_hit=False), 2)
    sp = synth.sponza_like(target_tris=20000, tex_size=64)
    out["sponza_small"] = (Config(sp, env, 64, 36, depth=8), 2)
    return out


def main():
    R = ref.lib()
    g = {}
    rng = np.random.default_rng(2024)
    # ---- integer known answers (random.glsl, compress.glsl, common.glsl)
    a, b = rng.integers(0, 2 ** 32, 256, dtype=np.uint64).astype(np.uint32), rng.integers(0, 2 ** 32, 256, dtype=np.uint64).astype(np.uint32)
    g["tea_a"], g["tea_b"] = a, b
    g["tea_out"] = np.array([R.ref_tea(int(x), int(y)) for x, y in zip(a, b)], np.uint32)
    seeds = rng.integers(0, 2 ** 32, 8, dtype=np.uint64).astype(np.uint32)
    words, floats, final = np.zeros((8, 32), np.uint32), np.zeros((8, 32), np.float32), np.zeros(8, np.uint32)
    for i, s in enumerate(seeds):
        st = C.c_uint32()
        R.ref_pcg_stream(int(s), 32, words[i].ctypes.data, floats[i].ctypes.data, C.byref(st))
        final[i] = st.value
    g["pcg_seed"], g["pcg_words"], g["pcg_floats"], g["pcg_final"] = seeds, words, floats, final
    v3 = rng.integers(0, 2 ** 32, (64, 3), dtype=np.uint64).astype(np.uint32)
    o3 = v3.copy()
    for row in o3:
        R.ref_pcg3d(row.ctypes.data)
    g["pcg3d_in"], g["pcg3d_out"] = v3, o3
    vec = unit_vectors(504, 31)
    g["oct_in"] = vec
    g["oct_packed"] = np.array([R.ref_compress_unit_vec(v.ctypes.data) for v in vec], np.uint32)
    g["oct_packed_host"] = np.array([R.ref_host_compress_unit_vec(v.ctypes.data) for v in vec], np.uint32)
    dec = np.zeros((len(vec), 3), np.float32)
    for p, d in zip(g["oct_packed"], dec):
        R.ref_decompress_unit_vec(int(p), d.ctypes.data)
    g["oct_unpacked"] = dec
    pts = np.concatenate([rng.normal(size=(200, 3)) * 10.0 ** rng.uniform(-4, 3, (200, 1)), rng.uniform(-1 / 32, 1 / 32, (55, 3)), np.zeros((1, 3))]).astype(np.float32)
    nrm = unit_vectors(len(pts) - 8, 32)
    offs, suv = np.zeros((len(pts), 3), np.float32), np.zeros((len(pts), 2), np.float32)
    for p, n, o, u in zip(np.ascontiguousarray(pts), nrm, offs, suv):
        R.ref_offset_ray(p.ctypes.data, n.ctypes.data, o.ctypes.data)
        R.ref_spherical_uv(n.ctypes.data, u.ctypes.data)
    g["offs_p"], g["offs_n"], g["offs_out"], g["spherical_uv"] = pts, nrm, offs, suv
    col = np.concatenate([rng.uniform(-0.2, 1.2, (64, 4)), (np.arange(0, 256, 4)[:, None] + np.array([0.5, 0.49999, 0.50001, 0.0])) / 255.0]).astype(np.float32)
    g["unorm_in"] = col
    g["unorm_out"] = np.array([R.ref_host_pack_unorm4x8(c.ctypes.data) for c in np.ascontiguousarray(col)], np.uint32)
    # ---- sun & sky (sun_and_sky.glsl)
    dirs = unit_vectors(248, 33)
    g["sky_dirs"] = dirs
    for k, ss in enumerate(sunsky_variants()):
        out = np.zeros((len(dirs), 3), np.float32)
        for d, o in zip(dirs, out):
            R.ref_sun_and_sky(C.byref(ss), d.ctypes.data, o.ctypes.data)
        g[f"sky_{k}"] = out
    # ---- BSDFs (pbr_disney.glsl, pbr_gltf.glsl): inputs are regenerated by tests.test_oracle_vs_ref.bsdf_inputs(seed)
    for pbr in (0, 1):
        rows = []
        for m, N, T, B, eta, thin, V, L, seed in bsdf_inputs(800, 40 + pbr):
            f, pdf = np.zeros(3, np.float32), np.zeros(1, np.float32)
            R.ref_bsdf_eval(pbr, m.ctypes.data, N.ctypes.data, T.ctypes.data, B.ctypes.data, eta, thin, V.ctypes.data, L.ctypes.data, f.ctypes.data, pdf.ctypes.data)
            s = C.c_uint32(seed)
            l2, f2, pdf2 = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(1, np.float32)
            R.ref_bsdf_sample(pbr, m.ctypes.data, N.ctypes.data, T.ctypes.data, B.ctypes.data, eta, thin, V.ctypes.data, C.byref(s), l2.ctypes.data, f2.ctypes.data, pdf2.ctypes.data)
            rows.append(np.concatenate([f, pdf, l2, f2, pdf2, np.array([s.value], np.uint32).view(np.float32)]))
        g[f"bsdf_{pbr}"] = np.array(rows, np.float32)
    # ---- environment alias table (src/hdr_sampling.cpp)
    env = np.ascontiguousarray(synth.procedural_sky(32, 16), np.float32)
    acc = np.zeros(32 * 16, hd.envaccel_dtype)
    i, av = C.c_float(), C.c_float()
    R.ref_env_accel(env.ctypes.data, 32, 16, acc.ctypes.data, C.byref(i), C.byref(av))
    g["envaccel_table"] = acc.view(np.uint32).reshape(-1, 4)
    g["envaccel_integral_average"] = np.array([i.value, av.value], np.float32)
    # ---- whole frames (pathtrace.comp dispatched over the image)
    for name, (cfg, frames) in frame_configs().items():
        g["frame_" + name] = ref.render_reference(cfg, frames)
    # ---- display pass (post.frag on RenderOutput's mip chain)
    for k, case in enumerate(TM_CASES):
        img = hdr_image(75, 41, 2 + k)
        out = np.zeros((41, 75, 4), np.float32)
        R.ref_tonemap(C.byref(tonemapper(**case)), img.ctypes.data, 75, 41, out.ctypes.data)
        g[f"post_{k}"] = out
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_golden.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path}: {len(g)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
