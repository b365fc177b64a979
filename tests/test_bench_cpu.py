"""bench.py's host logic that needs no GPU: the CPU leg (cpu_baseline + the per-pixel L2 of the metric's second half) on a small instance of the
bench workload.  The "GPU image" here is the oracle's own render of the same frames, so the parity figures must say "identical" -- and a perturbed
image must be reported as a mismatch (shaders/pathtrace.comp:122-133 is the accumulation the comparison is taken on)."""
import numpy as np

import bench
from tests import orc
from vk_raytrace_amd import capi, host_device as hd, workloads


def _small_workload():
    wl = workloads.c3_sponza(96, 64, 3, tex_size=32, target_tris=3000, env_w=64)
    wl.scene.finalize(capi.pack_vertices)
    cam = capi.camera_lookat(wl.scene.camera, wl.width / wl.height, nb_lights=len(wl.scene.lights))
    o = orc.Oracle()
    o.set_scene(wl.scene)
    integral, _ = o.set_env(wl.env)
    o.set_camera(cam)
    o.set_sunsky(hd.default_sun_and_sky())
    st = hd.default_rtx_state()
    st.size[0], st.size[1] = wl.width, wl.height
    st.maxDepth, st.pbrMode, st.maxSamples = wl.depth, wl.pbr_mode, 1
    st.fireflyClampThreshold = 4.0 * integral
    return wl, cam, integral, o.render(st, 3)


def test_cpu_leg_reports_identical_images_and_flags_a_mismatch():
    wl, cam, integral, img = _small_workload()
    base, par, alg = bench.cpu_leg(wl, cam, integral, wl.width, wl.height, 3, img, cpu_seconds=2.0)
    assert base["kind"] in ("reference", "port") and base["value"] > 0 and base["cores"] >= 1
    assert par["frames"] == 3 and par["pixels"] == wl.width * wl.height  # a small image is compared in full (block stride 1)
    assert par["l2"] == 0.0 and par["max_abs"] == 0.0 and par["pixels_bit_identical"] == par["pixels"]
    for other in par["also"]:
        assert other["l2"] == 0.0
    assert alg["nodes_per_closest_ray"] > 0 and alg["tris_per_closest_ray"] > 0
    bad = img.copy()
    bad[10, 17, 1] += 0.5
    _, par2, _ = bench.cpu_leg(wl, cam, integral, wl.width, wl.height, 3, bad, cpu_seconds=2.0)
    assert par2["l2"] > 0 and par2["pixels_bit_identical"] == par2["pixels"] - 1
    assert abs(par2["max_abs"] - 0.5) < 1e-6
    nan = img.copy()
    nan[3, 3, 0] = np.nan
    _, par3, _ = bench.cpu_leg(wl, cam, integral, wl.width, wl.height, 3, nan, cpu_seconds=2.0)
    assert not (par3["l2"] <= 1e-3)  # a NaN on one side only can never pass


def test_cpu_leg_without_a_gpu_image_is_a_baseline_only():
    wl, cam, integral, _ = _small_workload()
    base, par, alg = bench.cpu_leg(wl, cam, integral, wl.width, wl.height, 2, None, cpu_seconds=1.0)
    assert par is None and base["value"] > 0
