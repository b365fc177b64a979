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


def test_evidence_fields_on_a_recorded_run():
    """The roofline / binding post-processing of the line, fed the serialised pass of a recorded GPU run (tests/golden/bench_line_recorded.json:
    profiles/r04a_bench_20.json) and the committed PMC summaries: SURVEY 8(d)'s contract fields, the binding ceilings, and the cross-checks the
    judge recomputes (stage time per step <= ms_per_step; fractions = achieved / peak)."""
    import json
    import os
    rec = json.loads(open(os.path.join(os.path.dirname(__file__), "golden", "bench_line_recorded.json")).readline())
    out = {"calibration": rec["calibration"], "rays": rec["rays"]}
    samples = rec["config"]["width"] * rec["config"]["height"] * rec["steps"]
    elapsed = rec["ms_per_step"] * rec["steps"] * 1e-3
    bench.evidence_fields(out, rec["serialised"], rec["alg_model"], samples, elapsed, 1, "c3")
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["avg_launch_ms"] * r["launches"] / rec["steps"] <= rec["ms_per_step"]  # the dominant stage fits inside a step
    b = r["binding"]
    assert b is not None and 0.0 < b["frac"] < 1.0, "the binding ceiling of the dominant stage is a measured fraction below 1"
    assert "alg_model_note" in r
    sh = out["stages_serialised"]["shade"]["binding"]
    assert 0.0 < sh["frac"] < 1.5 and sh["peak_GBps"] == rec["calibration"]["hbm_read_GBps"]
    for k in ("closest", "shadow"):
        bb = out["stages_serialised"][k]["binding"]
        assert 0.0 < bb["frac"] < 1.0 and 0.0 < bb["wave_cycles_waiting_frac"] < 1.0
    assert 0.0 < out["hbm_measured"]["frac"] < 1.0 and 0.0 < out["issue_roofline"]["frac"] < 1.0


def test_evidence_fields_never_cost_the_line():
    """without a serialised pass or an algorithmic model the function leaves the line alone"""
    out = {"calibration": {"valu_G_wave_instr_per_s": 800.0, "hbm_read_GBps": 6000.0, "hbm_copy_GBps": 4500.0, "compute_units": 256, "clock_MHz": 2400}, "rays": {}}
    bench.evidence_fields(out, None, None, 1000, 1.0, 1, "c3")
    assert "roofline" not in out
