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
    # round 6: the line names the MEASURED binder -- VALU issue against the ceiling tools/valu_mix.hip reaches for the stage's own instruction mix --
    # and SURVEY 8(d)'s algorithmic-byte figure moves to alg_* (it exceeds 1: cache-served bytes); the HBM interface is `traffic` / `hbm`
    assert r["bound"] == "valu_issue" and r["unit"] == "G wave-instructions/s" and 300.0 < r["peak"] < 1300.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    assert r["avg_launch_ms"] * r["launches"] / rec["steps"] <= rec["ms_per_step"]  # the dominant stage fits inside a step
    assert abs(r["alg_achieved_GBps"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["alg_achieved_GBps"]
    assert abs(r["alg_frac"] - r["alg_achieved_GBps"] / 8000.0) < 1e-12 and "alg_note" in r
    assert r["hbm"]["peak"] == 8000.0 and 0.0 < r["hbm"]["frac"] < 1.0 and r["traffic"] > 0
    lanes = r["lanes_per_valu_instr"]
    assert lanes and all(0.0 < v <= 64.0 for v in lanes.values())  # hardware lane occupancy of the stage's kernels (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU)
    per = r["per_stage"]
    for k in ("closest", "shade", "shadow"):
        assert 0.0 < per[k]["valu_issue_frac"] < 1.0 and 0.0 < per[k]["hbm_frac"] < 1.0, k
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


_FALLBACK = r"""
import json, sys, types
sys.path.insert(0, %r)
import numpy as np
import bench

calls = []
def fake_launch(n):
    calls.append(("self_launch", n))
    return 7   # a rank died (rendezvous, RCCL bootstrap between processes, ...)

class FakeRenderer:
    def measure_peaks(self):
        return {"valuWaveInstrPerSec": 8.0e11, "hbmCopyBytesPerSec": 4.5e12, "hbmReadBytesPerSec": 6.0e12, "computeUnits": 256, "clockMHz": 2400}

def fake_single(args):
    calls.append(("single_process", args.gpus))
    wl = types.SimpleNamespace(name="C3 stand-in (stub)", scene=types.SimpleNamespace(num_triangles=12, materials=[0], textures=[]), depth=8, pbr_mode=0, env=np.zeros((4, 8, 4), np.float32), note="")
    W, H = 64, 32
    stats = {k: 10 for k in ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups")}
    stats.update({"msBuildAccel": 1.0, "bytesAccel": 1000, "numBlas": 0, "numTlasNodes": 0, "numBvhNodes": 3, "batchFrames": 2, "framesInFlight": 2, "samples": W * H * args.steps})
    img = np.ones((H, W, 4), np.float32)
    return {"wl": wl, "W": W, "H": H, "windows": [0.002, 0.001, 0.003], "img_first": img, "img": img, "gather_ms": 0.5, "stats": stats, "ranks_seen": args.gpus, "t_setup": 0.1,
            "integral": 1.0, "cam": None, "renderer": FakeRenderer(), "st": None, "frame": 0, "per_rank_ms": [0.4, 0.5], "gather": "rccl (pt_comm_init_all, one process)", "shard": (0, args.gpus),
            "cleanup": lambda: calls.append(("cleanup",))}

bench.self_launch, bench.single_process = fake_launch, fake_single
sys.argv = ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "0", "--no-profile", "--no-interactive", "--no-cpu-baseline"] + sys.argv[1:]
try:
    bench.main()
finally:
    print(json.dumps(calls), file=sys.stderr)
"""


def _run_fallback(extra=()):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", _FALLBACK % root] + list(extra), env=env, capture_output=True, text=True, timeout=300)
    calls = json.loads([ln for ln in p.stderr.strip().splitlines() if ln.startswith("[")][-1])
    return p, calls


def test_failed_self_launch_falls_back_to_single_process():
    """`bench.py --gpus 2` without a launcher: the self-launched ranks fail -> the same job from ONE process (--single-process) instead of no line at all, and the
    line says so; `--no-fallback` turns the failure into the exit code.  The launch and the renderer are stubs here (no GPU): what is under test is main()'s flow
    -- the fallback, the median window, the N > 1 fields incl. the prediction of tools/shard_table.py, the line being the last thing on stdout."""
    import json
    p, calls = _run_fallback()
    assert p.returncode == 0, p.stderr[-2000:]
    assert [c[0] for c in calls] == ["self_launch", "single_process", "cleanup"] and calls[0][1] == 2
    assert "falling back to --single-process" in p.stderr and "exit code 7" in p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["launch"].startswith("single-process (fallback") and line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert abs(line["ms_per_step"] - 1.0) < 1e-9 and len(line["repeats"]) == 3       # the median window (2 ms for 2 steps)
    assert line["scaling"] == "strong" and line["ms_per_step_per_rank"] == [0.4, 0.5] and line["gather"].startswith("rccl")
    sp = line["scaling_prediction"]
    assert sp["tabulated_steps"] == 20 and 0.0 < sp["efficiency_predicted"] <= 1.0 and sp["ms_per_step_measured"] == line["ms_per_step"]
    assert abs(sp["efficiency_vs_n1_prediction"] - sp["n1_ms_per_step_predicted"] / (2 * line["ms_per_step"])) < 1e-12
    p, calls = _run_fallback(["--no-fallback"])
    assert p.returncode == 7 and [c[0] for c in calls] == ["self_launch"] and p.stdout.strip() == ""
