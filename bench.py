#!/usr/bin/env python
"""Headline benchmark: Msamples/s on the Sponza-class configuration of BASELINE.json (C3: 1920x1080,
depth 8, Disney BSDF, HDR environment; 256 spp == 256 steps).

A "step" is one frame: one sample per pixel over the whole image (maxSamples = 1, the reference default),
accumulated into the RGBA32F framebuffer exactly like shaders/pathtrace.comp:122-133.  Scene, BVH, textures
and environment are resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps 256 --warmup 8
  python bench.py --gpus N ...          (self-launching: spawns one process per GPU on 127.0.0.1 and relays rank 0's line; falls back to --single-process
                                         when those ranks fail to come up)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --gpus N --single-process     (one process, N contexts, pt_comm_init_all)
  python bench.py --from-gltf scene.glb ...     (the scene through libptmi's own importer: data = "gltf")

With N > 1 the image tiles are sharded over the ranks (vk_raytrace_amd/shard.py); the ranks do not communicate while rendering and
the single framebuffer gather -- libptmi's own RCCL path over xGMI, pt_gather_shards / pt_gather_finish -- happens outside the timed loop (its
time is reported as gather_ms, like the reference metric which times the frame loop only - BASELINE.md section 2).  The communicator is built and the
gather route agreed in a PREFLIGHT before anything is timed (RCCL calls under a watchdog; rank 0 prints the RCCL version, the route and every rank's
time per frame to stderr); when RCCL cannot be brought up the shards travel through the control-plane socket instead and the line's `gather` field
says so.  `ranks_seen` is ncclCommCount of the communicator.  Every rank binds its GPU through pt_create BEFORE any rendezvous, so a box with fewer
than N devices fails with pt_create's device-count message.  The control plane (rendezvous of the launcher's ranks, barriers around the timed region,
the MAX of the ranks' times, handing out the ncclUniqueId) is vk_raytrace_amd/rendezvous.py -- a Unix-domain socket between the ranks of this node;
a rank imports nothing of torch, because the PyTorch wheel's bundled HIP / RCCL libraries break the system RCCL that libptmi opens (rendezvous.py).
The image gathered after the FIRST timed window is what the parity leg compares: the line of an N > 1 run carries `parity`, `cpu_baseline` and
`roofline` like the N = 1 line.  The JSON line is the last thing written to stdout.

After the timed region (never part of `value`), rank 0 measures what the JSON line's evidence fields need, all in this run:
  calibration        pt_measure_peaks: the VALU-issue and HBM-streaming ceilings of this box
  interactive        render + tonemap per frame (batch = 1, SampleExample's display loop)
  serialised         one batch on a second context with one frame slot and the SAME launch policy as the timed run (k_tail included):
                     standalone stage durations (HIP events on the launching stream, nothing overlapped)
  roofline           the stage with the largest standalone time against the ceiling that was MEASURED to bind it (round 6): `bound` = "valu_issue", `achieved` =
                     VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU, profiles/rNN_valu.json) / its average launch duration, `peak` = the issue rate
                     tools/valu_mix.hip reaches for that kernel's own instruction mix at its occupancy (profiles/rNN_valu_mix.txt), `lanes_per_valu_instr` =
                     hardware lane occupancy (profiles/rNN_binders.json); `traffic` / `hbm` = HBM bytes per launch from the PMC passes against 8 TB/s;
                     `alg_*` = SURVEY.md 8(d)'s ALGORITHMIC bytes per launch / launch duration vs 8 TB/s (exceeds 1: those bytes are cache-served);
                     `l2_*` = L2 requests of the stage (profiles/rNN_cache.json); `per_stage` = the same fractions for every stage
  hbm_measured       measured HBM bytes per sample x this run's rate
  issue_roofline     VALU wave-instructions per sample (profiles/rNN_valu.json, same PMC run) x this run's rate / the calibrated ceiling
  cpu_baseline       oracle/_ref (the reference's own pathtrace.comp compiled for the host, kind "reference") and the CPU oracle (the
                     restatement, `port_value`) on a bounded sample of the same workload, on rank 0; the same renders are the parity reference
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before torch / HIP initialise (see vk_raytrace_amd/capi.py)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth (MI355X_MICROARCH.md, "L2 (per XCD)")
L2_LINE = 128          # bytes per L2 request (TCC cache line)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=5, help="the timed window of --steps frames is run this many times back to back (each bracketed by a device synchronisation and the ranks' barrier); "
                    "`value` is the median window, every window is printed in `repeats`")
    # development knobs (the defaults are the BASELINE configuration)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--tris", type=int, default=262_267)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"], help="BASELINE.json configuration (stand-in scene); c3 is the bench line, the others are for the results table")
    ap.add_argument("--emulate-shard", default="", help="R/N: render only rank R's tiles of an N-GPU run on this one GPU (scaling estimate; value = this shard's rate)")
    ap.add_argument("--accel", default="flat", choices=["flat", "two"], help="acceleration structure: flat world-space hierarchy (default) or the reference's BLAS per prim-mesh + TLAS (pt_set_accel_mode)")
    ap.add_argument("--refit", type=int, default=0, help="with --accel two: time this many pt_update_instances calls (TLAS refit) after the run")
    ap.add_argument("--single-process", action="store_true", help="one process drives all --gpus N GPUs (N contexts, pt_comm_init_all) instead of one process per GPU; also the automatic "
                    "fallback when the self-launched ranks of --gpus N fail to come up")
    ap.add_argument("--no-fallback", action="store_true", help="with --gpus N: do not fall back to --single-process when the self-launched ranks fail")
    ap.add_argument("--from-gltf", default="", help="time a .glb / .gltf file through libptmi's own importer (pt_gltf_load) instead of building the stand-in scene in memory: data = \"gltf\"")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the serialised profiling pass (roofline fields become null)")
    ap.add_argument("--no-interactive", action="store_true", help="skip the frame-by-frame (render + tonemap) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="target duration of the CPU baseline sample (split between the compiled reference and the port)")
    return ap.parse_args()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: one child per GPU with the environment torch.distributed.run would give it.  Rank 0's
    stdout (the JSON line) is this process's stdout.  A child that fails ends the job: the others are stopped (by PID) and its exit code is ours."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PT_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = set(range(n))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    rc = code
                    print(f"bench.py: rank {r} of {n} exited with code {code}; stopping the other ranks", file=sys.stderr)
                    break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def usable_cores():
    """Host cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota (a container that sees 256 cores through
    the affinity mask but is throttled to a few would otherwise run 256 spinning OpenMP threads on them)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]       # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())       # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except Exception:
            pass
    return n


def mix_ceilings():
    """VALU issue ceilings measured for the kernels' own instruction mixes (tools/valu_mix.hip, output committed as profiles/rNN_valu_mix.txt):
    G wave-instructions/s for the trace-machine mix at 5 waves per SIMD and the k_shade mix at 4."""
    import re
    for rnd in ("r06",):
        p = os.path.join(ROOT, "profiles", f"{rnd}_valu_mix.txt")
        if not os.path.exists(p):
            continue
        txt, out = open(p).read(), {}
        for key, pat in (("trace", r"trace-machine mix \(k_closest_p: 5\)\s+waves/SIMD 5:\s+([0-9.]+) G"), ("packet", r"trace-machine mix \(80 VALU \+ 28 SALU\)\s+waves/SIMD 8:\s+([0-9.]+) G"),
                         ("shade", r"k_shade mix \(k_shade: 4\)\s+waves/SIMD 4:\s+([0-9.]+) G"), ("fmac", r"independent v_fmac_f32\s+waves/SIMD 8:\s+([0-9.]+) G")):
            m = re.search(pat, txt)
            if m:
                out[key] = float(m.group(1))
        if out:
            return out, f"profiles/{rnd}_valu_mix.txt"
    return None, None


def scaling_prediction(workload, steps, world, ms_per_step):
    """The N > 1 line next to what the all-rank emulation on ONE GPU predicted for it (tools/shard_table.py: every rank's shard rendered alone, job time =
    slowest rank; profiles/rNN_shard_table_<workload>.json): predicted ms per step and efficiency for this N at the nearest tabulated step count, and the
    measured time against it.  The N = 1 time of THIS run is not known inside an N > 1 run: the driver computes the measured efficiency from its own lines."""
    table, src = latest_profile("shard_table_" + workload)
    if not table:
        return None
    rows = table.get("rows", [])
    tab_steps = sorted({r["steps"] for r in rows})
    if not tab_steps:
        return None
    near = min(tab_steps, key=lambda t: abs(np.log(t / max(1, steps))))
    one = next((r for r in rows if r["steps"] == near and r["ranks"] == 1), None)
    mine = next((r for r in rows if r["steps"] == near and r["ranks"] == world), None)
    if not one or not mine:
        return None
    return {"source": src, "tabulated_steps": near, "n1_ms_per_step_predicted": one["ms_per_frame_slowest_rank"], "ms_per_step_predicted": mine["ms_per_frame_slowest_rank"],
            "efficiency_predicted": mine["efficiency"], "imbalance_predicted": mine["imbalance_max_over_mean"], "ms_per_step_measured": ms_per_step,
            "measured_over_predicted": ms_per_step / mine["ms_per_frame_slowest_rank"],
            "efficiency_vs_n1_prediction": one["ms_per_frame_slowest_rank"] / (world * ms_per_step),
            "note": "prediction = every rank's shard rendered alone on one GPU (no host sharing, no gather); efficiency_vs_n1_prediction = the table's N = 1 time / (N x this run's time)"}


def latest_profile(kind, workload="c3"):
    """profiles/rNN_<kind>.json of the newest round that has one (this round's PMC passes, else the previous round's); the other BASELINE configurations
    have summaries of their own, profiles/rNN_<kind>_<workload>.json (per-sample figures are a property of code AND workload)."""
    suffix = "" if workload == "c3" else f"_{workload}"
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", f"{rnd}_{kind}{suffix}.json")
        if os.path.exists(p):
            try:
                return json.load(open(p)), f"profiles/{rnd}_{kind}.json"
            except Exception:
                pass
    return None, None


def cpu_leg(wl, cam, integral, W, H, frames_first, img_first, cpu_seconds):
    """The CPU side of the line, N = 1 only: oracle/_ref (the reference's own pathtrace.comp compiled for the host, kind "reference") and the CPU oracle
    (the restatement, kind "port") render frames 0 .. frames_first-1 of a bounded pixel sample of the SAME workload.  Those renders are both the timing
    sample of `cpu_baseline` and -- when img_first (the GPU accumulation image of the same frames) is given -- the parity reference.
    Returns (cpu_baseline, parity | None, alg): alg = per-ray node / triangle counts of the oracle's BVH2 for the SURVEY 8(d) algorithmic bytes."""
    from vk_raytrace_amd import host_device as hd
    out = {}
    from tests import orc
    o = orc.Oracle()
    o.set_scene(wl.scene)
    o.set_env(wl.env)
    o.set_camera(cam)
    o.set_sunsky(hd.default_sun_and_sky())
    # bounded sample of the same workload: every s-th 8x8 pixel block, frames 0 .. F-1 with F = warmup + steps -- exactly the frames the GPU image
    # read after the first timed window holds, so the SAME CPU renders are the baseline's timing sample AND the parity reference
    # (BASELINE.json metric, second half: per-pixel L2 on the linear accumulation image, shaders/pathtrace.comp:122-133).  The block stride s
    # is chosen so that one leg stays near --cpu-seconds / 2 at ~4 Msamples/s.
    F = frames_first
    bx, by = (W + 7) // 8, (H + 7) // 8
    budget_samples = 4.0e6 * max(1.0, cpu_seconds / 2)
    stride = max(1, int(np.ceil(bx * by * 64.0 * F / budget_samples)))
    blocks = np.arange(bx * by)[::stride]
    xs = (blocks % bx)[:, None, None] * 8 + np.arange(8)[None, None, :]
    ys = (blocks // bx)[:, None, None] * 8 + np.arange(8)[None, :, None]
    ok = (xs < W) & (ys < H)
    ids = (ys * W + xs)[np.broadcast_to(ok, (len(blocks), 8, 8))].astype(np.uint32)
    ost = hd.default_rtx_state()
    ost.size[0], ost.size[1] = W, H
    ost.maxDepth, ost.pbrMode, ost.maxSamples = wl.depth, wl.pbr_mode, 1
    ost.fireflyClampThreshold = 4.0 * integral
    # threads: what the scheduler / cgroup quota say, then a short probe (a box may show 256 cores and deliver a fraction: spinning
    # OpenMP threads on throttled cores are slower than fewer threads) -- the fastest of {all, 64, 32, 16, 8} on two frames of the sample
    cores, best_rate = usable_cores(), 0.0
    probe_acc = np.zeros((H, W, 4), np.float32)
    o.L.orc_set_threads(o.ctx, cores)
    o.render_frames(ost, 0, 1, probe_acc, ids)  # (lazy BVH build of the oracle: not part of any timing)
    for cand in sorted({c for c in (cores, 64, 32, 16, 8) if 1 <= c <= cores}, reverse=True):
        o.L.orc_set_threads(o.ctx, cand)
        t0 = time.perf_counter()
        o.render_frames(ost, 1, 2, probe_acc, ids)
        rate = 2 * len(ids) / (time.perf_counter() - t0)
        if rate > best_rate * 1.05:
            best_rate, threads = rate, cand
    cores = threads
    o.L.orc_set_threads(o.ctx, cores)
    del probe_acc

    def timed(render_frames):
        """frames 0 .. F-1 of the sample in ONE call = one thread team (no fork / join per frame); BVH build and page faults happened in the probe"""
        acc = np.zeros((H, W, 4), np.float32)
        t0 = time.perf_counter()
        render_frames(ost, 0, F, acc, ids)
        dt = time.perf_counter() - t0
        return len(ids) * F / dt / 1e6, dt, acc

    def parity_against(acc, name):
        """per-pixel L2 (sqrt(mean over pixels and RGB of (a - b)^2)) of the GPU accumulation image against a CPU render of the same frames"""
        py_, px_ = np.divmod(ids.astype(np.int64), W)
        a = img_first[py_, px_, :3].astype(np.float64)
        b = acc[py_, px_, :3].astype(np.float64)
        both_nan = np.isnan(a) & np.isnan(b)  # the reference's own NaN pixels are equal when they are NaN on both sides
        d = np.where(both_nan, 0.0, np.nan_to_num(a - b, nan=np.inf))  # a NaN on one side only is a mismatch
        bits_equal = int(np.count_nonzero(np.all((img_first[py_, px_, :3].view(np.uint32) == acc[py_, px_, :3].view(np.uint32)) | both_nan, axis=-1)))
        return {"against": name, "l2": float(np.sqrt(np.mean(d * d))), "max_abs": float(np.max(np.abs(d))), "pixels": int(len(ids)), "pixels_bit_identical": bits_equal,
                "frames": int(F), "nan_pixels": int(np.count_nonzero(np.any(both_nan, axis=-1)))}

    port_v, port_t, port_acc = timed(o.render_frames)
    os_ = o.stats()
    sample_txt = f"one 8x8 pixel block in {stride} of the same {W}x{H} workload ({len(ids)} pixels), frames 0..{F - 1}"
    base = {"value": port_v, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (restatement of pathtrace.comp, OpenMP, one thread team), {sample_txt}, {port_t:.1f} s"}
    parity = [parity_against(port_acc, "oracle (oracle/liborc.so)")] if img_first is not None else []
    del port_acc
    try:
        from tests import ref
        if os.path.exists(ref.LIB_PATH):
            rr = ref.Reference(wl.scene, wl.env, oracle=o)
            rr.set_camera(cam)
            rr.set_sunsky(hd.default_sun_and_sky())
            ref_v, ref_t, ref_acc = timed(lambda st_, f0, nf, acc, ids_: rr.render_frames(st_, f0, nf, acc, ids_, threads=cores))
            base = {"value": ref_v, "unit": "Msamples/s", "cores": cores, "kind": "reference",
                    "sample": f"oracle/_ref = the reference's shaders/pathtrace.comp compiled for the host (OpenMP, one invocation per pixel like vkCmdDispatch; ray queries and "
                              f"texture filtering bound to the oracle's trace contract), {sample_txt}, {ref_t:.1f} s",
                    "port_value": port_v, "port_sample": base["sample"]}
            if img_first is not None:
                parity.insert(0, parity_against(ref_acc, "oracle/_ref (the reference's pathtrace.comp compiled for the host)"))
            del ref_acc
    except Exception as e:  # the compiled reference is optional evidence; the port above stands
        base["reference_error"] = repr(e)
    if parity:
        out["parity"] = dict(parity[0], tolerance=1e-3, also=parity[1:],
                             note="GPU accumulation image after the first timed window (frames 0 .. warmup+steps-1) vs CPU renders of the same frames on the sampled pixels")
    out["cpu_baseline"] = base
    cr, sr = max(1, os_["closestRays"]), max(1, os_["shadowRays"])
    alg = {
        "nodes_per_closest_ray": (os_["nodesVisited"] - os_["nodesShadow"]) / cr,
        "tris_per_closest_ray": (os_["trisTested"] - os_["trisShadow"]) / cr,
        "nodes_per_shadow_ray": os_["nodesShadow"] / sr,
        "tris_per_shadow_ray": os_["trisShadow"] / sr,
        "tex_taps_per_hit": os_["texTaps"] / max(1, os_["shadedHits"]),
    }
    return out["cpu_baseline"], out.get("parity"), alg


# SURVEY.md 8(d) algorithmic bytes: reference-layout BVH2 visits x 32 B, triangle tests x 36 B, hit shading 348 B + 16 B per texture tap,
# any-hit evaluation 340 B, NEE lookup 80 B, miss 64 B, framebuffer 32 B per sample
def trace_bytes(alg, closest, shadow, alpha_c, alpha_s):
    return (closest * (alg["nodes_per_closest_ray"] * 32 + alg["tris_per_closest_ray"] * 36) + alpha_c * 340,
            shadow * (alg["nodes_per_shadow_ray"] * 32 + alg["tris_per_shadow_ray"] * 36) + alpha_s * 340)


def shade_bytes(alg, hits, misses, nee):
    return hits * (348 + 16 * alg["tex_taps_per_hit"]) + nee * 80 + misses * 64


def evidence_fields(out, serial, alg, samples, elapsed, world, workload):
    """Roofline / binding / HBM / VALU fields of the line from the serialised pass of this run and the committed PMC summaries (profiles/).  Pure
    post-processing: a CPU test feeds it a recorded run (tests/test_bench_cpu.py)."""
    # ---- roofline of the dominant stage: chosen by its standalone time, priced on algorithmic bytes ---------------------
    if alg is not None and serial is not None:
        sr, tl = serial["rays"], serial["rays_in_tail"]
        st_c, st_s = sr["closestRays"] - tl["closestRays"], sr["shadowRays"] - tl["shadowRays"]           # rays of the staged kernels
        st_a, tl_a = sr["alphaTests"] - tl["alphaTests"], tl["alphaTests"]
        fa = st_c / max(1, st_c + st_s)
        ft = tl["closestRays"] / max(1, tl["closestRays"] + tl["shadowRays"])
        bc, bs = trace_bytes(alg, st_c, st_s, st_a * fa, st_a * (1 - fa))
        tc, ts = trace_bytes(alg, tl["closestRays"], tl["shadowRays"], tl_a * ft, tl_a * (1 - ft))
        st_hits, st_miss = sr["shadedHits"] - tl["shadedHits"], sr["misses"] - tl["misses"]
        stage_bytes = {
            "closest": bc, "shadow": bs, "shade": shade_bytes(alg, st_hits, st_miss, st_hits),
            "tail": tc + ts + shade_bytes(alg, tl["shadedHits"], tl["misses"], tl["shadedHits"]),
            "generate": 0.0, "fused": 0.0,  # (k_trace_p runs in launch sequences of ONE frame only: the serialised batch has none; its rays are priced under closest / shadow)
            # per sample the radiance handed over (16 B); per launch the running mean of the local framebuffer read and written ONCE (32 B per pixel)
            "accumulate": serial["samples"] * 16.0 + serial["samples"] / max(1, serial["frames"]) * 32.0,
        }
        kernels = {"closest": "k_closest_k + k_closest_p (+ k_closest_x)", "shadow": "k_shadow_p (+ k_shadow_x)", "shade": "k_shade", "tail": "k_tail", "generate": "k_generate", "fused": "k_trace_p (+ k_trace_x)",
                   "accumulate": "k_accumulate"}
        launches = max(1, serial["launches_per_stage"])
        cache, cache_src = latest_profile("cache", workload)
        traffic_j, traffic_src = latest_profile("traffic", workload)
        table = {}
        for k, ms in serial["stage_ms"].items():
            n_l = launches if k in ("closest", "shade", "shadow") else max(1, serial["launches_tail"]) if k == "tail" else 1
            row = {"kernel": kernels[k], "ms": ms, "launches": n_l, "avg_launch_ms": ms / n_l, "alg_bytes": stage_bytes[k],
                   "alg_GBps": (stage_bytes[k] / (ms * 1e-3) / 1e9) if ms > 0 else None}
            if traffic_j and traffic_j["hbm_bytes_per_sample"].get(k) is not None and ms > 0:
                row["hbm_bytes"] = traffic_j["hbm_bytes_per_sample"][k] * serial["samples"]
                row["hbm_GBps"] = row["hbm_bytes"] / (ms * 1e-3) / 1e9
            if cache and cache.get("l2_requests_per_sample", {}).get(k) is not None and ms > 0:
                row["l2_bytes"] = cache["l2_requests_per_sample"][k] * L2_LINE * serial["samples"]
                row["l2_GBps"] = row["l2_bytes"] / (ms * 1e-3) / 1e9
                row["l2_hit_rate"] = cache.get("l2_hit_rate", {}).get(k)
            table[k] = row
        # The ceiling that actually binds each stage, as a measured fraction (SURVEY 8(d)'s HBM model above does not bind: its `frac` exceeds 1):
        #   shade            HBM-side bytes (PMC) / standalone time against the streaming-read ceiling CALIBRATED IN THIS RUN (pt_measure_peaks)
        #   closest, shadow  L1 (TCP) accesses (PMC, 64-byte accesses after the texture addresser's coalescing) / standalone time against one access per
        #                    clock and CU -- the per-lane request path of divergent rays -- next to the share of a wavefront's cycles spent waiting
        #                    (SQ_WAIT_ANY / SQ_WAVE_CYCLES of the stage's kernels): these stages are latency-chained, neither ceiling is reached
        valu_b, valu_b_src = latest_profile("valu", workload)
        cu, clk = out["calibration"]["compute_units"], out["calibration"]["clock_MHz"] * 1e6
        stage_kernels = {"closest": ("k_closest_k", "k_closest_p"), "shadow": ("k_shadow_p",), "shade": ("k_shade",), "tail": ("k_tail",)}
        for k, row in table.items():
            if not row.get("ms"):
                continue
            if k == "shade" and row.get("hbm_GBps"):
                row["binding"] = {"ceiling": "HBM streaming read measured in this run (pt_measure_peaks)", "achieved_GBps": row["hbm_GBps"], "peak_GBps": out["calibration"]["hbm_read_GBps"],
                                  "frac": row["hbm_GBps"] / out["calibration"]["hbm_read_GBps"], "frac_of_spec_peak": row["hbm_GBps"] / HBM_PEAK_GBS}
            elif k in ("closest", "shadow") and cache and cache.get("l1_accesses_per_sample", {}).get(k) is not None:
                acc = cache["l1_accesses_per_sample"][k] * serial["samples"] / (row["ms"] * 1e-3)
                b = {"ceiling": "L1 (TCP) request rate: one 64-byte access per clock and CU", "achieved_G_per_s": acc / 1e9, "peak_G_per_s": cu * clk / 1e9, "frac": acc / (cu * clk),
                     "l1_accesses_per_sample": cache["l1_accesses_per_sample"][k], "source": cache_src}
                try:
                    cyc = valu_b["sq_cycles_per_sample"]
                    w = sum(cyc[kn]["SQ_WAIT_ANY"] for kn in stage_kernels[k] if kn in cyc)
                    t = sum(cyc[kn]["SQ_WAVE_CYCLES"] for kn in stage_kernels[k] if kn in cyc)
                    va = sum(cyc[kn]["SQ_ACTIVE_INST_VALU"] for kn in stage_kernels[k] if kn in cyc)
                    b.update({"wave_cycles_waiting_frac": w / t, "wave_cycles_valu_active_frac": va / t, "sq_source": valu_b_src})
                except Exception:
                    pass
                row["binding"] = b
        out["stages_serialised"] = table
        dom = max(("closest", "shade", "shadow", "tail"), key=lambda k: serial["stage_ms"][k])
        d = table[dom]
        traffic = d.get("hbm_bytes") / d["launches"] if d.get("hbm_bytes") else None
        if traffic_j:
            tot = traffic_j["hbm_bytes_per_sample"]["total"]
            out["hbm_measured"] = {"bytes_per_sample": tot, "GBps": tot * samples / max(1, world) / elapsed / 1e9, "frac": tot * samples / max(1, world) / elapsed / 1e9 / HBM_PEAK_GBS,
                                   "source": f"{traffic_src} (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh on the timed pipeline incl. k_tail, gfx950 corrections) x this run's rate",
                                   "note": "what actually crosses the HBM interface per sample, against the 8 TB/s peak: the scene's working set lives in L2 / Infinity Cache"}
        # ---- what binds, measured (round 6).  Per stage: VALU wave-instructions per sample (rocprofv3 SQ_INSTS_VALU, profiles/rNN_valu.json) x the samples of
        # the serialised pass / the stage's standalone time = achieved issue rate, against the ceiling tools/valu_mix.hip reaches for THAT kernel's instruction mix
        # at that kernel's occupancy (no memory instruction at all); next to it the hardware lane occupancy of those instructions
        # (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, profiles/rNN_binders.json) and the HBM and L1 fractions, which do NOT bind.
        mix, mix_src = mix_ceilings()
        binders, binders_src = latest_profile("binders", workload)
        valu_k = (valu_b or {}).get("kernels", {})
        mix_of = {"k_closest_k": "packet", "k_closest_p": "trace", "k_shadow_p": "trace", "k_shade": "shade"}
        for k, row in table.items():
            if not row.get("ms") or k not in stage_kernels or not mix:
                continue
            ks = [kn for kn in stage_kernels[k] if kn in valu_k and "SQ_INSTS_VALU" in valu_k[kn]]
            if not ks:
                continue
            instr = sum(valu_k[kn]["SQ_INSTS_VALU"] for kn in ks)                       # per sample
            ceil = sum(valu_k[kn]["SQ_INSTS_VALU"] * mix.get(mix_of.get(kn, "trace"), 0.0) for kn in ks) / instr  # instruction-weighted when a stage has two kernels
            ach = instr * serial["samples"] / (row["ms"] * 1e-3) / 1e9
            iss = {"valu_wave_instr_per_sample": instr, "achieved_G_per_s": ach, "mix_ceiling_G_per_s": ceil, "frac": ach / ceil if ceil else None, "ceiling_source": mix_src, "instr_source": valu_b_src}
            if binders:
                bk = binders.get("kernels", {})
                iss["lanes_per_valu_instr"] = {kn: bk[kn].get("lanes_per_valu_instr") for kn in stage_kernels[k] if kn in bk}
                iss["lanes_source"] = binders_src
            row["issue"] = iss
        # `roofline`: the dominant stage against the ceiling that binds it.  SURVEY.md 8(d)'s HBM accounting moves to `alg_*` (it exceeds 1: the numerator prices
        # BVH2-equivalent visits whose bytes are cache-served); the HBM interface itself is `traffic` / `hbm`.
        iss = d.get("issue")
        out["roofline"] = {"bound": "valu_issue" if iss else "hbm", "kernel": d["kernel"], "stage": dom,
                           "achieved": iss["achieved_G_per_s"] if iss else d.get("hbm_GBps"), "peak": iss["mix_ceiling_G_per_s"] if iss else HBM_PEAK_GBS,
                           "unit": "G wave-instructions/s" if iss else "GB/s", "frac": iss["frac"] if iss else (d["hbm_GBps"] / HBM_PEAK_GBS if d.get("hbm_GBps") else None),
                           "basis": ("VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU) / average standalone launch duration (HIP events on the launching stream, serialised pass of this run) "
                                     "against the issue rate tools/valu_mix.hip reaches for this kernel's instruction mix at its occupancy") if iss else "HBM-side bytes (PMC) / standalone duration",
                           "lanes_per_valu_instr": iss.get("lanes_per_valu_instr") if iss else None,
                           "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
                           "traffic": traffic, "traffic_source": traffic_src,
                           "hbm": {"achieved": d.get("hbm_GBps"), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["hbm_GBps"] / HBM_PEAK_GBS if d.get("hbm_GBps") else None,
                                   "note": "HBM-side bytes of this stage (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 corrections) / its standalone time: the HBM interface does not bind"},
                           "alg_bytes_per_launch": d["alg_bytes"] / d["launches"], "alg_achieved_GBps": d["alg_GBps"], "alg_frac": d["alg_GBps"] / HBM_PEAK_GBS if d["alg_GBps"] else None,
                           "alg_note": "SURVEY 8(d) accounting: algorithmic bytes per launch (BVH2-EQUIVALENT node visits of the oracle's binary tree, ~110 per ray x 32 B, + triangle, shading, "
                                       "any-hit, NEE records) / the launch duration vs 8 TB/s.  It exceeds 1 because those bytes are served by L2 / Infinity Cache: an accounting figure, not a speed",
                           "l2_GBps": d.get("l2_GBps"), "l2_frac": d["l2_GBps"] / L2_PEAK_GBS if d.get("l2_GBps") else None, "l2_hit_rate": d.get("l2_hit_rate"), "l2_source": cache_src,
                           "l1_request_path": d.get("binding") if dom in ("closest", "shadow") else None,
                           "measured_hbm_copy_GBps": out["calibration"]["hbm_copy_GBps"],
                           "per_stage": {k: {"ms": row["ms"], "valu_issue_frac": (row.get("issue") or {}).get("frac"), "lanes_per_valu_instr": (row.get("issue") or {}).get("lanes_per_valu_instr"),
                                             "hbm_frac": row["hbm_GBps"] / HBM_PEAK_GBS if row.get("hbm_GBps") else None} for k, row in table.items() if row.get("ms")},
                           "note": "dominant stage = largest STANDALONE time.  Every kernel of the pipeline sits at 0.6-0.8 of the VALU issue ceiling of its own mix with a third (trace machine) to "
                                   "three quarters (k_shade) of the lanes active per instruction; HBM 0.17-0.5, L1 request path 0.3-0.44: issue binds, lane occupancy is what it is spent on"}
    # ---- VALU issue: instruction counts per sample are a property of the code and the workload (rocprofv3 PMC pass of this round); the rate is
    # this run's; the ceiling is the one measured above on this box.
    valu_j, valu_src = latest_profile("valu", workload)
    if valu_j:
        try:
            per_sample = valu_j["valu_wave_instr_per_sample"]
            peak = out["calibration"]["valu_G_wave_instr_per_s"] * 1e9
            ach = per_sample * samples / elapsed / max(1, world)
            out["issue_roofline"] = {"bound": "valu", "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s per GPU", "frac": ach / peak,
                                     "valu_wave_instr_per_sample": per_sample, "source": f"{valu_src} x this run's rate / this run's calibration"}
        except Exception:
            pass


def build_workload(args):
    from vk_raytrace_amd import capi, workloads
    if args.from_gltf:
        # the FILE route as the timed route (reference: Scene::load, src/scene.cpp:57-118): the .glb / .gltf goes through libptmi's own importer
        from vk_raytrace_amd.scene import GltfFileScene
        from vk_raytrace_amd import synth
        sc = GltfFileScene(args.from_gltf)
        wl = workloads.Workload(f"C3 configuration on the glTF file {os.path.basename(args.from_gltf)} ({sc.num_triangles} tris) {args.width}x{args.height} {args.steps}spp depth8 Disney + HDR env",
                                sc, synth.procedural_sky(2048, 1024), args.width, args.height, args.steps, 8, 0, note="gltf")
    elif args.workload == "c3":
        wl = workloads.c3_sponza(args.width, args.height, args.steps, tex_size=args.tex_size, target_tris=args.tris)
    else:
        wl = {"c2": workloads.c2_helmet, "c4": workloads.c4_sponza_4k, "c5": workloads.c5_bistro}[args.workload]()
        wl.name = wl.name.replace(f"{wl.spp}spp", f"{args.steps}spp")
    wl.scene.finalize(capi.pack_vertices)
    return wl


def setup_renderer(args, wl, device, shard_rank, shard_n):
    """One context on `device` that renders rank `shard_rank`'s tiles of `shard_n`: scene, environment, camera, acceleration structure, frame slots."""
    from vk_raytrace_amd import capi
    from vk_raytrace_amd.renderer import HipRenderer
    from vk_raytrace_amd import host_device as hd
    r = HipRenderer()
    r.setup(device)
    if args.accel == "two":
        r.set_accel_mode(capi.PT_ACCEL_TWO_LEVEL)
    r.set_shard(shard_rank, shard_n)
    r.set_scene(wl.scene)
    integral, _ = r.set_env(wl.env)
    cam = capi.camera_lookat(wl.scene.camera, wl.width / wl.height, nb_lights=len(wl.scene.lights))
    r.set_camera(cam)
    r.set_sunsky(hd.default_sun_and_sky())
    r.create((wl.width, wl.height))
    return r, integral, cam


def rccl_version():
    import ctypes as C
    from vk_raytrace_amd import capi
    v = C.c_int(0)
    rc = capi.lib().pt_comm_version(C.byref(v))
    return v.value if rc == capi.PT_OK else None


def with_timeout(fn, seconds, what):
    """fn() on a helper thread; a collective that never returns (a wedged RCCL) must not take the line with it.  Returns (ok, result | error text)."""
    import threading
    box = {}

    def run():
        try:
            box["v"] = fn()
        except Exception as e:  # noqa: BLE001
            box["e"] = repr(e)
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        # the helper thread is still INSIDE the call (and inside the pt_context it was given): that context must not be touched again
        return False, TimedOut(f"{what}: no answer after {seconds:.0f} s")
    return ("e" not in box), box.get("e", box.get("v"))


class TimedOut(str):
    """error text of a with_timeout() call whose helper thread never came back: the objects it was using are poisoned"""


def single_process(args):
    """`--single-process`: ONE process drives all N GPUs -- N contexts, pt_comm_init_all, the shards gathered inside one pt_comm_group_begin / end
    (SURVEY.md 8(e) "single process ... one stream per device"; include/pt_api.h).  Also what `--gpus N` falls back to when its self-launched ranks fail
    to come up (rendezvous, RCCL bootstrap between processes)."""
    import ctypes as C
    from vk_raytrace_amd import capi
    from vk_raytrace_amd import host_device as hd
    n = args.gpus
    t_setup = time.time()
    wl = build_workload(args)
    W, H = wl.width, wl.height
    rs, integral, cam = [], None, None
    for i in range(n):
        r, integral, cam = setup_renderer(args, wl, i, i, n)
        rs.append(r)
    L = capi.lib()
    comms = (C.c_void_p * n)()
    dev = (C.c_int * n)(*range(n))
    rc = L.pt_comm_init_all(n, dev, comms)
    if rc != capi.PT_OK:
        raise capi.PtError(rc, "pt_comm_init_all: " + L.pt_comm_last_error().decode())
    cnt = C.c_int(0)
    L.pt_comm_count(comms[0], C.byref(cnt))
    print(f"bench.py preflight: single process, {n} context(s), RCCL {rccl_version()}, communicator spans {cnt.value} rank(s)", file=sys.stderr)
    st = hd.default_rtx_state()
    st.size[0], st.size[1] = W, H
    st.maxDepth, st.pbrMode, st.maxSamples = wl.depth, wl.pbr_mode, 1
    st.fireflyClampThreshold = 4.0 * integral
    t_setup = time.time() - t_setup

    def gather():
        t0 = time.perf_counter()
        for r in rs:
            r.synchronize()
        rs[0]._check(L.pt_comm_group_begin())
        for i, r in enumerate(rs):
            r._check(L.pt_gather_shards(r._ctx, comms[i], 0))
        rs[0]._check(L.pt_comm_group_end())
        for r in rs[1:]:
            r.synchronize()
        rs[0]._check(L.pt_gather_finish(rs[0]._ctx))
        img = rs[0].read_accum()
        return img, (time.perf_counter() - t0) * 1e3

    frame = 0

    def frames(k):
        nonlocal frame
        for _ in range(k):
            st.frame = frame
            for r in rs:
                r.setPushContants(st)
                r.run()
            frame += 1
    frames(args.warmup)
    for r in rs:
        r.synchronize()
        r.reset_stats()
    windows, per_rank_ms, img_first, stats = [], None, None, None
    for rep in range(max(1, args.repeats)):
        t0 = time.perf_counter()
        frames(args.steps)
        done = []
        for r in rs:
            r.synchronize()
            done.append((time.perf_counter() - t0) * 1e3 / args.steps)
        windows.append(time.perf_counter() - t0)
        if rep == 0:
            per_rank_ms = done
            stats = [r.stats() for r in rs]
            img_first, _ = gather()
    img, gather_ms = gather()
    RAY_KEYS = ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups")
    total = dict(stats[0])
    for k in RAY_KEYS:
        total[k] = int(sum(s_[k] for s_ in stats))
    return {"wl": wl, "W": W, "H": H, "windows": windows, "img_first": img_first, "img": img, "gather_ms": gather_ms, "stats": total, "ranks_seen": cnt.value, "t_setup": t_setup,
            "integral": integral, "cam": cam, "renderer": rs[0], "st": st, "frame": frame, "per_rank_ms": per_rank_ms, "gather": "rccl (pt_comm_init_all, one process)",
            "cleanup": lambda: [L.pt_comm_destroy(c) for c in comms]}


def multi_process(args, rank, local_rank, world):
    """One process per GPU (the driver's launch, or bench.py's self-launch): this rank's part of the job.  Returns the same record as single_process on
    rank 0, None on the other ranks."""
    from vk_raytrace_amd import capi
    from vk_raytrace_amd import shard
    from vk_raytrace_amd import host_device as hd

    # bind the GPU first: "device ordinal R out of range: K HIP device(s) visible" comes from pt_create, before any rendezvous can hang
    same_device = os.environ.get("PT_BENCH_SAME_DEVICE") == "1"  # TEST HOOK (tests/test_comm.py): every rank on device 0, so that the multi-process flow --
    if same_device:                                              # rendezvous, barriers around the windows, MAX of the ranks' times, counters summed, the gathered
        local_rank = 0                                           # image's parity -- runs on a one-GPU box; RCCL refuses two ranks on one device: host gather
    t_setup = time.time()
    wl = build_workload(args)
    W, H = wl.width, wl.height
    shard_rank, shard_n = rank, world
    if args.emulate_shard:
        shard_rank, shard_n = (int(x) for x in args.emulate_shard.split("/"))
        args.no_cpu_baseline = True
    r, integral, cam = setup_renderer(args, wl, local_rank, shard_rank, shard_n)

    force_dist = os.environ.get("PT_BENCH_FORCE_DIST") == "1"  # exercise the group / RCCL plumbing with one rank
    dist = None
    if world > 1 or force_dist:
        from vk_raytrace_amd.rendezvous import LocalGroup
        dist = LocalGroup(rank, world)  # control plane only (see the module docstring)
    st = hd.default_rtx_state()
    st.size[0], st.size[1] = W, H
    st.maxDepth, st.pbrMode, st.maxSamples = wl.depth, wl.pbr_mode, 1
    st.fireflyClampThreshold = 4.0 * integral
    t_setup = time.time() - t_setup

    def sync():
        r.synchronize()  # pt_synchronize: every stream of the context has drained (the device-side bracket of the timed region)
        if dist is not None:
            dist.barrier()

    # ---- the one collective of the path, set up and tried BEFORE anything is timed (preflight): libptmi's own RCCL gather behind the C ABI
    # (pt_gather_shards / pt_gather_finish).  If the communicator cannot be built, or the first gather does not come back, every rank agrees on the
    # host route instead (the shards travel through the control-plane socket): slower, untimed either way, and the line says which one it was.
    gatherer, how, poisoned = None, "none (one rank)", False
    if world > 1 or force_dist:
        ok, res = (False, "PT_BENCH_SAME_DEVICE: two ranks on one device") if (same_device and world > 1) else with_timeout(lambda: shard.NativeGather(rank, world, local_rank, dist), 120, "pt_comm_init_rank")
        all_ok = dist.all_reduce([1.0 if ok else 0.0], "sum")[0] == world
        if all_ok:
            gatherer, how = res, "rccl (pt_comm_init_rank, one process per GPU)"
        else:
            how = f"host (control-plane socket); RCCL route unavailable: {res if not ok else 'another rank failed'}"
        if rank == 0:
            print(f"bench.py preflight: {world} process(es), RCCL {rccl_version()}, gather route: {how}", file=sys.stderr)
    ranks_seen = gatherer.ranks_seen() if gatherer is not None else (world if (world > 1 and dist is not None) else 1)

    def gather():
        """the full image on rank 0 (None elsewhere) and the milliseconds it took"""
        nonlocal gatherer, how, poisoned
        if poisoned:
            return None, 0.0
        r.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        if world == 1 and not force_dist:
            return r.read_accum(), (time.perf_counter() - t0) * 1e3
        if gatherer is not None:
            ok, res = with_timeout(lambda: gatherer.gather(r), 120, "pt_gather_shards")
            votes = dist.all_reduce([1.0 if ok else 0.0, 1.0 if isinstance(res, TimedOut) else 0.0], "sum")
            if votes[0] == world:
                return res, (time.perf_counter() - t0) * 1e3
            if votes[1] > 0:
                # a rank's helper thread is still inside pt_gather_shards / RCCL on its context: pt_context is not thread-safe, so no rank reads an image
                # (the collective may still be writing it) -- the line is emitted with the gather error and without the parity leg
                poisoned = True
                gatherer, how = None, f"FAILED: pt_gather_shards did not return on {int(votes[1])} rank(s) (contexts poisoned; no image, no parity): {res if not ok else 'on another rank'}"
                return None, (time.perf_counter() - t0) * 1e3
            gatherer, how = None, f"host (control-plane socket); the RCCL gather failed: {res if not ok else 'on another rank'}"
        mine = r.read_accum()  # this rank's pixels are valid in it
        ids = shard.local_pixel_ids(W, H, shard_rank, shard_n)
        parts = dist.gather_bytes(np.ascontiguousarray(mine.reshape(-1, 4)[ids]).tobytes())
        if rank != 0:
            return None, (time.perf_counter() - t0) * 1e3
        out = np.zeros((H * W, 4), np.float32)
        for q, blob in enumerate(parts):
            out[shard.local_pixel_ids(W, H, q, world)] = np.frombuffer(blob, np.float32).reshape(-1, 4)
        return out.reshape(H, W, 4), (time.perf_counter() - t0) * 1e3

    frame = 0
    for _ in range(args.warmup):
        st.frame = frame
        r.setPushContants(st)
        r.run()
        frame += 1
    sync()
    r.reset_stats()

    # The timed window: exactly --steps frames between two (device synchronisation + barrier) brackets, MAX over the ranks.  It is run --repeats
    # times back to back on the same accumulation image (frames keep counting up); `value` is the MEDIAN window and every window is reported, so
    # that one 30 ms window on a box that has just come up (clocks, first use of a frame slot) cannot decide the line on its own.
    windows, img_first, stats, per_rank_ms = [], None, None, None
    for rep in range(max(1, args.repeats)):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.frame = frame
            r.setPushContants(st)
            r.run()
            frame += 1
        r.synchronize()
        mine = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            dt = dist.all_reduce([dt], "max")[0]
        windows.append(dt)
        if rep == 0:
            stats = r.stats()  # ray counters of ONE window (the per-sample figures below are per window)
            if dist is not None:
                slot = [0.0] * world
                slot[rank] = mine * 1e3 / args.steps
                per_rank_ms = dist.all_reduce(slot, "sum")   # every rank's own time for the window (before the barrier): who is the slowest
            # frames 0 .. warmup+steps-1 of the WHOLE image: what the parity leg re-renders on the CPU (untimed: between two windows)
            img_first, _ = gather()
    img, gather_ms = gather()

    RAY_KEYS = ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups")
    if dist is not None:
        v = dist.all_reduce([float(stats[k]) for k in RAY_KEYS], "sum")  # whole-job counters
        for i, k in enumerate(RAY_KEYS):
            stats[k] = int(v[i])
    if gatherer is not None:
        gatherer.close()
    if rank != 0:
        if dist is not None:
            dist.barrier()  # rank 0 finishes its evidence legs, then everybody leaves
            dist.close()
        return None
    return {"wl": wl, "W": W, "H": H, "windows": windows, "img_first": img_first, "img": img, "gather_ms": gather_ms, "stats": stats, "ranks_seen": ranks_seen, "t_setup": t_setup,
            "integral": integral, "cam": cam, "renderer": r, "st": st, "frame": frame, "per_rank_ms": per_rank_ms, "gather": how, "poisoned": poisoned, "shard": (shard_rank, shard_n),
            "cleanup": (lambda: (dist.barrier(), dist.close())) if dist is not None else (lambda: None)}


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # multi-process GPU work on this driver: dmabuf IPC only (RCCL reads it at init)
    launched = "WORLD_SIZE" in os.environ
    if args.single_process and not launched:
        job = single_process(args)
        world, local_rank, mode = args.gpus, 0, "single-process"
    else:
        if not launched and args.gpus > 1:
            rc = self_launch(args.gpus)
            if rc == 0 or args.no_fallback:
                raise SystemExit(rc)
            # the ranks did not come up (rendezvous, RCCL bootstrap between processes, ...): the same job from ONE process instead of no line at all
            print(f"bench.py: the self-launched ranks failed (exit code {rc}); falling back to --single-process", file=sys.stderr)
            job = single_process(args)
            world, local_rank, mode = args.gpus, 0, "single-process (fallback: the self-launched ranks failed)"
        else:
            rank = int(os.environ.get("RANK", "0"))
            local_rank = int(os.environ.get("LOCAL_RANK", "0"))
            world = int(os.environ.get("WORLD_SIZE", "1"))
            args.gpus = world  # under a launcher the launcher's world size is the truth
            job = multi_process(args, rank, local_rank, world)
            mode = "one process per GPU"
            if job is None:
                return
    from vk_raytrace_amd import capi
    from vk_raytrace_amd.renderer import HipRenderer
    from vk_raytrace_amd import host_device as hd
    wl, W, H, windows, img_first, img, gather_ms, stats = (job[k] for k in ("wl", "W", "H", "windows", "img_first", "img", "gather_ms", "stats"))
    r, st, frame, integral, cam, ranks_seen, t_setup = (job[k] for k in ("renderer", "st", "frame", "integral", "cam", "ranks_seen", "t_setup"))
    shard_rank, shard_n = job.get("shard", (0, world))
    elapsed = float(np.median(windows))
    frames_first = args.warmup + args.steps
    RAY_KEYS = ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups")
    if job.get("per_rank_ms") and world > 1:
        pr = job["per_rank_ms"]
        print(f"bench.py preflight: ms/frame per rank in the first window {[round(x, 4) for x in pr]}; slowest rank {int(np.argmax(pr))} at {max(pr) / (sum(pr) / len(pr)):.3f} x the mean", file=sys.stderr)

    # pixels actually rendered by this job (an emulated shard renders one rank's tiles only)
    job_pixels = W * H if not args.emulate_shard else int(stats["samples"] // max(1, args.steps))
    samples = job_pixels * args.steps
    value = samples / elapsed / 1e6
    is_file = getattr(wl, "note", "") == "gltf"
    out = {
        "metric": "Msamples/s on Sponza 1920x1080; per-pixel L2 vs ref at equal spp",
        "value": value,
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "repeats": [samples / w / 1e6 for w in windows],
        "repeat_policy": f"{len(windows)} windows of {args.steps} steps each, back to back, each bracketed by device synchronisation + barrier; value = median window",
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "gltf" if is_file else "synthetic",
        "config": {"workload": wl.name, "triangles": wl.scene.num_triangles, "materials": len(wl.scene.materials), "textures": len(wl.scene.textures),
                   "width": W, "height": H, "spp": args.steps, "max_depth": wl.depth, "bsdf": "disney" if wl.pbr_mode == 0 else "gltf", "env": f"{wl.env.shape[1]}x{wl.env.shape[0]} procedural HDR",
                   "parallelism": f"image tiles {hd.TILE}x{hd.TILE} over {world} GPU(s), scene replicated" + (f"; this run = shard {args.emulate_shard} on one GPU" if args.emulate_shard else "")},
        "ranks_seen": ranks_seen,
        "launch": mode,
        "gather": job.get("gather"),
        "ms_per_step_per_rank": job.get("per_rank_ms"),
        "setup_s": t_setup,
        "gather_ms": gather_ms,
        "bvh_build_ms": stats["msBuildAccel"],
        "accel": {"mode": args.accel, "bytes": stats["bytesAccel"], "blas": stats["numBlas"], "tlas_nodes": stats["numTlasNodes"], "nodes": stats["numBvhNodes"]},
        "batch": {"frames": stats["batchFrames"], "in_flight": stats["framesInFlight"]},
        "rays": {k: stats[k] for k in RAY_KEYS},
        "image_mean": float(np.mean(img[..., :3])) if img is not None else None,
    }

    if world > 1 and not args.emulate_shard:
        try:
            out["scaling_prediction"] = scaling_prediction(args.workload, args.steps, world, out["ms_per_step"])
        except Exception as e:  # noqa: BLE001
            out["scaling_prediction"] = {"error": repr(e)}

    if args.refit > 0:
        # instance update (pt_update_instances): every node's world matrix is re-sent; two-level mode redoes only the instance boxes + TLAS
        nodes = wl.scene.node_array()
        r.update_instances(nodes)
        t0 = time.perf_counter()
        for _ in range(args.refit):
            r.update_instances(nodes)
        out["accel"]["update_instances_ms"] = (time.perf_counter() - t0) * 1e3 / args.refit

    # ---- CPU baseline + algorithmic bytes ---------------------------------------------------------------------------------------------
    # kind "reference": oracle/_ref/libref.so = shaders/pathtrace.comp:87-134 with everything it includes, compiled for the host by the committed
    # recipe (oracle/ref_glue), dispatched over a bounded pixel sample on the host cores; what the Vulkan driver would supply (ray queries,
    # texture filtering) is bound to the oracle's trace contract.  `port_value`: the oracle's own restatement on the same sample.
    # N > 1: the same leg on rank 0, against the GATHERED image of the first window -- the line of a multi-GPU run carries both halves of the metric too
    alg = None
    if not args.no_cpu_baseline:
        out["cpu_baseline"], par, alg = cpu_leg(wl, cam, integral, W, H, frames_first, img_first, args.cpu_seconds)
        if par is not None:
            out["parity"] = par
            if world > 1:
                par["note"] = f"image GATHERED from {world} ranks ({job.get('gather')}) after the first timed window vs CPU renders of the same frames on the sampled pixels"
    elif os.path.exists(os.path.join(ROOT, "profiles", "alg_bytes_c3.json")):
        alg = json.load(open(os.path.join(ROOT, "profiles", "alg_bytes_c3.json")))

    if alg is not None:
        rays = out["rays"]
        fc = rays["closestRays"] / max(1, rays["closestRays"] + rays["shadowRays"])
        bc, bs = trace_bytes(alg, rays["closestRays"], rays["shadowRays"], rays["alphaTests"] * fc, rays["alphaTests"] * (1 - fc))
        b_total = bc + bs + shade_bytes(alg, rays["shadedHits"], rays["misses"], rays["neeLookups"]) + samples * 32
        out["alg_bytes_per_sample"] = b_total / samples
        out["alg_model"] = alg

    # ---- measurement passes after the timed region (never part of `value`) ---------------------------------------------------------
    # (1) ceilings measured on this box: VALU issue (independent wave64 v_fmac_f32, 8 waves/SIMD on every CU) and HBM streaming
    if job.get("poisoned"):
        # a collective never returned on this context (helper thread still inside it): nothing below may touch the renderer again -- the line goes out as it is
        out["gather_error"] = job.get("gather")
        print(json.dumps(out))
        sys.stdout.flush()
        with_timeout(job["cleanup"], 10, "control-plane barrier")  # let the other ranks leave (the control plane is independent of the wedged collective)
        os._exit(4)  # (not sys.exit: the helper thread inside the collective would keep the interpreter alive)
    peaks = r.measure_peaks()
    out["calibration"] = {"valu_G_wave_instr_per_s": peaks["valuWaveInstrPerSec"] / 1e9, "hbm_copy_GBps": peaks["hbmCopyBytesPerSec"] / 1e9,
                          "hbm_read_GBps": peaks["hbmReadBytesPerSec"] / 1e9, "compute_units": peaks["computeUnits"], "clock_MHz": peaks["clockMHz"],
                          "note": "pt_measure_peaks on this device; theoretical VALU issue = CUs x 4 SIMDs x clock / 2 cycles (MI355X_MICROARCH.md), tools/valu_peak.hip shows 0.45-0.8 of it depending on the instruction form"}
    # (2) the interactive path: SampleExample's loop tonemaps after every frame, so every frame is its own batch of one
    if not args.no_interactive and world == 1 and not args.emulate_shard:
        tm = hd.default_tonemapper()
        n_i = 24
        for _ in range(4):
            st.frame = frame; r.setPushContants(st); r.run(); r.tonemap(tm); frame += 1
        t0 = time.perf_counter()
        for _ in range(n_i):
            st.frame = frame; r.setPushContants(st); r.run(); r.tonemap(tm); frame += 1
        ti = time.perf_counter() - t0
        out["interactive"] = {"value": job_pixels * n_i / ti / 1e6, "unit": "Msamples/s", "ms_per_frame": ti / n_i * 1e3, "frames": n_i,
                              "note": "render + pt_tonemap (RGBA8 read back to the host) per frame: batch = 1, the host waits for every image"}
        # the same loop with frames in flight (pt_tonemap_begin / pt_tonemap_end, the reference's prepareFrame / submitFrame loop): the host
        # collects the image of the frame issued five calls earlier -- six launch sequences of one frame each in flight, the four batch slots and
        # the two one-frame display slots (profiles/r04y_display_slots.txt)
        n_p = 96
        behind = 5
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_p):
            st.frame = frame; r.setPushContants(st); r.run(); r.tonemap_begin(tm); frame += 1
            if r.tonemap_pending() > behind:
                r.tonemap_end()
        while r.tonemap_pending():
            r.tonemap_end()
        tp = time.perf_counter() - t0
        out["interactive"].update({"pipelined_value": job_pixels * n_p / tp / 1e6, "pipelined_ms_per_frame": tp / n_p * 1e3, "pipelined_frames": n_p,
                                   "pipelined_frames_behind": behind,
                                   "pipelined_note": "render + pt_tonemap_begin per frame, pt_tonemap_end for the frame issued five calls earlier (every image still reaches the host)"})
    # (3) standalone kernel durations: the timed loop overlaps four launch sequences on separate streams, so a HIP-event bracket there is not a
    # kernel's own duration.  One batch is rendered again on a second context with ONE frame slot (PT_TUNE inflight=1) and otherwise the launch
    # policy of the timed run (k_tail takes the late bounces): nothing overlaps, HIP events on the launching stream bracket each stage.
    serial = None
    if not args.no_profile:
        os.environ["PT_TUNE"] = (os.environ.get("PT_TUNE", "") + ",inflight=1").lstrip(",")
        r2 = HipRenderer()
        r2.setup(local_rank if mode == "one process per GPU" and os.environ.get("PT_BENCH_SAME_DEVICE") != "1" else 0)
        if args.accel == "two":
            r2.set_accel_mode(capi.PT_ACCEL_TWO_LEVEL)
        r2.set_shard(shard_rank, shard_n)
        r2.set_scene(wl.scene)
        r2.set_env(wl.env)
        r2.set_camera(cam)
        r2.set_sunsky(hd.default_sun_and_sky())
        r2.create((W, H))
        nser = min(args.steps, 32)
        # three batches: the first two give the queue-size feedback its observation (it decides where k_tail takes over), the third is measured
        for phase in range(3):
            r2.reset_stats()
            r2.set_profiling(phase == 2)
            t0 = time.perf_counter()
            for f in range(nser):
                st.frame = f; r2.setPushContants(st); r2.run()
            r2.synchronize()
            tser = time.perf_counter() - t0
        s2 = r2.stats()
        r2.set_profiling(False)
        r2.destroy()
        nsamp = float(s2["samples"])
        serial = {"frames": nser, "wall_ms": tser * 1e3, "launches_per_stage": int(s2["launchesTraceClosest"]), "launches_tail": int(s2["launchesTail"]),
                  "stage_ms": {"generate": s2["msGenerate"], "closest": s2["msTraceClosest"], "shade": s2["msShade"], "shadow": s2["msTraceShadow"], "tail": s2["msTail"],
                               "accumulate": s2["msAccumulate"], "fused": s2.get("msTraceFused", 0.0)},
                  "launches_fused": int(s2.get("launchesTraceFused", 0)),
                  "rays": {k: s2[k] for k in RAY_KEYS}, "rays_in_tail": {k: s2["tail" + k[0].upper() + k[1:]] for k in ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests")},
                  "samples": nsamp}
        out["serialised"] = serial

    try:
        evidence_fields(out, serial, alg, samples, elapsed, world, args.workload + ("two" if args.accel == "two" else ""))  # (the PMC summaries are per workload AND structure)
    except Exception as e:  # the evidence fields are additions: a failure in their post-processing must never cost the line itself
        out["evidence_error"] = repr(e)
    try:  # RCCL writes a version banner through C stdio, which a pipe holds back until exit -- AFTER the line.  Out with it now: the line is the last thing on stdout.
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out))
    sys.stdout.flush()
    job["cleanup"]()
    try:
        sys.stdout.flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)  # nothing a library prints at exit may follow the line
    except Exception:
        pass
    if out.get("parity") and not (out["parity"]["l2"] <= out["parity"]["tolerance"]):
        print(f"bench.py: parity FAILED: per-pixel L2 {out['parity']['l2']} against {out['parity']['against']} exceeds {out['parity']['tolerance']}", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
