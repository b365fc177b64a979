#!/usr/bin/env python
"""Headline benchmark: Msamples/s on the Sponza-class configuration of BASELINE.json (C3: 1920x1080,
depth 8, Disney BSDF, HDR environment; 256 spp == 256 steps).

A "step" is one frame: one sample per pixel over the whole image (maxSamples = 1, the reference default),
accumulated into the RGBA32F framebuffer exactly like shaders/pathtrace.comp:122-133.  Scene, BVH, textures
and environment are resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps 256 --warmup 8
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

With N > 1 the image tiles are sharded over the ranks (vk_raytrace_amd/shard.py); the ranks do not communicate while rendering and
the single framebuffer gather -- libptmi's own RCCL path, pt_gather_shards / pt_gather_finish -- happens after the timed loop (its time
is reported as gather_ms, like the reference metric which times the frame loop only - BASELINE.md section 2).

After the timed region (never part of `value`), rank 0 measures what the JSON line's evidence fields need, all in this run:
  calibration        pt_measure_peaks: the VALU-issue and HBM-streaming ceilings of this box
  interactive        render + tonemap per frame (batch = 1, SampleExample's display loop)
  serialised         one batch on a second context with one frame slot: standalone stage durations (HIP events, nothing overlapped)
  roofline           the stage with the largest standalone time, algorithmic bytes per launch / its average launch duration vs 8 TB/s;
                     `traffic` = measured HBM bytes per launch from profiles/r02_traffic.json (this round's PMC passes, tools/pmc_r02.sh)
  hbm_measured       measured HBM bytes per sample x this run's rate
  issue_roofline     VALU wave-instructions per sample (profiles/r02_valu.json, same PMC run) x this run's rate / the calibrated ceiling
  cpu_baseline       the CPU oracle (kind "port") on a bounded sample of the same workload, N = 1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before torch / HIP initialise (see vk_raytrace_amd/capi.py)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    # development knobs (the defaults are the BASELINE configuration)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--tris", type=int, default=262_267)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"], help="BASELINE.json configuration (stand-in scene); c3 is the bench line, the others are for the results table")
    ap.add_argument("--emulate-shard", default="", help="R/N: render only rank R's tiles of an N-GPU run on this one GPU (scaling estimate; value = this shard's rate)")
    ap.add_argument("--accel", default="flat", choices=["flat", "two"], help="acceleration structure: flat world-space hierarchy (default) or the reference's BLAS per prim-mesh + TLAS (pt_set_accel_mode)")
    ap.add_argument("--refit", type=int, default=0, help="with --accel two: time this many pt_update_instances calls (TLAS refit) after the run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the serialised profiling pass (roofline fields become null)")
    ap.add_argument("--no-interactive", action="store_true", help="skip the frame-by-frame (render + tonemap) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run with --nproc-per-node N")
        args.gpus = world

    dist = None
    torch = None
    force_dist = os.environ.get("PT_BENCH_FORCE_DIST") == "1"  # exercise the torch.distributed / RCCL plumbing with one rank
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from vk_raytrace_amd import capi, workloads
    from vk_raytrace_amd.renderer import HipRenderer
    from vk_raytrace_amd import shard

    t_setup = time.time()
    if args.workload == "c3":
        wl = workloads.c3_sponza(args.width, args.height, args.steps, tex_size=args.tex_size, target_tris=args.tris)
    else:
        wl = {"c2": workloads.c2_helmet, "c4": workloads.c4_sponza_4k, "c5": workloads.c5_bistro}[args.workload]()
        wl.name = wl.name.replace(f"{wl.spp}spp", f"{args.steps}spp")
        args.no_cpu_baseline = True  # the CPU leg and its algorithmic-byte model are sized for the bench line only
    wl.scene.finalize(capi.pack_vertices)
    W, H = wl.width, wl.height

    r = HipRenderer()
    r.setup(local_rank)
    if args.accel == "two":
        r.set_accel_mode(capi.PT_ACCEL_TWO_LEVEL)
    if args.emulate_shard:
        er, en = (int(x) for x in args.emulate_shard.split("/"))
        r.set_shard(er, en)
        args.no_cpu_baseline = True
    else:
        r.set_shard(rank, world)
    r.set_scene(wl.scene)
    integral, _ = r.set_env(wl.env)
    cam = capi.camera_lookat(wl.scene.camera, W / H, nb_lights=len(wl.scene.lights))
    r.set_camera(cam)
    from vk_raytrace_amd import host_device as hd
    r.set_sunsky(hd.default_sun_and_sky())
    r.create((W, H))
    st = hd.default_rtx_state()
    st.size[0], st.size[1] = W, H
    st.maxDepth, st.pbrMode, st.maxSamples = wl.depth, wl.pbr_mode, 1
    st.fireflyClampThreshold = 4.0 * integral
    t_setup = time.time() - t_setup

    def sync():
        r.synchronize()
        if torch is not None:
            torch.cuda.synchronize()
            dist.barrier()

    frame = 0
    for _ in range(args.warmup):
        st.frame = frame
        r.setPushContants(st)
        r.run()
        frame += 1
    sync()
    r.reset_stats()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.frame = frame
        r.setPushContants(st)
        r.run()
        frame += 1
    r.synchronize()
    if torch is not None:
        torch.cuda.synchronize()
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = r.stats()

    # the one collective of the path (untimed, reported): libptmi's own RCCL gather behind the C ABI (pt_gather_shards / pt_gather_finish)
    if world > 1 or force_dist:
        gatherer = shard.NativeGather(rank, world, local_rank, dist)
        r.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        img = gatherer.gather(r)
        gather_ms = (time.perf_counter() - t0) * 1e3
        gatherer.close()
    else:
        t0 = time.perf_counter()
        img = r.read_accum()
        gather_ms = (time.perf_counter() - t0) * 1e3

    if dist is not None:
        # whole-job counters
        keys = ["closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups"]
        v = torch.tensor([float(stats[k]) for k in keys], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        for i, k in enumerate(keys):
            stats[k] = int(v[i].item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    samples = W * H * args.steps
    value = samples / elapsed / 1e6
    out = {
        "metric": "Msamples/s on Sponza 1920x1080; per-pixel L2 vs ref at equal spp",
        "value": value,
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl.name, "triangles": wl.scene.num_triangles, "materials": len(wl.scene.materials), "textures": len(wl.scene.textures),
                   "width": W, "height": H, "spp": args.steps, "max_depth": wl.depth, "bsdf": "disney", "env": f"{wl.env.shape[1]}x{wl.env.shape[0]} procedural HDR",
                   "parallelism": f"image tiles {hd.TILE}x{hd.TILE} over {world} GPU(s), scene replicated"},
        "setup_s": t_setup,
        "gather_ms": gather_ms,
        "bvh_build_ms": stats["msBuildAccel"],
        "accel": {"mode": args.accel, "bytes": stats["bytesAccel"], "blas": stats["numBlas"], "tlas_nodes": stats["numTlasNodes"], "nodes": stats["numBvhNodes"]},
        "rays": {k: stats[k] for k in ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups")},
        "image_mean": float(np.mean(img[..., :3])) if img is not None else None,
    }

    if args.refit > 0:
        # instance update (pt_update_instances): every node's world matrix is re-sent; two-level mode redoes only the instance boxes + TLAS
        nodes = wl.scene.node_array()
        r.update_instances(nodes)
        t0 = time.perf_counter()
        for _ in range(args.refit):
            r.update_instances(nodes)
        out["accel"]["update_instances_ms"] = (time.perf_counter() - t0) * 1e3 / args.refit

    # ---- CPU baseline (oracle == literal restatement of pathtrace.comp, kind "port") + algorithmic bytes ----
    alg = None
    if world == 1 and not args.no_cpu_baseline:
        from tests import orc
        o = orc.Oracle()
        o.set_scene(wl.scene)
        o.set_env(wl.env)
        o.set_camera(cam)
        o.set_sunsky(hd.default_sun_and_sky())
        # bounded sample of the same workload: every 16th 8x8 pixel block, frames 0..F-1
        bx, by = (W + 7) // 8, (H + 7) // 8
        blocks = np.arange(bx * by)[::16]
        xs = (blocks % bx)[:, None, None] * 8 + np.arange(8)[None, None, :]
        ys = (blocks // bx)[:, None, None] * 8 + np.arange(8)[None, :, None]
        ok = (xs < W) & (ys < H)
        ids = (ys * W + xs)[np.broadcast_to(ok, (len(blocks), 8, 8))].astype(np.uint32)
        acc = np.zeros((H, W, 4), np.float32)
        ost = hd.default_rtx_state()
        ost.size[0], ost.size[1] = W, H
        ost.maxDepth, ost.pbrMode, ost.maxSamples = wl.depth, wl.pbr_mode, 1
        ost.fireflyClampThreshold = 4.0 * integral
        t0 = time.perf_counter()
        frames = 0
        while frames < 2 or (time.perf_counter() - t0 < args.cpu_seconds and frames < args.steps):
            ost.frame = frames
            o.render_frame(ost, acc, ids)
            frames += 1
        cpu_t = time.perf_counter() - t0
        os_ = o.stats()
        cores = os.cpu_count()
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        out["cpu_baseline"] = {"value": len(ids) * frames / cpu_t / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
                               "sample": f"CPU oracle (restatement of pathtrace.comp, OpenMP), every 16th 8x8 pixel block of the same {W}x{H} workload ({len(ids)} pixels), frames 0..{frames - 1}, {cpu_t:.1f} s"}
        cr, sr = max(1, os_["closestRays"]), max(1, os_["shadowRays"])
        alg = {
            "nodes_per_closest_ray": (os_["nodesVisited"] - os_["nodesShadow"]) / cr,
            "tris_per_closest_ray": (os_["trisTested"] - os_["trisShadow"]) / cr,
            "nodes_per_shadow_ray": os_["nodesShadow"] / sr,
            "tris_per_shadow_ray": os_["trisShadow"] / sr,
            "tex_taps_per_hit": os_["texTaps"] / max(1, os_["shadedHits"]),
        }
    elif os.path.exists(os.path.join(ROOT, "profiles", "alg_bytes_c3.json")):
        alg = json.load(open(os.path.join(ROOT, "profiles", "alg_bytes_c3.json")))
    if alg is not None:
        rays = out["rays"]
        b_total = (rays["closestRays"] * (alg["nodes_per_closest_ray"] * 32 + alg["tris_per_closest_ray"] * 36) + rays["shadowRays"] * (alg["nodes_per_shadow_ray"] * 32 + alg["tris_per_shadow_ray"] * 36)
                   + rays["shadedHits"] * (348 + 16 * alg["tex_taps_per_hit"]) + rays["alphaTests"] * 340 + rays["neeLookups"] * 80 + rays["misses"] * 64 + samples * 32)
        out["alg_bytes_per_sample"] = b_total / samples
        out["alg_model"] = alg

    # ---- measurement passes after the timed region (never part of `value`) ---------------------------------------------------------
    # (1) ceilings measured on this box: VALU issue (independent wave64 v_fmac_f32, 8 waves/SIMD on every CU) and HBM streaming
    peaks = r.measure_peaks()
    out["calibration"] = {"valu_G_wave_instr_per_s": peaks["valuWaveInstrPerSec"] / 1e9, "hbm_copy_GBps": peaks["hbmCopyBytesPerSec"] / 1e9,
                          "hbm_read_GBps": peaks["hbmReadBytesPerSec"] / 1e9, "compute_units": peaks["computeUnits"], "clock_MHz": peaks["clockMHz"],
                          "note": "pt_measure_peaks on this device; theoretical VALU issue = CUs x 4 SIMDs x clock / 2 cycles (MI355X_MICROARCH.md), tools/valu_peak.hip shows 0.45-0.8 of it depending on the instruction form"}
    # (2) the interactive path: SampleExample's loop tonemaps after every frame, so every frame is its own batch of one
    if not args.no_interactive and world == 1:
        tm = hd.default_tonemapper()
        n_i = 24
        for _ in range(4):
            st.frame = frame; r.setPushContants(st); r.run(); r.tonemap(tm); frame += 1
        t0 = time.perf_counter()
        for _ in range(n_i):
            st.frame = frame; r.setPushContants(st); r.run(); r.tonemap(tm); frame += 1
        ti = time.perf_counter() - t0
        out["interactive"] = {"value": W * H * n_i / ti / 1e6, "unit": "Msamples/s", "ms_per_frame": ti / n_i * 1e3, "frames": n_i,
                              "note": "render + pt_tonemap (RGBA8 read back to the host) per frame: batch = 1, like SampleExample's display loop"}
    # (3) standalone kernel durations: the timed loop overlaps four launch sequences on separate streams, so a HIP-event bracket there is not a
    # kernel's own duration.  One batch is rendered again on a second context with ONE frame slot (PT_TUNE inflight=1): nothing overlaps,
    # HIP events on the launching stream bracket each stage.
    serial = None
    if not args.no_profile and not args.emulate_shard:
        # tail=0: every bounce goes through the staged kernels, so that a stage's time and its rays belong together (the timed run hands the
        # late, small bounces to the fused k_tail: stats["msTail"])
        os.environ["PT_TUNE"] = (os.environ.get("PT_TUNE", "") + ",inflight=1,tail=0").lstrip(",")
        r2 = HipRenderer()
        r2.setup(local_rank)
        r2.set_shard(rank, world)
        r2.set_scene(wl.scene)
        r2.set_env(wl.env)
        r2.set_camera(cam)
        r2.set_sunsky(hd.default_sun_and_sky())
        r2.create((W, H))
        nser = min(args.steps, 32)
        for phase in range(2):  # warm-up batch, then the measured one
            r2.reset_stats()
            r2.set_profiling(phase == 1)
            t0 = time.perf_counter()
            for f in range(nser):
                st.frame = f; r2.setPushContants(st); r2.run()
            r2.synchronize()
            tser = time.perf_counter() - t0
        s2 = r2.stats()
        r2.set_profiling(False)
        r2.destroy()
        nsamp = (W * H * nser) / max(1, world)  # samples of the measured batch on this GPU (image tiles are split evenly)
        serial = {"frames": nser, "wall_ms": tser * 1e3, "launches_per_stage": int(s2["launchesTraceClosest"]),
                  "stage_ms": {"generate": s2["msGenerate"], "closest": s2["msTraceClosest"], "shade": s2["msShade"], "shadow": s2["msTraceShadow"], "accumulate": s2["msAccumulate"]},
                  "rays": {k: s2[k] for k in ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups")}, "samples": nsamp}
        out["serialised"] = serial

    # ---- roofline of the dominant kernel: chosen by its standalone time, priced on algorithmic bytes --------------------
    if alg is not None and serial is not None:
        sr = serial["rays"]
        stage_bytes = {
            "closest": sr["closestRays"] * (alg["nodes_per_closest_ray"] * 32 + alg["tris_per_closest_ray"] * 36) + sr["alphaTests"] * 340 * (sr["closestRays"] / max(1, sr["closestRays"] + sr["shadowRays"])),
            "shadow": sr["shadowRays"] * (alg["nodes_per_shadow_ray"] * 32 + alg["tris_per_shadow_ray"] * 36) + sr["alphaTests"] * 340 * (sr["shadowRays"] / max(1, sr["closestRays"] + sr["shadowRays"])),
            "shade": sr["shadedHits"] * (348 + 16 * alg["tex_taps_per_hit"]) + sr["neeLookups"] * 80 + sr["misses"] * 64,
            "generate": 0.0,
            "accumulate": serial["samples"] * 32.0,
        }
        kernels = {"closest": "k_closest_k + k_closest_p (+ k_closest_x)", "shadow": "k_shadow_p (+ k_shadow_x)", "shade": "k_shade", "generate": "k_generate", "accumulate": "k_accumulate"}
        launches = max(1, serial["launches_per_stage"])
        table = {}
        for k, ms in serial["stage_ms"].items():
            n_l = launches if k in ("closest", "shade", "shadow") else max(1, launches // max(1, wl.depth))
            table[k] = {"kernel": kernels[k], "ms": ms, "launches": n_l, "avg_launch_ms": ms / n_l, "alg_bytes": stage_bytes[k],
                        "alg_GBps": (stage_bytes[k] / (ms * 1e-3) / 1e9) if ms > 0 else None}
        out["stages_serialised"] = table
        dom = max(("closest", "shade", "shadow"), key=lambda k: serial["stage_ms"][k])
        d = table[dom]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r02_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                per_sample = tj["hbm_bytes_per_sample"].get(dom)
                traffic = per_sample * serial["samples"] / d["launches"] if per_sample else None
                out["hbm_measured"] = {"bytes_per_sample": tj["hbm_bytes_per_sample"]["total"], "GBps": tj["hbm_bytes_per_sample"]["total"] * samples / max(1, world) / elapsed / 1e9,
                                       "frac": tj["hbm_bytes_per_sample"]["total"] * samples / max(1, world) / elapsed / 1e9 / HBM_PEAK_GBS,
                                       "source": "profiles/r02_traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of tools/pmc_r02.sh, gfx950 corrections) x this run's rate",
                                       "note": "what actually crosses the HBM interface per sample, against the 8 TB/s peak: the scene's working set lives in L2 / Infinity Cache"}
            except Exception:
                traffic = None
        # `achieved` is what the stage moves across the HBM interface (measured, PMC) per second of its own standalone run time; the
        # algorithmic bytes (reference-layout BVH2 visits x 32 B, triangle tests x 36 B, material / texture / light records)
        # are reported next to it -- most of them are served by L2 / Infinity Cache (the whole scene + BVH is ~115 MB), which is why the
        # algorithmic rate can exceed the HBM peak while the interface is far from saturated.
        ach = (traffic / (d["avg_launch_ms"] * 1e-3) / 1e9) if traffic else None
        out["roofline"] = {"bound": "hbm", "kernel": d["kernel"], "stage": dom, "achieved": ach if ach is not None else d["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ((ach if ach is not None else d["alg_GBps"]) / HBM_PEAK_GBS), "traffic": traffic,
                           "achieved_basis": "measured HBM bytes per launch (profiles/r02_traffic.json) / standalone launch duration" if ach is not None else "algorithmic bytes / standalone launch duration",
                           "alg_bytes_per_launch": d["alg_bytes"] / d["launches"], "alg_GBps": d["alg_GBps"], "alg_frac": d["alg_GBps"] / HBM_PEAK_GBS if d["alg_GBps"] else None,
                           "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"], "measured_hbm_copy_GBps": out["calibration"]["hbm_copy_GBps"],
                           "note": "dominant stage = largest STANDALONE time (serialised pass of this run: one frame slot, nothing overlapped, HIP events on the launching stream; "
                                   "profiles/r02_trace_batch.txt holds the rocprofv3 per-dispatch durations of the same kind of batch)"}
    # ---- VALU issue: instruction counts per sample are a property of the code and the workload (rocprofv3 PMC pass of this round,
    # profiles/r02_valu.json); the rate is this run's; the ceiling is the one measured above on this box.
    vpath = os.path.join(ROOT, "profiles", "r02_valu.json")
    if args.workload == "c3" and os.path.exists(vpath):
        try:
            per_sample = json.load(open(vpath))["valu_wave_instr_per_sample"]
            peak = out["calibration"]["valu_G_wave_instr_per_s"] * 1e9
            ach = per_sample * samples / elapsed / max(1, world)
            out["issue_roofline"] = {"bound": "valu", "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s per GPU", "frac": ach / peak,
                                     "valu_wave_instr_per_sample": per_sample, "source": "profiles/r02_valu.json x this run's rate / this run's calibration"}
        except Exception:
            pass
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
