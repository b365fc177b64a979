#!/usr/bin/env python
"""Headline benchmark: Msamples/s on the Sponza-class configuration of BASELINE.json (C3: 1920x1080,
depth 8, Disney BSDF, HDR environment; 256 spp == 256 steps).

A "step" is one frame: one sample per pixel over the whole image (maxSamples = 1, the reference default),
accumulated into the RGBA32F framebuffer exactly like shaders/pathtrace.comp:122-133.  Scene, BVH, textures
and environment are resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps 256 --warmup 8
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

With N > 1 the image tiles are sharded over the ranks (vk_raytrace_amd/shard.py); the ranks do not
communicate while rendering and the single RCCL framebuffer gather happens after the timed loop (its time is
reported as gather_ms, like the reference metric which times the frame loop only - BASELINE.md section 2).
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events (roofline fields become null)")
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
    r.set_profiling(not args.no_profile)

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
    r.set_profiling(False)

    # the one collective of the path (untimed, reported)
    t0 = time.perf_counter()
    img = shard.gather_framebuffer(r, rank, world, f"cuda:{local_rank}" if dist is not None else None, force=force_dist)
    gather_ms = (time.perf_counter() - t0) * 1e3

    if dist is not None:
        # whole-job counters
        keys = ["closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups"]
        v = torch.tensor([float(stats[k]) for k in keys] + [stats["msTraceClosest"], stats["msShade"], stats["msTraceShadow"]], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        for i, k in enumerate(keys):
            stats[k] = int(v[i].item())
        # kernel time: mean over ranks (each rank runs its own launches)
        stats["msTraceClosest"], stats["msShade"], stats["msTraceShadow"] = (float(v[len(keys) + i].item()) / world for i in range(3))

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
        "rays": {k: stats[k] for k in ("closestRays", "shadowRays", "shadedHits", "misses", "alphaTests", "neeLookups")},
        "image_mean": float(np.mean(img[..., :3])) if img is not None else None,
    }

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

    # ---- roofline of the dominant kernel (k_closest: BVH traversal + ray/triangle tests) ----
    if alg is not None:
        rays = out["rays"]
        bytes_closest = rays["closestRays"] * (alg["nodes_per_closest_ray"] * 32 + alg["tris_per_closest_ray"] * 36)
        bytes_shadow = rays["shadowRays"] * (alg["nodes_per_shadow_ray"] * 32 + alg["tris_per_shadow_ray"] * 36)
        b_total = (bytes_closest + bytes_shadow + rays["shadedHits"] * (348 + 16 * alg["tex_taps_per_hit"]) + rays["alphaTests"] * 340 + rays["neeLookups"] * 80
                   + rays["misses"] * 64 + samples * 32)
        out["alg_bytes_per_sample"] = b_total / samples
        out["alg_GBps_whole_pipeline"] = b_total / elapsed / 1e9
        out["alg_model"] = alg
        ms_c = stats["msTraceClosest"]
        launches = max(1, stats["launchesTraceClosest"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r1.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_closest_bytes_per_launch")
            except Exception:
                traffic = None
        if ms_c > 0:
            achieved = bytes_closest / max(1, world) / (ms_c * 1e-3) / 1e9  # per GPU: whole-job bytes / ranks over the mean per-rank stage time
            out["roofline"] = {"bound": "hbm", "kernel": "k_closest", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                               "traffic": traffic, "alg_bytes_per_launch": bytes_closest / max(1, world) / launches, "avg_launch_ms": ms_c / launches, "launches": launches,
                               "note": "algorithmic bytes (reference-layout BVH2 visits x 32 B + triangle tests x 36 B of this stage's rays) are served from L2 / Infinity Cache: "
                                       "`traffic` is the measured HBM bytes per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE passes), so `frac` can exceed 1; "
                                       "the binding resource is VALU issue -- see issue_roofline"}
        else:
            out["roofline"] = None
    # ---- what actually bounds the path: VALU issue.  Instruction counts per sample are a property of the code and the workload
    # (rocprofv3 PMC pass, profiles/valu_r1.json); the rate is this run's.  Peak: 256 CUs x 4 SIMDs, one wave64 VALU instruction per
    # 4 cycles and SIMD, 2.4 GHz (MI355X_MICROARCH.md).
    vpath = os.path.join(ROOT, "profiles", "valu_r1.json")
    if args.workload == "c3" and os.path.exists(vpath):
        try:
            per_sample = json.load(open(vpath))["valu_wave_instr_per_sample"]
            peak = 256 * 4 * 2.4e9 / 4
            ach = per_sample * samples / elapsed / max(1, world)
            out["issue_roofline"] = {"bound": "valu", "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s per GPU", "frac": ach / peak,
                                     "valu_wave_instr_per_sample": per_sample, "source": "profiles/valu_r1.json"}
        except Exception:
            pass
    out["stage_ms"] = {k: stats[k] for k in ("msGenerate", "msTraceClosest", "msShade", "msTraceShadow", "msAccumulate")}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
