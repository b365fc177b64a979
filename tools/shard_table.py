"""GPU box: predicted strong scaling from EVERY rank's shard (not rank 0's alone), in one process.
For N in 1, 2, 4, 8 every rank r of N renders its tiles of the workload alone on this one GPU (pt_set_shard(r, N)); the job's time is the SLOWEST rank's
(max over ranks of the median of `--repeats` windows of `--steps` frames).  What one GPU cannot show -- eight processes sharing a host, the xGMI gather --
is stated next to the table: the gather is not in the timed loop of the metric (bench.py times the frame loop, gather_ms is reported separately).
   python tools/shard_table.py --workload c3 --steps 20 128 [--repeats 5] > table.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from vk_raytrace_amd import capi, workloads, host_device as hd  # noqa: E402
from vk_raytrace_amd.renderer import HipRenderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3", choices=["c3", "c4"])
ap.add_argument("--steps", type=int, nargs="+", default=[20, 128])
ap.add_argument("--repeats", type=int, default=5)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
args = ap.parse_args()

wl = workloads.c3_sponza() if args.workload == "c3" else workloads.c4_sponza_4k()
wl.scene.finalize(capi.pack_vertices)
W, H = wl.width, wl.height
r = HipRenderer()
r.setup(0)
r.set_scene(wl.scene)
integral, _ = r.set_env(wl.env)
cam = capi.camera_lookat(wl.scene.camera, W / H, nb_lights=len(wl.scene.lights))
r.set_camera(cam)
r.set_sunsky(hd.default_sun_and_sky())
st = hd.default_rtx_state()
st.size[0], st.size[1] = W, H
st.maxDepth, st.pbrMode, st.maxSamples = wl.depth, wl.pbr_mode, 1
st.fireflyClampThreshold = 4.0 * integral
out = {"workload": wl.name, "width": W, "height": H, "repeats": args.repeats, "rows": []}
for steps in args.steps:
    t1 = None
    for n in args.ranks:
        per_rank = []
        for rank in range(n):
            r.set_shard(rank, n)
            r.create((W, H))
            frame = 0
            for _ in range(args.warmup):
                st.frame = frame; r.setPushContants(st); r.run(); frame += 1
            r.synchronize()
            wins = []
            for _ in range(args.repeats):
                t0 = time.perf_counter()
                for _ in range(steps):
                    st.frame = frame; r.setPushContants(st); r.run(); frame += 1
                r.synchronize()
                wins.append(time.perf_counter() - t0)
            per_rank.append(float(np.median(wins)) / steps * 1e3)
        worst = max(per_rank)
        if n == 1:
            t1 = worst
        row = {"steps": steps, "ranks": n, "ms_per_frame_slowest_rank": worst, "ms_per_frame_per_rank": per_rank, "imbalance_max_over_mean": worst / (sum(per_rank) / len(per_rank)),
               "efficiency": (t1 / (n * worst)) if t1 else None, "msamples_per_s_job": W * H / (worst * 1e-3) / 1e6}
        out["rows"].append(row)
        print(f"{args.workload} steps {steps:4d} N={n}: slowest rank {worst:.4f} ms/frame, imbalance {row['imbalance_max_over_mean']:.3f}, efficiency {row['efficiency']:.3f}", file=sys.stderr)
# the one collective, priced: shard bytes per peer over its own xGMI link (MI355X_MICROARCH.md: ~153 GB/s per link peak; half of it assumed)
out["gather_estimate"] = {"bytes_per_peer_at_8": W * H * 16 / 8, "ms_at_75_GBps_per_link": W * H * 16 / 8 / 75e9 * 1e3,
                          "note": "one grouped RCCL send / recv, every peer on its own link; outside the timed frame loop like the reference metric (bench.py gather_ms)"}
print(json.dumps(out))
