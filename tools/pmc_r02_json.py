"""Turns the rocprofv3 PMC passes of tools/pmc_r02.sh into traffic.json (HBM bytes per sample, per stage and total) and valu.json
(VALU wave-instructions per sample, per kernel and total).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of 16-B-per-lane reads -> x2 (every hot load of this
code is a 16-byte load); WRITE_SIZE is taken as reported."""
import collections
import csv
import json
import os
import sys

out, frames = sys.argv[1], int(sys.argv[2])
STAGE = {"k_generate": "generate", "k_closest_k": "closest", "k_closest_p": "closest", "k_closest_s": "closest", "k_closest_x": "closest", "k_shade": "shade",
         "k_shadow_p": "shadow", "k_shadow_s": "shadow", "k_shadow_k": "shadow", "k_shadow_x": "shadow", "k_accumulate": "accumulate",
         "k_raysort_hist": "sort", "k_raysort_scan": "sort", "k_raysort_scatter": "sort"}


def kernel_of(name):
    for k in STAGE:
        if k + "<" in name or k + "(" in name or name.endswith(k):
            return k
    return None


def agg(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(int)
    if not os.path.exists(path):
        return tot, calls
    seen = set()
    for r in csv.DictReader(open(path)):
        k = kernel_of(r["Kernel_Name"])
        if not k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), r["Counter_Name"])
        if key not in seen:
            seen.add(key)
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES"):
                calls[k] += 1
    return tot, calls


bench = json.load(open(os.path.join(out, "bench1.json")))
samples = bench["config"]["width"] * bench["config"]["height"] * frames
f, fc = agg(os.path.join(out, "counters1.csv"))
w, wc = agg(os.path.join(out, "counters2.csv"))
v, vc = agg(os.path.join(out, "counters3.csv"))
traffic = {"units": "HBM-side bytes per sample: FETCH_SIZE x 1024 x 2 (gfx950 half-count of 16-B/lane reads) + WRITE_SIZE x 1024", "frames": frames, "samples": samples,
           "kernels": {}, "hbm_bytes_per_sample": collections.defaultdict(float)}
for k in sorted(set(f) | set(w)):
    rd, wr = f[k].get("FETCH_SIZE", 0.0) * 1024 * 2, w[k].get("WRITE_SIZE", 0.0) * 1024
    traffic["kernels"][k] = {"launches": fc.get(k, 0), "read_bytes": rd, "write_bytes": wr, "bytes_per_sample": (rd + wr) / samples}
    traffic["hbm_bytes_per_sample"][STAGE[k]] += (rd + wr) / samples
    traffic["hbm_bytes_per_sample"]["total"] += (rd + wr) / samples
traffic["hbm_bytes_per_sample"] = dict(traffic["hbm_bytes_per_sample"])
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
valu = {"units": "wave64 instructions per sample (SQ_INSTS_* summed over the dispatches of one 32-frame batch / samples)", "frames": frames, "samples": samples, "kernels": {}}
total = 0.0
for k in sorted(v):
    valu["kernels"][k] = {c: v[k][c] / samples for c in v[k]}
    total += v[k].get("SQ_INSTS_VALU", 0.0) / samples
valu["valu_wave_instr_per_sample"] = total
json.dump(valu, open(os.path.join(out, "valu.json"), "w"), indent=1)
print(json.dumps(traffic["hbm_bytes_per_sample"], indent=1))
print("valu wave-instr / sample:", total, {k: round(d.get("SQ_INSTS_VALU", 0), 1) for k, d in valu["kernels"].items()})
