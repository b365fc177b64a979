"""Summarises the passes of tools/pmc_binders.sh into binders.json: per kernel, the hardware figures that say what binds it.
  lanes_per_valu_instr   SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU x (64 / ... ) -- see `lane_note`: both raw counters and the two ratios are kept
  icache                 SQC_ICACHE_REQ / HITS / MISSES (+ duplicates) per sample and the miss rate
  ifetch                 SQ_IFETCH per sample, SQ_IFETCH_LEVEL / SQ_BUSY_CYCLES
  vmem                   SQ_INST_CYCLES_VMEM* and SQ_ACTIVE_INST_VMEM against SQ_WAVE_CYCLES
  valu_classes           SQ_INSTS_VALU_* per sample
  waits                  SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY shares of SQ_WAVE_CYCLES
Every number is `counter summed over the run's dispatches of the kernel / samples of the run` unless it says "share"."""
import collections
import csv
import json
import os
import sys

out, frames = sys.argv[1], int(sys.argv[2])
KERNELS = ("k_generate", "k_closest_k", "k_closest_p", "k_closest_q", "k_closest_x", "k_shade", "k_shadow_p", "k_shadow_q", "k_shadow_x", "k_trace_p", "k_trace_x", "k_tail", "k_accumulate")


def kernel_of(name):
    for k in KERNELS:
        if k + "<" in name or k + "(" in name or name.endswith(k):
            return k
    return None


per = collections.defaultdict(dict)  # kernel -> counter -> per-sample value
raw_passes = {}
for i in range(21, 40):
    p = os.path.join(out, f"counters{i}.csv")
    if not os.path.exists(p):
        continue
    try:
        b = json.load(open(os.path.join(out, f"bench{i}.json")))
        samples = b["config"]["width"] * b["config"]["height"] * frames
    except Exception as e:  # the bench of this pass did not print its line
        print(f"pass {i}: no bench line ({e})")
        continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(p)):
        k = kernel_of(r["Kernel_Name"])
        if k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in tot.items():
        for c, v in d.items():
            per[k][c] = v / samples
    raw_passes[i] = sorted({c for d in tot.values() for c in d})


def ratio(a, b):
    return (a / b) if (a is not None and b) else None


res = {"units": "counter / samples of the run (frames x pixels) unless a field says share or ratio", "frames": frames, "passes": raw_passes,
       "lane_note": "SQ_THREAD_CYCLES_VALU = sum over VALU instructions of active lanes x cycles (quad-cycles, like SQ_ACTIVE_INST_VALU); "
                    "active lanes per VALU instruction = 64 x SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) when the counter ticks per lane, "
                    "so `lanes_per_valu_instr` = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, clamped interpretation: <= 64",
       "kernels": {}}
for k in KERNELS:
    d = per.get(k)
    if not d:
        continue
    g = d.get
    e = {"counters_per_sample": {c: d[c] for c in sorted(d)}}
    e["lanes_per_valu_instr"] = ratio(g("SQ_THREAD_CYCLES_VALU"), g("SQ_ACTIVE_INST_VALU"))
    e["lanes_per_valu_instr_by_insts"] = ratio(g("SQ_THREAD_CYCLES_VALU"), g("SQ_INSTS_VALU"))
    e["active_valu_quadcycles_per_instr"] = ratio(g("SQ_ACTIVE_INST_VALU"), g("SQ_INSTS_VALU"))
    e["valu_active_share_of_wave_cycles"] = ratio(g("SQ_ACTIVE_INST_VALU"), g("SQ_WAVE_CYCLES"))
    e["icache_miss_rate"] = ratio(g("SQC_ICACHE_MISSES"), g("SQC_ICACHE_REQ"))
    e["icache_miss_rate_incl_duplicates"] = ratio((g("SQC_ICACHE_MISSES") or 0) + (g("SQC_ICACHE_MISSES_DUPLICATE") or 0), g("SQC_ICACHE_REQ"))
    e["ifetch_level_per_busy_cycle"] = ratio(g("SQ_IFETCH_LEVEL"), g("SQ_BUSY_CYCLES"))
    e["wait_any_share"] = ratio(g("SQ_WAIT_ANY"), g("SQ_WAVE_CYCLES"))
    e["wait_inst_any_share"] = ratio(g("SQ_WAIT_INST_ANY"), g("SQ_WAVE_CYCLES"))
    e["active_inst_any_share"] = ratio(g("SQ_ACTIVE_INST_ANY"), g("SQ_WAVE_CYCLES"))
    e["vmem_active_share"] = ratio(g("SQ_ACTIVE_INST_VMEM"), g("SQ_WAVE_CYCLES"))
    e["dcache_miss_rate"] = ratio(g("SQC_DCACHE_MISSES"), g("SQC_DCACHE_REQ"))
    res["kernels"][k] = e
json.dump(res, open(os.path.join(out, "binders.json"), "w"), indent=1)
for k, e in res["kernels"].items():
    fmt = lambda x: "-" if x is None else f"{x:.3f}"
    print(f"{k:13s} lanes/VALU {fmt(e['lanes_per_valu_instr'])}  quadcyc/VALU {fmt(e['active_valu_quadcycles_per_instr'])}  I$ miss {fmt(e['icache_miss_rate'])} (+dup {fmt(e['icache_miss_rate_incl_duplicates'])})"
          f"  wait {fmt(e['wait_any_share'])}  wait_inst {fmt(e['wait_inst_any_share'])}  valu_active {fmt(e['valu_active_share_of_wave_cycles'])}")
