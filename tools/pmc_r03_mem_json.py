"""mempipe.json from the passes of tools/pmc_r03_mem.sh: per kernel, the raw sums and
  ta_busy_frac        TA_TA_BUSY summed over the TAs / (256 TAs x GRBM_GUI_ACTIVE of the same dispatches)
  l1_wave_latency     TCP_TCP_LATENCY / TCP_TA_TCP_STATE_READ  (cycles a vector-memory wave instruction spends in the L1)
  l2_read_latency     TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ  (cycles of one L1 -> L2 read round trip)
  l1_pending_frac     TCP_PENDING_STALL_CYCLES / TCP_GATE_EN2  (share of the L1's clocked cycles spent stalled on data pending from L2)
  tlb_miss_rate       TCP_UTCL1_TRANSLATION_MISS / TCP_UTCL1_REQUEST"""
import collections
import csv
import json
import os
import sys

out, frames = sys.argv[1], int(sys.argv[2])
KERNELS = ("k_generate", "k_closest_k", "k_closest_p", "k_closest_x", "k_shade", "k_shadow_p", "k_shadow_x", "k_accumulate", "k_tail")


def kernel_of(name):
    for k in KERNELS:
        if k + "<" in name or k + "(" in name or name.endswith(k):
            return k
    return None


tot = collections.defaultdict(lambda: collections.defaultdict(float))
for i in range(1, 10):
    p = os.path.join(out, f"counters{i}.csv")
    if not os.path.exists(p):
        continue
    seen = set()
    for r in csv.DictReader(open(p)):
        k = kernel_of(r["Kernel_Name"])
        if k:
            c = r["Counter_Name"]
            key = f"{c}@{i}" if c == "GRBM_GUI_ACTIVE" else c
            tot[k][key] += float(r["Counter_Value"])


def ratio(d, a, b, scale=1.0):
    return d[a] / (d[b] * scale) if d.get(b) else None


res = {"frames": frames, "pipeline": "timed launch policy, one frame slot (PT_TUNE=inflight=1)", "kernels": {}}
for k, d in tot.items():
    res["kernels"][k] = {
        "ta_busy_frac": ratio(d, "TA_TA_BUSY_sum", "GRBM_GUI_ACTIVE@1", 256.0),
        "ta_addr_stalled_by_tc_frac": ratio(d, "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "GRBM_GUI_ACTIVE@1", 256.0),
        "ta_data_stalled_by_tc_frac": ratio(d, "TA_DATA_STALLED_BY_TC_CYCLES_sum", "GRBM_GUI_ACTIVE@1", 256.0),
        "l1_wave_latency_cycles": ratio(d, "TCP_TCP_LATENCY_sum", "TCP_TA_TCP_STATE_READ_sum"),
        "l2_read_latency_cycles": ratio(d, "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"),
        "l1_pending_frac": ratio(d, "TCP_PENDING_STALL_CYCLES_sum", "TCP_GATE_EN2_sum"),
        "l1_ta_data_stall_frac": ratio(d, "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_GATE_EN2_sum"),
        "l1_accesses_per_clocked_cycle": ratio(d, "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_GATE_EN2_sum"),
        "tlb_miss_rate": ratio(d, "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum"),
        "raw": dict(d),
    }
json.dump(res, open(os.path.join(out, "mempipe.json"), "w"), indent=1)
for k, v in res["kernels"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "raw"})
