"""Turns the rocprofv3 PMC passes of tools/pmc_passes.sh into
  traffic.json  HBM bytes per sample, per kernel / stage / total
  valu.json     VALU (and SALU / VMEM / LDS / SMEM) wave-instructions per sample, per kernel and total; SQ cycle counters where collected
  cache.json    L2 (TCC) requests, hits and misses per sample and the hit rate, per kernel / stage; L1 (TCP) accesses where collected
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of
the bytes of 16-B-per-lane reads -> x2 (every hot load of this code is a 16-byte load); WRITE_SIZE is taken as reported."""
import collections
import csv
import json
import os
import sys

out, frames = sys.argv[1], int(sys.argv[2])
STAGE = {"k_generate": "generate", "k_closest_k": "closest", "k_closest_p": "closest", "k_closest_s": "closest", "k_closest_x": "closest", "k_shade": "shade",
         "k_shadow_p": "shadow", "k_shadow_s": "shadow", "k_shadow_k": "shadow", "k_shadow_x": "shadow", "k_accumulate": "accumulate", "k_tail": "tail",
         "k_raysort_hist": "sort", "k_raysort_scan": "sort", "k_raysort_scatter": "sort"}


def kernel_of(name):
    for k in STAGE:
        if k + "<" in name or k + "(" in name or name.endswith(k):
            return k
    return None


def agg(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    if not os.path.exists(path):
        return tot, {}
    for r in csv.DictReader(open(path)):
        k = kernel_of(r["Kernel_Name"])
        if not k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r.get("Dispatch_Id"))
    return tot, {k: len(v) for k, v in calls.items()}


def samples_of(i):
    b = json.load(open(os.path.join(out, f"bench{i}.json")))
    return b["config"]["width"] * b["config"]["height"] * frames


passes = {}
for i in range(1, 10):
    p = os.path.join(out, f"counters{i}.csv")
    if os.path.exists(p):
        passes[i] = agg(p)


def find(counter):
    """(per-kernel totals, launches per kernel, samples) of the pass that collected `counter`"""
    for i, (tot, calls) in passes.items():
        if any(counter in d for d in tot.values()):
            try:
                return {k: d.get(counter, 0.0) for k, d in tot.items()}, calls, samples_of(i)
            except Exception:
                continue
    return None, None, None


def per_stage(per_kernel):
    s = collections.defaultdict(float)
    for k, v in per_kernel.items():
        s[STAGE[k]] += v
        s["total"] += v
    return dict(s)


# ---- HBM traffic
f, fc, fs = find("FETCH_SIZE")
w, wc, ws = find("WRITE_SIZE")
if f is not None and w is not None:
    traffic = {"units": "HBM-side bytes per sample: FETCH_SIZE x 1024 x 2 (gfx950 half-count of 16-B/lane reads) + WRITE_SIZE x 1024", "frames": frames, "samples": fs,
               "pipeline": "timed launch policy (k_tail takes the late bounces), one frame slot", "kernels": {}}
    pk = {}
    for k in sorted(set(f) | set(w)):
        rd, wr = f.get(k, 0.0) * 1024 * 2 / fs, w.get(k, 0.0) * 1024 / ws
        traffic["kernels"][k] = {"launches": fc.get(k, 0), "read_bytes_per_sample": rd, "write_bytes_per_sample": wr, "bytes_per_sample": rd + wr}
        pk[k] = rd + wr
    traffic["hbm_bytes_per_sample"] = per_stage(pk)
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("HBM bytes / sample:", json.dumps({k: round(v, 1) for k, v in traffic["hbm_bytes_per_sample"].items()}))

# ---- instruction mix
v, vc, vs = find("SQ_INSTS_VALU")
if v is not None:
    valu = {"units": "wave64 instructions per sample (SQ_INSTS_* summed over the dispatches of the run / samples)", "frames": frames, "samples": vs,
            "pipeline": "timed launch policy (k_tail takes the late bounces), one frame slot", "kernels": {}}
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAVES"):
        d, _, s = find(name)
        if d is None:
            continue
        for k, x in d.items():
            valu["kernels"].setdefault(k, {})[name] = x / s
    valu["valu_wave_instr_per_sample"] = sum(x / vs for x in v.values())
    valu["valu_wave_instr_per_sample_by_stage"] = per_stage({k: x / vs for k, x in v.items()})
    cyc = {}
    for name in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        d, _, s = find(name)
        if d is not None:
            for k, x in d.items():
                cyc.setdefault(k, {})[name] = x / s
    if cyc:
        valu["sq_cycles_per_sample"] = cyc
    json.dump(valu, open(os.path.join(out, "valu.json"), "w"), indent=1)
    print("VALU wave-instr / sample:", round(valu["valu_wave_instr_per_sample"], 1), {k: round(d.get("SQ_INSTS_VALU", 0), 1) for k, d in valu["kernels"].items()})

# ---- caches
h, hc, hs = find("TCC_HIT_sum")
m, mc, ms = find("TCC_MISS_sum")
if h is not None and m is not None:
    cache = {"units": "L2 (TCC) requests per sample; one request = one 128-byte line", "frames": frames, "samples": hs, "kernels": {}}
    req_k, hit_k, miss_k = {}, {}, {}
    for k in sorted(set(h) | set(m)):
        hit_k[k], miss_k[k] = h.get(k, 0.0) / hs, m.get(k, 0.0) / ms
        req_k[k] = hit_k[k] + miss_k[k]
        cache["kernels"][k] = {"l2_hits_per_sample": hit_k[k], "l2_misses_per_sample": miss_k[k], "l2_hit_rate": hit_k[k] / req_k[k] if req_k[k] else None}
    cache["l2_requests_per_sample"] = per_stage(req_k)
    hs_, ms_ = per_stage(hit_k), per_stage(miss_k)
    cache["l2_hit_rate"] = {k: (hs_[k] / (hs_[k] + ms_[k]) if hs_[k] + ms_[k] else None) for k in hs_}
    for name, key in (("TCC_REQ_sum", "tcc_req_per_sample"), ("TCC_EA0_RDREQ_sum", "tcc_ea_rdreq_per_sample"), ("TCP_TOTAL_CACHE_ACCESSES_sum", "l1_accesses_per_sample"),
                      ("TCP_TCC_READ_REQ_sum", "l1_to_l2_read_req_per_sample")):
        d, _, s = find(name)
        if d is not None:
            cache[key] = per_stage({k: x / s for k, x in d.items()})
    json.dump(cache, open(os.path.join(out, "cache.json"), "w"), indent=1)
    print("L2 requests / sample:", json.dumps({k: round(v, 1) for k, v in cache["l2_requests_per_sample"].items()}), "hit rate", json.dumps({k: (round(v, 3) if v is not None else None) for k, v in cache["l2_hit_rate"].items()}))
