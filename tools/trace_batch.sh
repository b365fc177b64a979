#!/bin/bash
# GPU box: per-dispatch kernel durations of ONE serialised 32-frame batch (rocprofv3 kernel trace; PT_TUNE inflight=1 so that the batches do
# not overlap), grouped by kernel in bounce order: where inside a batch the time goes.
# usage: tools/trace_batch.sh <tag>
TAG=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
PT_TUNE=inflight=1,$PT_TUNE timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- python $REPO/bench.py --steps 64 --warmup 0 --no-cpu-baseline --no-profile --no-interactive > $OUT/bench.json 2>/dev/null
find $OUT/raw -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/raw
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["k_generate", "k_closest_k", "k_closest_s", "k_closest_p", "k_closest_x", "k_shade", "k_shadow_s", "k_shadow_p", "k_shadow_x", "k_accumulate", "k_raysort"]
batches, cur = [], None
for r in rows:
    k = next((n for n in names if n in r["Kernel_Name"]), None)
    if k is None: continue
    if k == "k_generate":
        cur = []; batches.append(cur)
    if cur is not None:
        cur.append((k, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
f = batches[-1]
line = {}
for k, us in f:
    line.setdefault(k, []).append(us)
tot = 0
for k in names:
    if k in line:
        tot += sum(line[k])
        print("%-14s" % k, " ".join("%7.0f" % v for v in line[k]), "  sum %.2f ms" % (sum(line[k]) / 1e3))
print("batch total %.2f ms over %d batches traced" % (tot / 1e3, len(batches)))
PY
