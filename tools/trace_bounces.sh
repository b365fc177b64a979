#!/bin/bash
# GPU box: per-dispatch kernel durations of one frame (rocprofv3 kernel trace), grouped by kernel and bounce order.
# usage: PT_TUNE=... tools/trace_bounces.sh <tag>
TAG=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- python $REPO/tools/gpu_stats.py 3 > $OUT/stats.txt 2>/dev/null
find $OUT/raw -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/raw
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["k_generate", "k_closest_s", "k_closest_p", "k_closest_x", "k_shade", "k_shadow_s", "k_shadow_p", "k_shadow_x", "k_accumulate"]
frames = []
cur = None
for r in rows:
    k = next((n for n in names if n + "(" in r["Kernel_Name"]), None)
    if k is None: continue
    if k == "k_generate":
        cur = []; frames.append(cur)
    if cur is not None:
        cur.append((k, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
f = frames[-1]
line = {}
for k, us in f:
    line.setdefault(k, []).append(us)
for k in names:
    if k in line:
        print("%-14s" % k, " ".join("%7.0f" % v for v in line[k]), "  sum %.2f ms" % (sum(line[k]) / 1e3))
PY
tail -1 $OUT/stats.txt
