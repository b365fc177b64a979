#!/bin/bash
# GPU box: the kernels of ONE serialised launch sequence (PT_TUNE inflight=1, warm=0) in order, with duration and the idle GAP before each
# (rocprofv3 kernel trace): what the dependent chain of ~25 launches costs beyond the kernels themselves.
# usage: tools/trace_gaps.sh <tag> [steps]
TAG=$1; STEPS=${2:-20}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/gaps_$TAG
rm -rf $OUT; mkdir -p $OUT
PT_TUNE=inflight=1,warm=0,$PT_TUNE timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- python $REPO/bench.py --steps $STEPS --warmup 0 --repeats 2 --no-cpu-baseline --no-profile --no-interactive > $OUT/bench.json 2>/dev/null
find $OUT/raw -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/raw
python3 - <<PY | tee $OUT/gaps.txt
import csv
rows = list(csv.DictReader(open("$OUT/kernel_trace.csv")))
import re
STAGE = re.compile(r"k_(generate|closest_k|closest_p|closest_x|shade|shadow_p|shadow_x|tail|accumulate)")
rows = [r for r in rows if STAGE.search(r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last sequence: from the last k_generate on
last = max(i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"])
seq = rows[last:]
t0 = int(seq[0]["Start_Timestamp"]); prev_end = t0
busy = gaps = 0.0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = STAGE.search(r["Kernel_Name"]).group(0)
    gap = (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  gap {gap:7.1f}  run {(e - s) / 1e3:8.1f}  {name}")
    busy += (e - s) / 1e3; gaps += max(0.0, gap); prev_end = max(prev_end, e)
print(f"sequence: {(prev_end - t0) / 1e3:.1f} us, kernels {busy:.1f} us, gaps {gaps:.1f} us ({100 * gaps / max(1.0, (prev_end - t0) / 1e3):.1f} %), {len(seq)} launches")
PY
