#!/bin/bash
# GPU box: every render kernel of the LAST timed window of a short bench run with its start, duration and queue -- what the end of a window looks like
# (which kernels of which launch sequence are still running while the others have finished).   usage: tools/window_timeline.sh <tag> [steps] [PT_TUNE]
TAG=$1; STEPS=${2:-20}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/window_$TAG
rm -rf $OUT; mkdir -p $OUT
PT_TUNE=$3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- python $REPO/bench.py --steps $STEPS --warmup 5 --repeats 3 --no-cpu-baseline --no-profile --no-interactive > $OUT/bench.json 2>/dev/null
find $OUT/raw -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/raw
python3 - <<PY | tee $OUT/window.txt
import csv, re
rows = list(csv.DictReader(open("$OUT/kernel_trace.csv")))
STAGE = re.compile(r"k_(generate|closest_k|closest_p|closest_x|shade|shadow_p|shadow_x|trace_p|trace_x|tail|accumulate)")
rows = [r for r in rows if STAGE.search(r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gens = [i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"]]
# the last window = the last group of k_generate launches that start within 2 ms of each other
last = gens[-1]; first = last
for i in reversed(gens):
    if int(rows[last]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) < 3_000_000: first = i
seq = rows[first:]
t0 = int(seq[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seq)
qs = sorted({r.get("Queue_Id", "?") for r in seq})
print(f"window {(t1 - t0) / 1e6:.2f} ms, {len(seq)} kernels, queues {qs}")
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  run {(e - s) / 1e3:8.1f}  q{qs.index(r.get('Queue_Id', '?'))}  {STAGE.search(r['Kernel_Name']).group(0)}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?'))}")
PY
