#!/bin/bash
# GPU box: kernel timeline of the driver's short run on one rank's shard of an 8-GPU job (who runs when, on which queue, how long).
# usage: tools/gpu_shard_timeline.sh <tag> <shard R/N> <PT_TUNE>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-shard_timeline}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PT_TUNE="$3" timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-interactive --emulate-shard ${2:-0/8} > $OUT/bench.json 2> $OUT/bench.err
find $OUT/raw -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/raw
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_trace.csv")))
print("columns:", list(rows[0].keys()))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
render = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_generate", "k_closest", "k_shade", "k_shadow", "k_accumulate", "k_tail"))]
gens = [i for i, r in enumerate(render) if "k_generate" in r["Kernel_Name"]]
# the timed region = the last 4 launch sequences (20 frames as 4 pieces); warm-up = the ones before
first = gens[-4] if len(gens) >= 4 else gens[0]
t0 = int(render[first]["Start_Timestamp"])
qkey = "Queue_Id" if "Queue_Id" in render[0] else ("Stream_Id" if "Stream_Id" in render[0] else None)
end = 0
for r in render[first:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    end = max(end, e)
    print("%8.1f us  +%7.1f us  q=%s  %s  grid=%s" % (s / 1e3, (e - s) / 1e3, r.get(qkey, "?") if qkey else "?", short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
print("timed region %.3f ms, %d kernels" % (end / 1e6, len(render) - first))
PY
