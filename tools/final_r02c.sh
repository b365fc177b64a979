#!/bin/bash
# GPU box: last evidence run of round 2 (after the step-wise launcher): full GPU suite, the driver's bench line, rocprofv3 kernel stats of it.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_r02c
mkdir -p $OUT
cd $REPO
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20_under_rocprof.json 2> /dev/null
find $OUT/raw -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_bench20.csv \;
rm -rf $OUT/raw
cd $REPO
python - <<PY
import json
d = json.load(open("$OUT/bench_20.json"))
print("bench_20", round(d["value"], 1), "Msamples/s interactive", d.get("interactive", {}).get("value"), "roofline", d["roofline"]["stage"], d["roofline"]["frac"], "hbm", d["hbm_measured"]["frac"])
PY
timeout 420 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/gputest.txt; cat $OUT/gputest.txt
