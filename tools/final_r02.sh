#!/bin/bash
# GPU box: the end-of-round evidence of round 2, everything under gpurun_out/final_r02/ (copied to profiles/ afterwards).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_r02
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/gputest.txt; cat $OUT/gputest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err
timeout 300 python bench.py --steps 256 --warmup 8 --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err
timeout 300 python bench.py --steps 32 --warmup 5 --no-cpu-baseline > $OUT/bench_32.json 2> $OUT/bench_32.err
python - <<PY
import json
for f in ("bench_20", "bench_256", "bench_32"):
    d = json.load(open("$OUT/" + f + ".json"))
    print(f, round(d["value"], 1), "Msamples/s", "roofline", d["roofline"]["stage"], round(d["roofline"]["frac"], 3), "interactive", d.get("interactive", {}).get("value"), "build ms", round(d["bvh_build_ms"], 1))
PY
# rocprofv3 kernel stats of the bench command itself (the driver's command line) and the per-dispatch trace of a serialised 32-frame batch
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20_under_rocprof.json 2> /dev/null
find $OUT/raw -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_bench20.csv \;
rm -rf $OUT/raw
cd $REPO
timeout 300 bash tools/trace_batch.sh final > $OUT/trace_batch.txt 2>&1; cat $OUT/trace_batch.txt | tail -12
timeout 600 bash tools/pmc_r02.sh final > $OUT/pmc.txt 2>&1; tail -12 $OUT/pmc.txt
cp $REPO/gpurun_out/pmc_final/traffic.json $OUT/traffic.json; cp $REPO/gpurun_out/pmc_final/valu.json $OUT/valu.json
timeout 60 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/valu_peak.hip -o /tmp/valu_peak 2>/dev/null && timeout 60 /tmp/valu_peak > $OUT/valu_peak.txt
STEPS=128 timeout 900 bash tools/scaling_estimate.sh 1 2 4 8 > $OUT/scaling_estimate.txt 2>&1; cat $OUT/scaling_estimate.txt
