#!/bin/bash
# GPU box: the end-of-round evidence set of round 3 in one call -> gpurun_out/final_r03/ (copied to profiles/r03_final_* and profiles/r03_*.json).
# Order matters: the PMC passes run first and their JSON summaries are put where bench.py reads them (profiles/r03_{traffic,valu,cache}.json),
# so that the bench lines written afterwards price their roofline fields from THIS binary's counters.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/final_r03; mkdir -p $O
cd $REPO
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.txt
PMC_FRAMES=96 timeout 1200 bash tools/pmc_r03.sh r03final > $O/pmc.txt 2>&1
for k in traffic valu cache; do
  [ -s gpurun_out/pmc_r03final/$k.json ] && cp gpurun_out/pmc_r03final/$k.json profiles/r03_$k.json && cp gpurun_out/pmc_r03final/$k.json $O/r03_$k.json
done
tail -4 $O/pmc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
timeout 600 python bench.py --steps 256 --warmup 8 --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof20 -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-interactive > $O/bench_20_profiled.json 2> $O/prof20.err)
find $O/prof20 -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_bench20.csv \;
rm -rf $O/prof20
for wl in c3 c4; do
  for n in 1 2 4 8; do
    for steps in 20 128; do
      [ $wl = c4 ] && [ $steps = 128 ] && continue
      timeout 300 python bench.py --workload $wl --emulate-shard 0/$n --steps $steps --warmup 5 --no-cpu-baseline --no-profile --no-interactive > $O/shard_${wl}_0of${n}_${steps}.json 2>/dev/null
    done
  done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/final_r03"
for f in sorted(glob.glob(O + "/bench_*.json")) + sorted(glob.glob(O + "/shard_*.json")):
    try:
        d = json.loads(open(f).readline())
        print(os.path.basename(f), round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4), (d.get("interactive") or {}).get("pipelined_value"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic_frac"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
P
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/gputest.txt
