#!/bin/bash
# GPU box, round 3 first call: the full GPU suite on the phase-1 tree, the driver's bench line, the PMC passes (HBM / VALU / L2) on the timed
# pipeline, and one rank's shard of a 1 / 2 / 4 / 8-GPU C4 run (3840x2160) rendered alone.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03a
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/gputest.txt; cat $OUT/gputest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; tail -3 $OUT/bench_20.err
python - <<PY
import json
d = json.load(open("$OUT/bench_20.json"))
print("bench_20", round(d["value"], 1), "interactive", d.get("interactive", {}).get("value"), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("port_value"), d.get("cpu_baseline", {}).get("kind"))
print(json.dumps(d.get("roofline"), indent=1))
PY
bash tools/pmc_r03.sh r03a 2>&1 | tail -20
for N in 1 2 4 8; do
  timeout 200 python bench.py --workload c4 --emulate-shard 0/$N --steps 20 --warmup 5 --no-profile --no-interactive > $OUT/c4_shard_0of$N.json 2> $OUT/c4_shard_0of$N.err
  python -c "import json; d=json.load(open('$OUT/c4_shard_0of$N.json')); print('c4 shard 0/$N', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],3), 'ms/frame')"
done
