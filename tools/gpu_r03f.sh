#!/bin/bash
# GPU box: after the return to the 4-wide structure: full GPU suite, the driver's bench line (CPU baselines with the thread probe), and the
# one-rank form of the N > 1 flow (gloo control plane + native RCCL gather in the same process).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03f
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/gputest.txt; tail -15 $OUT/gputest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; tail -3 $OUT/bench_20.err
python - <<PY
import json
d = json.load(open("$OUT/bench_20.json"))
print("bench_20", round(d["value"], 1), "interactive", round(d.get("interactive", {}).get("value", 0), 1), "cpu", d.get("cpu_baseline"))
PY
echo "== forced single-rank gloo control plane + native gather"
PT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-interactive > $OUT/force_dist.json 2> $OUT/force_dist.err; echo "rc $?"; python -c "import json; d=json.load(open('$OUT/force_dist.json')); print(d['value'], d['ranks_seen'], d['gather_ms'], d['image_mean'])"; grep -v "alt_rsmi\|^$" $OUT/force_dist.err | tail -5
