#!/bin/bash
# GPU box: 4-wide structure with the stack spill area in global memory (no scratch array) vs the previous commit; the one-rank form of the
# N > 1 flow with the torch-free control plane; GPU suite.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03g
mkdir -p $OUT
cd $REPO
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.txt
for S in 20 96; do
  echo "== steps $S"
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh prev default w4 2>&1 | tee -a $OUT/variants_$S.txt
done
echo "== forced single-rank group + native gather"
PT_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-interactive > $OUT/force_dist.json 2> $OUT/force_dist.err; echo "rc $?"; python -c "import json; d=json.load(open('$OUT/force_dist.json')); print(d['value'], d['ranks_seen'], d['gather_ms'], d['image_mean'])"; grep -v "alt_rsmi\|^$" $OUT/force_dist.err | tail -5
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/gputest.txt; tail -8 $OUT/gputest.txt
