#!/bin/bash
# GPU box: first run of the 8-wide structure: smoke + the parity tests that matter most (-x), then the variants at 20 and 96 steps.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03c
mkdir -p $OUT
cd $REPO
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee $OUT/smoke.txt
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $OUT/gputest.txt; tail -25 $OUT/gputest.txt
for S in 20 96; do
  echo "== steps $S"
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh old4wide default w4 w6 w5s8 2>&1 | tee -a $OUT/variants_$S.txt
done
