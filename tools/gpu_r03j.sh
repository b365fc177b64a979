#!/bin/bash
# GPU box: XCD-aware region mapping of the bounce-0 launches vs the previous commit; k_shade at 5 waves; packet kernel at 6 waves; parity subset.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03j
mkdir -p $OUT
cd $REPO
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.txt
for S in 20 96; do
  echo "== steps $S"
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh prev default sw5 pk6 2>&1 | tee -a $OUT/variants_$S.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c1_quad or path_traced or sponza_like or c3_full or multiple_samples or launch_policy_never or shard or heatmap" 2>&1 | tail -6 | tee $OUT/gputest_subset.txt
