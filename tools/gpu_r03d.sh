#!/bin/bash
# GPU box: the 8-wide fp16-plane structure with nearest-first order: smoke, full GPU suite (comm test apart, with RCCL diagnostics), variants.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03d
mkdir -p $OUT
cd $REPO
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
for S in 20 96; do
  echo "== steps $S"
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh old4wide default w4 2>&1 | tee -a $OUT/variants_$S.txt
done

for E in; do
  echo "== comm test with [$E]"
  env $E timeout 200 python -m pytest tests/test_comm.py -m gpu -q -x -s -k single_process 2>&1 | grep -v "^$" | tail -25 | tee -a $OUT/comm.txt
done
