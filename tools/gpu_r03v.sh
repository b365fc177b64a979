#!/bin/bash
# GPU box: compact nodes at a 128-byte stride (never straddling a cache line) against the 80-byte stride
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03v; mkdir -p $O
for S in 20 96; do
  echo "== steps $S" | tee -a $O/variants.txt
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh default cnpad default cnpad 2>&1 | tee -a $O/variants.txt
done
