#!/bin/bash
# GPU box: measurement variants on the 4-wide structure: robust T2 (fp32 + error bound, fp64 when ambiguous), k_shade at 4 waves / SIMD,
# persistent-wave counts.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03h
mkdir -p $OUT
cd $REPO
for S in 20 96; do
  echo "== steps $S"
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh default robust sw4 2>&1 | tee -a $OUT/variants_$S.txt
done
for T in "waves=1024" "waves=3072" "waves=4096" "waves=3072,packetWaves=4096" "refill=32" "refill=56"; do
  echo "== PT_TUNE=$T"
  PT_TUNE=$T STEPS=96 BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh default 2>&1 | tee -a $OUT/tune_96.txt
done
