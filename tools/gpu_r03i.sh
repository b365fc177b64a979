#!/bin/bash
# GPU box: launch-policy sweep on the sw4 build (k_shade at 4 waves / SIMD): persistent waves, frames in flight, chunk, refill, tail threshold.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03i
mkdir -p $OUT
cd $REPO
run() { echo "== [$2] steps $1 PT_TUNE=$3"; PT_TUNE=$3 STEPS=$1 BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh $2 2>&1 | tee -a $OUT/sweep.txt; }
for T in "waves=4096" "waves=5120" "waves=6144" "waves=8192" "waves=5120,refill=32" "waves=5120,inflight=3" "waves=5120,inflight=6" "waves=5120,inflight=8" "waves=5120,chunk=128" "waves=5120,packetWaves=4096" "waves=5120,packetWaves=16384" "waves=5120,tail=32768" "waves=5120,tail=262144" "waves=5120,batch=16" "waves=10240,inflight=2"; do
  run 96 sw4 "$T"
done
for T in "waves=4096" "waves=5120" "waves=8192" "waves=5120,inflight=8" "waves=5120,tail=32768" "waves=5120,tail=262144"; do
  run 20 sw4 "$T"
done
run 96 default "waves=5120"
run 20 default "waves=5120"
