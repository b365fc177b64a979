#!/bin/bash
# GPU box: the driver's short run (--steps 20) on one rank's shard of an N-GPU job, under launch-policy variants (PT_TUNE).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-shard_sweep}
mkdir -p $OUT
cd $REPO
B="--steps 20 --warmup 5 --no-profile --no-interactive --no-cpu-baseline"
run() {  # shard tune
  PT_TUNE="$2" timeout 120 python bench.py $B --emulate-shard $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('shard %-4s tune %-28s %8.4f ms/frame  %8.1f Msamples/s' % ('$1', '$2', d['ms_per_step'], d['value']))" | tee -a $OUT/sweep.txt
}
for s in 0/1 0/2 0/4 0/8 3/8; do run $s ""; done
for t in "inflight=1" "inflight=2" "inflight=8" "waves=1024" "waves=4096" "packetWaves=2048" "inflight=8,waves=1024" "refill=32"; do run 0/8 "$t"; done
for t in "inflight=6" "inflight=8" "waves=4096"; do run 0/1 "$t"; done
