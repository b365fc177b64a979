#!/bin/bash
# GPU box: interleaved submission of the pieces of a cut batch -- ordering tests, then A/B on the short run.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-interleave}
mkdir -p $OUT
cd $REPO
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "launch_policy or camera_change or checkpoint or sample_example or shard_gather" 2>&1 | tail -15 > $OUT/tests.txt; tail -4 $OUT/tests.txt
B="--warmup 5 --no-profile --no-cpu-baseline --no-interactive"
run() {  # shard steps tune
  PT_TUNE="$3" timeout 120 python bench.py $B --steps $2 --emulate-shard $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('shard %-4s steps %-4s tune %-16s %8.4f ms/frame  %8.1f Msamples/s' % ('$1', '$2', '$3', d['ms_per_step'], d['value']))" | tee -a $OUT/sweep.txt
}
for t in interleave=0 interleave=1; do run 0/8 20 $t; run 0/1 20 $t; done
run 0/4 20 interleave=1; run 0/2 20 interleave=1
run 0/1 256 interleave=1
