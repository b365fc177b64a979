#!/bin/bash
# GPU box: k_tail -- parity subset, then the threshold sweep on the driver's short run (1-GPU and one rank of 8) and on a long run.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-tail_sweep}
mkdir -p $OUT
cd $REPO
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or path_traced or launch_policy_never or tiny or sponza_like_reduced or multiple_samples or rtx_pipeline" 2>&1 | tail -15 > $OUT/tests.txt; tail -4 $OUT/tests.txt
B="--warmup 5 --no-profile --no-cpu-baseline"
run() {  # shard steps tune extra
  PT_TUNE="$3" timeout 120 python bench.py $B --steps $2 --emulate-shard $1 $4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('shard %-4s steps %-4s tune %-16s %8.4f ms/frame  %8.1f Msamples/s  interactive %s' % ('$1', '$2', '$3', d['ms_per_step'], d['value'], d.get('interactive', {}).get('value')))" | tee -a $OUT/sweep.txt
}
for t in tail=0 tail=65536 tail=262144 tail=1048576; do run 0/8 20 $t --no-interactive; done
for t in tail=0 tail=131072 tail=1048576; do run 0/1 20 $t; done
for t in tail=0 tail=262144; do run 0/1 256 $t --no-interactive; done
