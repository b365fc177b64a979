#!/bin/bash
# GPU box: acceleration-structure build times (arena for the flat build, concurrent BLAS builds) + the two-level GPU tests.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-build_times}
mkdir -p $OUT
cd $REPO
timeout 150 python -m pytest tests/test_two_level.py -m gpu -x -q 2>&1 | tail -4 > $OUT/tests.txt; tail -2 $OUT/tests.txt
B="--steps 8 --warmup 2 --no-profile --no-interactive --no-cpu-baseline"
run() {  # workload accel tune
  PT_TUNE="$3" timeout 100 python bench.py $B --workload $1 --accel $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-3s %-5s tune %-14s build %7.1f ms  %8.1f Msamples/s mean %.9f' % ('$1', '$2', '$3', d['bvh_build_ms'], d['value'], d['image_mean']))" | tee -a $OUT/build.txt
}
run c3 flat ""; run c5 flat ""; run c5 two "blasWorkers=1"; run c5 two "blasWorkers=4"
