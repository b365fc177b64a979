#!/bin/bash
# usage: tools/scaling_tune.sh R/N tune1 tune2 ...
S=$1; shift
for t in "$@"; do
  out=$(PT_TUNE=$t python bench.py --emulate-shard $S --steps ${STEPS:-256} --warmup 8 --no-profile --no-cpu-baseline --no-interactive 2>&1 | tail -1)
  echo "shard $S [$t] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), "ms/frame")')"
done
