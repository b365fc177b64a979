#!/bin/bash
# GPU box (one GPU): renders each rank's tile shard of an N-GPU run on its own and prints ms/frame; the N-GPU frame time is the
# max over ranks (no communication while rendering), so predicted efficiency = t(1) / (N * max_r t(r/N)).
for N in "$@"; do
  for ((R=0; R<N; R++)); do
    out=$(python bench.py --emulate-shard $R/$N --steps ${STEPS:-256} --warmup 8 --no-profile --no-cpu-baseline --no-interactive 2>&1 | tail -1)
    echo "N=$N rank=$R $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), "ms/frame")')"
  done
done
