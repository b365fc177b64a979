#!/bin/bash
# GPU box: bench.py under a list of PT_TUNE settings ("-" = defaults)
STEPS=${STEPS:-96}
for t in "$@"; do
  if [ "$t" = "-" ]; then T=""; else T="$t"; fi
  out=$(PT_TUNE=$T python bench.py --steps $STEPS --warmup 8 --no-cpu-baseline --no-profile 2>&1 | tail -1)
  echo "[$t] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "Msamples/s", round(d["ms_per_step"],3), "ms")' 2>/dev/null || echo "$out")"
done
