#!/bin/bash
# usage: tools/variants_shard_tune.sh R/N variant tune...
S=$1; v=$2; shift 2
if [ "$v" = default ]; then L=""; else L="$PWD/vk_raytrace_amd/variants/libptmi_$v.so"; fi
for t in "$@"; do
  out=$(PT_TUNE=$t PT_LIB=$L python bench.py --emulate-shard $S --steps ${STEPS:-256} --warmup 8 --no-profile --no-cpu-baseline 2>&1 | tail -1)
  echo "shard $S [$v] [$t] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), "ms/frame")')"
done
