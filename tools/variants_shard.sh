#!/bin/bash
# usage: tools/variants_shard.sh R/N variant...   (ms/frame of one emulated shard per libptmi variant)
S=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then L=""; else L="$PWD/vk_raytrace_amd/variants/libptmi_$v.so"; fi
  out=$(PT_LIB=$L python bench.py --emulate-shard $S --steps ${STEPS:-256} --warmup 8 --no-profile --no-cpu-baseline 2>&1 | tail -1)
  echo "shard $S [$v] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), "ms/frame")')"
done
