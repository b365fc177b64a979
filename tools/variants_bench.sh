#!/bin/bash
# GPU box: bench every libptmi variant given (names under vk_raytrace_amd/variants/, or "default")
STEPS=${STEPS:-96}
for v in "$@"; do
  if [ "$v" = default ]; then L=""; else L="$PWD/vk_raytrace_amd/variants/libptmi_$v.so"; fi
  out=$(PT_LIB=$L python bench.py --steps $STEPS --warmup 8 --no-cpu-baseline ${BENCH_FLAGS:---no-profile} 2>&1 | tail -1)
  echo "$v: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Msamples/s", d["ms_per_step"], "ms", (d.get("roofline") or {}))' 2>/dev/null || echo "$out")"
done
