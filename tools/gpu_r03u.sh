#!/bin/bash
# GPU box: compact nodes as the default; any-hit records fetched only for candidates (PT_LAZY_ALPHA) against fetching them with the triangle
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03u; mkdir -p $O
for S in 20 96; do
  echo "== steps $S" | tee -a $O/variants.txt
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh default lazy default lazy 2>&1 | tee -a $O/variants.txt
done
PT_LIB=$PWD/vk_raytrace_amd/variants/libptmi_lazy.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "path_traced or sponza_like or c3_full or fuzz" 2>&1 | tail -3 | tee -a $O/gputest_lazy.txt
