#!/bin/bash
# GPU box: cheaper ordering of a node's hit children (nearest entered, the others pushed in slot order / second nearest on top) against the selection loop
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03s; mkdir -p $O
for S in 20 96; do
  echo "== steps $S" | tee -a $O/variants.txt
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh default push1 push2 2>&1 | tee -a $O/variants.txt
done
for v in push1 push2; do
  PT_LIB=$PWD/vk_raytrace_amd/variants/libptmi_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c1_quad or path_traced or sponza_like or c3_full or fuzz" 2>&1 | tail -3 | tee -a $O/gputest_$v.txt
done
