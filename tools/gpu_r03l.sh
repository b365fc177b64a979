#!/bin/bash
# GPU box: lane utilisation of the persistent trace kernels (PT_HIST build) and the leaf-step postponement variants (PT_LEAF_MIN)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03l
mkdir -p $OUT
cd $REPO
PT_LIB=$PWD/vk_raytrace_amd/variants/libptmi_hist.so timeout 200 python tools/gpu_hist.py 4 > $OUT/hist_base.txt 2>&1
PT_LIB=$PWD/vk_raytrace_amd/variants/libptmi_hist16.so timeout 200 python tools/gpu_hist.py 4 > $OUT/hist_leaf16.txt 2>&1
for S in 20 96; do
  echo "== steps $S"
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh default leaf8 leaf16 leaf24 leaf32 2>&1 | tee -a $OUT/variants_$S.txt
done
tail -4 $OUT/hist_base.txt; tail -3 $OUT/hist_leaf16.txt
