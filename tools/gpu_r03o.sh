#!/bin/bash
# GPU box: per-slot vertex-attribute copy for k_shade (DeviceScene::shadeTris) + no state writes for paths that end in k_shade, against the previous commit
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_two_level.py -m gpu -q -x -k "c1_quad or path_traced or sponza_like or c3_full or multiple_samples or heatmap or feature_box or update_instances or golden or c2_full" 2>&1 | tail -5 | tee $O/gputest_subset.txt
for S in 20 96; do
  echo "== steps $S" | tee -a $O/variants.txt
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh prev default 2>&1 | tee -a $O/variants.txt
  echo "default with PT_TUNE=shadeTris=0:" | tee -a $O/variants.txt
  PT_TUNE=shadeTris=0 STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh default 2>&1 | tee -a $O/variants.txt
done
