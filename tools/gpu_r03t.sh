#!/bin/bash
# GPU box: 80-byte compact nodes for the persistent trace kernels (PT_TUNE cnodes=1: five 16-byte requests per node instead of seven)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03t; mkdir -p $O
for S in 20 96; do
  echo "== steps $S" | tee -a $O/variants.txt
  for t in "cnodes=0" "cnodes=1" "cnodes=0" "cnodes=1"; do
    echo -n "$t " | tee -a $O/variants.txt
    PT_TUNE=$t STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh default 2>&1 | tee -a $O/variants.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "launch_policy" 2>&1 | tail -4 | tee $O/gputest_policy.txt
