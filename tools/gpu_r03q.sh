#!/bin/bash
# GPU box: postponed leaves in the persistent trace kernels (PT_TUNE leafMin, pt_machine.h lane_round) against the previous commit; launch-policy parity
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
for S in 20 96; do
  echo "== steps $S" | tee -a $O/variants.txt
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh prev 2>&1 | tee -a $O/variants.txt
  for lm in -1 0 8 16 24 32 48; do
    echo -n "leafMin=$lm " | tee -a $O/variants.txt
    PT_TUNE=leafMin=$lm STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh default 2>&1 | tee -a $O/variants.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "launch_policy" 2>&1 | tail -4 | tee $O/gputest_policy.txt
