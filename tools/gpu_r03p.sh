#!/bin/bash
# GPU box: BLAS builds of the C5 stand-in with 4 / 8 / 12 / 16 host workers (a stream and an arena each); parity subset for the all-opaque shadow early-out
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03p; mkdir -p $O
for w in 4 8 12 16; do
  echo "== blasWorkers=$w" | tee -a $O/blas_workers.txt
  PT_TUNE=accel=two,blasWorkers=$w timeout 300 python tools/build_only.py c5 3 2>&1 | tail -3 | tee -a $O/blas_workers.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_two_level.py tests/test_golden.py -m gpu -q -x -k "c1_quad or use_any_hit or c2_full or golden or tiny_and_ragged or mode_switch or punctual" 2>&1 | tail -4 | tee $O/gputest_subset.txt
