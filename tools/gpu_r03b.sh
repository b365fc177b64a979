#!/bin/bash
# GPU box: the comm test with RCCL's own diagnostics, then the whole GPU suite (no -x).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03b
mkdir -p $OUT
cd $REPO
NCCL_DEBUG=INFO timeout 300 python -m pytest tests/test_comm.py -m gpu -q -x 2>&1 | tail -80 > $OUT/comm.txt; tail -40 $OUT/comm.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/gputest.txt; tail -25 $OUT/gputest.txt
