#!/bin/bash
# GPU box: two-level structure with compact nodes (BLASes + TLAS) and the block table of the world-triangle -> instance lookup: parity suite, C5 rate
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03z; mkdir -p $O
timeout 400 python -m pytest tests/test_two_level.py -m gpu -q -x 2>&1 | tail -3 | tee $O/test_two_level.txt
for t in "cnodes=0" ""; do
  echo -n "c5 two-level PT_TUNE=$t " | tee -a $O/c5_two.txt
  PT_TUNE=$t timeout 300 python bench.py --workload c5 --accel two --steps 32 --warmup 8 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['accel'], 'build', round(d['bvh_build_ms'],1))" | tee -a $O/c5_two.txt
done
