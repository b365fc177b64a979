#!/bin/bash
# GPU box (branch cnodes64-experiment): the first measurement of the 64-byte nodes -- launch-policy parity (bit-identical images with cnodes=2), then
# the bench at 20 / 96 steps, default (80-byte nodes) against PT_TUNE=cnodes=2, two alternating pairs
cd $GRAFT_REPO_ROOT; O=gpurun_out/cnodes64; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "launch_policy" > $O/parity.txt 2>&1; grep -E "passed|failed" $O/parity.txt | tail -1
for S in 20 96; do
  echo "== steps $S" | tee -a $O/variants.txt
  for t in "" "cnodes=2" "" "cnodes=2"; do
    echo -n "PT_TUNE=$t " | tee -a $O/variants.txt
    PT_TUNE=$t timeout 200 python bench.py --steps $S --warmup 8 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1))" | tee -a $O/variants.txt
  done
done
