#!/bin/bash
# GPU box: launch-policy knobs of the persistent kernels once more on the final binary (compact nodes shifted the balance)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03y; mkdir -p $O
for t in "" "refill=32" "refill=40" "refill=56" "waves=4096" "waves=6144" "waves=7168" ""; do
  echo -n "PT_TUNE=$t " | tee -a $O/sweep96.txt
  PT_TUNE=$t timeout 200 python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1))" | tee -a $O/sweep96.txt
done
