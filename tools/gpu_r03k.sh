#!/bin/bash
# memory-pipeline PMC passes + a few 20-step policy checks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
PMC_FRAMES=64 timeout 900 bash tools/pmc_r03_mem.sh r03mem > $O/pmc_mem.txt 2>&1
for t in "" "inflight=2" "inflight=3" "batch=10" "batch=20,inflight=1"; do
  echo "== PT_TUNE=$t" >> $O/policy20.txt
  PT_TUNE=$t timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['batch'])" >> $O/policy20.txt 2>&1
done
cat $O/pmc_mem.txt | tail -15; cat $O/policy20.txt
