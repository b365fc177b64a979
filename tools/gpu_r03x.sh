#!/bin/bash
# GPU box: per-slot vertex copies as the default -- 20-step check against shadeTris=0, parity subset and launch-policy invariance
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03x; mkdir -p $O
for t in "shadeTris=0" "" "shadeTris=0" ""; do
  echo -n "PT_TUNE=$t " | tee -a $O/variants20.txt
  PT_TUNE=$t timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1))" | tee -a $O/variants20.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_two_level.py tests/test_golden.py -m gpu -q -x -k "launch_policy or c1_quad or path_traced or sponza_like or c3_full or c2_full or golden or feature_box or update_instances or gltf_file or fuzz" 2>&1 | tail -4 | tee $O/gputest.txt
