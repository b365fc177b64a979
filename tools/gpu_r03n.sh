#!/bin/bash
# GPU box: two-level structure with the merged world-space structure for the prim-meshes instantiated once -- parity suites, then C3 / C5 throughput
# flat vs two-level (merged, default) vs two-level with a BLAS per prim-mesh (mergeSingles=0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03n; mkdir -p $O
#timeout 900 python -m pytest tests/test_two_level.py -m gpu -q -x 2>&1 | tail -6 | tee $O/test_two_level.txt
#timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "launch_policy or c4_c5 or c3_full or smoke" 2>&1 | tail -6 | tee $O/test_policy.txt
for cfg in c3 c5; do
  for mode in "flat|" "two|" "two|mergeSingles=0"; do
    a=${mode%%|*}; t=${mode##*|}
    echo "== $cfg accel=$a PT_TUNE=$t" | tee -a $O/two_level_bench.txt
    PT_TUNE=$t timeout 300 python bench.py --workload $cfg --accel $a --steps 32 --warmup 8 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'Msamples/s', d.get('accel'), 'build ms', d.get('bvh_build_ms'))" | tee -a $O/two_level_bench.txt
  done
done
