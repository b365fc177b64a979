#!/bin/bash
# GPU box: pipelined display loop -- parity test and the interactive numbers of bench.py
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03m; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipelined_display or interactive_state or tonemap_matches" 2>&1 | tail -8 | tee $O/test.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['interactive'])"
