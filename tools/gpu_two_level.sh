#!/bin/bash
# GPU box: two-level acceleration structure -- parity tests, then flat vs two-level on the instancing-heavy C5 stand-in and on C3.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-r02_two}
mkdir -p $OUT
cd $REPO
timeout 420 python -m pytest tests/test_two_level.py -m gpu -q 2>&1 | tail -40 > $OUT/tests.txt; tail -5 $OUT/tests.txt
B="--warmup 4 --no-profile --no-interactive --no-cpu-baseline"
timeout 240 python bench.py --workload c5 --steps 32 $B > $OUT/c5_flat.json 2> $OUT/c5_flat.err
timeout 240 python bench.py --workload c5 --steps 32 $B --accel two --refit 5 > $OUT/c5_two.json 2> $OUT/c5_two.err
timeout 120 python bench.py --steps 20 $B > $OUT/c3_flat.json 2> $OUT/c3_flat.err
timeout 120 python bench.py --steps 20 $B --accel two --refit 5 > $OUT/c3_two.json 2> $OUT/c3_two.err
python - <<PY
import json
for f in ("c5_flat", "c5_two", "c3_flat", "c3_two"):
    try:
        d = json.load(open("$OUT/" + f + ".json"))
        print(f, round(d["value"], 1), "Msamples/s  build ms", round(d["bvh_build_ms"], 1), d["accel"], "mean", d["image_mean"])
    except Exception as e:
        print(f, "FAILED", e, open("$OUT/" + f + ".err").read()[-1500:])
PY
