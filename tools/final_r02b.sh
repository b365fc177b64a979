#!/bin/bash
# GPU box: end-of-round evidence after k_tail and the two-level structure landed (gpurun_out/final_r02b/, copied to profiles/r02b_*).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_r02b
mkdir -p $OUT
cd $REPO
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err
timeout 300 python bench.py --steps 256 --warmup 8 --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err
B="--warmup 4 --no-profile --no-interactive --no-cpu-baseline"
timeout 200 python bench.py --workload c5 --steps 32 $B --accel two --refit 5 > $OUT/c5_two.json 2> $OUT/c5_two.err
PT_LIB=$REPO/vk_raytrace_amd/variants/libptmi_two3.so timeout 200 python bench.py --workload c5 --steps 32 $B --accel two > $OUT/c5_two_waves3.json 2> $OUT/c5_two_waves3.err
python - <<PY
import json
for f in ("bench_20", "bench_256", "c5_two", "c5_two_waves3"):
    try:
        d = json.load(open("$OUT/" + f + ".json"))
        print(f, round(d["value"], 1), "Msamples/s", "interactive", d.get("interactive", {}).get("value"), "build ms", round(d["bvh_build_ms"], 1), d.get("accel"), (d.get("roofline") or {}).get("stage"), (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20_under_rocprof.json 2> /dev/null
find $OUT/raw -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_bench20.csv \;
rm -rf $OUT/raw
head -12 $OUT/kernel_stats_bench20.csv | cut -c1-160
