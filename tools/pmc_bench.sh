#!/bin/bash
# GPU box: rocprofv3 PMC passes (counters only, with --kernel-trace) over a short bench run.
# usage: tools/pmc_bench.sh <tag> "<counters pass 1>" "<counters pass 2>" ...   (bench args via BENCH_ARGS)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for CTRS in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw$i -o p -- python $REPO/bench.py --no-cpu-baseline --no-profile ${BENCH_ARGS:---steps 4 --warmup 1 --tex-size 256} > $OUT/bench$i.json 2> $OUT/bench$i.err
  find $OUT/raw$i -name '*counter_collection.csv' -exec cp {} $OUT/counters$i.csv \;
  rm -rf $OUT/raw$i
done
python3 - <<PY
import csv, glob, collections, os
out = "$OUT"
for f in sorted(glob.glob(out + "/counters*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        if k.startswith(("k_closest", "k_shadow", "k_shade", "k_generate")) or "k_closest" in k or "k_shadow" in k or "k_shade" in k:
            print(os.path.basename(f), k[:40], {c: "%.4g" % v for c, v in d.items()})
PY
