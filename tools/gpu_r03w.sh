#!/bin/bash
# GPU box: per-slot vertex copies for k_shade once more, this time with proof that they are in use (accel bytes) and with k_shade's HBM reads counted
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03w; mkdir -p $O
for t in "shadeTris=0" "shadeTris=1" "shadeTris=0" "shadeTris=1"; do
  echo -n "$t " | tee -a $O/variants.txt
  PT_TUNE=$t timeout 200 python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'accel bytes', d['accel']['bytes'])" | tee -a $O/variants.txt
done
cd /tmp && export TMPDIR=/tmp
for t in "shadeTris=0" "shadeTris=1"; do
  PT_TUNE=inflight=1,$t timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw_$t -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 0 --no-cpu-baseline --no-profile --no-interactive > /dev/null 2>&1
  f=$(find $O/raw_$t -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$t" <<'P' | tee -a $O/fetch.txt
import csv, sys, collections
tot = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("k_shade", "k_closest_p", "k_shadow_p"):
        if k + "<" in r["Kernel_Name"] or k + "(" in r["Kernel_Name"]:
            tot[k] += float(r["Counter_Value"])
n = 64 * 1920 * 1080
print(sys.argv[2], {k: round(v * 1024 * 2 / n, 1) for k, v in tot.items()}, "read bytes / sample")
P
  rm -rf $O/raw_$t
done
