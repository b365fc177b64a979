#!/bin/bash
# GPU box: the packet kernel (k_closest_k) under two library variants -- duration from a kernel trace, instruction counts from one PMC pass; one frame slot
# so that kernels do not overlap.   usage: tools/packet_ab.sh <tag> <libA> <libB> ...   ("default" = the product)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$REPO/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  L=""; [ "$v" != default ] && L="$REPO/vk_raytrace_amd/variants/libptmi_$v.so"
  for wl in "c3:--steps 32" "c5:--workload c5 --accel two --steps 32"; do
    n=${wl%%:*}; a=${wl#*:}
    PT_LIB=$L PT_TUNE=inflight=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o p -- python $REPO/bench.py $a --warmup 0 --repeats 1 --no-cpu-baseline --no-profile --no-interactive > /dev/null 2> $O/err_${v}_$n.txt
    find $O/raw -name '*kernel_stats.csv' -exec cp {} $O/stats_${v}_$n.csv \; ; rm -rf $O/raw
    echo "$v $n: $(grep -E 'k_closest_k' $O/stats_${v}_$n.csv | cut -d, -f1-5 | cut -c1-30,100-200 | tr '\n' ' ')" | tee -a $O/log.txt
  done
  PT_LIB=$L PT_TUNE=inflight=1,warm=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/raw -o p -- python $REPO/bench.py --steps 32 --warmup 0 --repeats 1 --no-cpu-baseline --no-profile --no-interactive > /dev/null 2>> $O/err_${v}_c3.txt
  find $O/raw -name '*counter_collection.csv' -exec cp {} $O/ctr_${v}.csv \; ; rm -rf $O/raw
  python3 - $O/ctr_${v}.csv $v <<'P' | tee -a $O/log.txt
import csv, sys, collections
agg = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_closest_k" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
w = agg.get("SQ_WAVES", 1) or 1
print(sys.argv[2], "k_closest_k per wave:", {k: round(v / w, 1) for k, v in agg.items()})
P
done
