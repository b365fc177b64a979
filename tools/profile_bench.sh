#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of a short bench run; copies the stats summary to gpurun_out/.
# usage: tools/profile_bench.sh <tag> [bench args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o trace -- python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
find $OUT/raw -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
ls -R $OUT/raw | head -20
head -20 $OUT/kernel_stats.csv
rm -rf $OUT/raw
