#!/bin/bash
# GPU box: the PMC passes behind profiles/r02_traffic.json and profiles/r02_valu.json (read by bench.py).
# One rocprofv3 run per counter group (counters + kernel trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass), each under `timeout`:
# a pass that wedges costs minutes, not the box.  The bench runs ONE batch of 32 frames on one frame slot (inflight=1): rocprofv3 serialises
# kernels for counter collection anyway, and the per-kernel attribution is then exact.
# usage: tools/pmc_r02.sh [tag]      -> gpurun_out/pmc_<tag>/{traffic.json,valu.json}
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
FRAMES=32
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  PT_TUNE=inflight=1 timeout 420 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw$i -o p -- \
    python $REPO/bench.py --steps $FRAMES --warmup 0 --no-cpu-baseline --no-profile --no-interactive > $OUT/bench$i.json 2> $OUT/bench$i.err
  echo "pass $i ($CTRS): rc $?"
  find $OUT/raw$i -name '*counter_collection.csv' -exec cp {} $OUT/counters$i.csv \;
  rm -rf $OUT/raw$i
done
python3 $REPO/tools/pmc_r02_json.py $OUT $FRAMES
