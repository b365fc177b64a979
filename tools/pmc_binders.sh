#!/bin/bash
# GPU box: the "what binds" PMC passes (round 6) -> gpurun_out/pmc_<tag>/binders.json (copied to profiles/rNN_binders.json by hand).
# Hardware lane occupancy of the vector ALU (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), instruction-cache requests / misses, instruction-fetch
# level, VMEM issue cycles and the VALU instruction classes, per kernel.  Same discipline as tools/pmc_passes.sh: one rocprofv3 run per counter
# group (counters + kernel trace only), each under `timeout`, one frame slot so that the per-kernel attribution is exact.  Counter names that this
# rocprofv3 does not list are dropped from their group (the list of what was asked for and what exists is kept in the output).
# usage: tools/pmc_binders.sh [tag]
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
FRAMES=${PMC_FRAMES:-64}
WL=${PMC_WORKLOAD:-c3}
rocprofv3 -L > $OUT/counters_available.txt 2>&1
have() { grep -qw "$1" $OUT/counters_available.txt; }
GROUPS_=(
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL"
 "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
 "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F16"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVE_CYCLES"
 "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ"
)
[ "${PMC_SET:-full}" = "min" ] && GROUPS_=("SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES")
i=20
for G in "${GROUPS_[@]}"; do
  i=$((i+1)); CTRS=""; n=0
  for c in $G; do if have $c && [ $n -lt 8 ]; then CTRS="$CTRS $c"; n=$((n+1)); else echo "not collected: $c" >> $OUT/binders_missing.txt; fi; done
  [ -z "$CTRS" ] && continue
  PT_TUNE=${PMC_TUNE:-inflight=1,warm=0} timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw$i -o p -- \
    python $REPO/bench.py --workload $WL --accel ${PMC_ACCEL:-flat} --steps $FRAMES --warmup 0 --repeats 1 --no-cpu-baseline --no-profile --no-interactive > $OUT/bench$i.json 2> $OUT/bench$i.err
  echo "pass $i ($CTRS): rc $?"
  # keep the rows of the render kernels only (a two-level build dispatches thousands of builder kernels: the merged output must stay small)
  find $OUT/raw$i -name '*counter_collection.csv' -exec sh -c 'head -1 "$1" > "$2"; grep -E "k_(generate|closest|shade|shadow|trace|tail|accumulate)" "$1" >> "$2"' _ {} $OUT/counters$i.csv \;
  rm -rf $OUT/raw$i
done
python3 $REPO/tools/pmc_binders_json.py $OUT $FRAMES
