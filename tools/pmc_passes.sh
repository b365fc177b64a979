#!/bin/bash
# GPU box: the PMC passes behind profiles/rNN_traffic.json, rNN_valu.json and rNN_cache.json (read by bench.py).
# One rocprofv3 run per counter group (counters + kernel trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass; never together with
# --stats / --sys-trace), each under `timeout`: a pass that wedges costs minutes, not the box.  The bench runs the TIMED pipeline (default
# launch policy: k_tail takes the late bounces) as batches of 32 frames on one frame slot (inflight=1): rocprofv3 serialises kernels for
# counter collection anyway, and the per-kernel attribution is then exact.  96 frames = three batches, so that the queue-size feedback
# has settled for two of them.
# usage: tools/pmc_passes.sh [tag]      -> gpurun_out/pmc_<tag>/{traffic.json,valu.json,cache.json}
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
FRAMES=${PMC_FRAMES:-96}
WL=${PMC_WORKLOAD:-c3}     # BASELINE configuration (bench.py --workload); the summaries of c2 / c4 / c5 are committed as profiles/rNN_<kind>_<workload>.json
ACCEL=${PMC_ACCEL:-flat}
rocprofv3 -L > $OUT/counters_available.txt 2>&1
i=0
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU")
[ "${PMC_SET:-full}" = "min" ] && GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU")
for CTRS in "${GROUPS_[@]}"; do
  i=$((i+1))
  PT_TUNE=inflight=1,warm=0 timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw$i -o p -- \
    python $REPO/bench.py --workload $WL --accel $ACCEL --steps $FRAMES --warmup 0 --repeats 1 --no-cpu-baseline --no-profile --no-interactive > $OUT/bench$i.json 2> $OUT/bench$i.err
  echo "pass $i ($CTRS): rc $?"
  # keep the rows of the render kernels only (a two-level build dispatches thousands of builder kernels: the merged output must stay small)
  find $OUT/raw$i -name '*counter_collection.csv' -exec sh -c 'head -1 "$1" > "$2"; grep -E "k_(generate|closest|shade|shadow|trace|tail|accumulate)" "$1" >> "$2"' _ {} $OUT/counters$i.csv \;
  rm -rf $OUT/raw$i
done
python3 $REPO/tools/pmc_passes_json.py $OUT $FRAMES
