#!/bin/bash
# GPU box: vector-memory pipeline counters of the timed pipeline (TA busy / stalls, L1 (TCP) wave latency, L1->L2 read latency, address translation),
# the evidence behind "what bounds the trace kernels" in DESIGN.md section 5.  Same bench invocation as tools/pmc_passes.sh, one rocprofv3 run per
# counter group (at most four counters of a block per pass), each under `timeout`.
# usage: tools/pmc_r03_mem.sh [tag]      -> gpurun_out/pmc_<tag>/mempipe.json
TAG=${1:-r03mem}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
FRAMES=${PMC_FRAMES:-96}
i=0
for CTRS in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PT_TUNE=inflight=1,warm=0 timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw$i -o p -- \
    python $REPO/bench.py --steps $FRAMES --warmup 0 --repeats 1 --no-cpu-baseline --no-profile --no-interactive > $OUT/bench$i.json 2> $OUT/bench$i.err
  echo "pass $i ($CTRS): rc $?"
  find $OUT/raw$i -name '*counter_collection.csv' -exec cp {} $OUT/counters$i.csv \;
  rm -rf $OUT/raw$i
done
python3 $REPO/tools/pmc_r03_mem_json.py $OUT $FRAMES
