#!/bin/bash
# GPU box: 8-wide fp16 structure with ONE copy of the triangle code: variants; then the RCCL gather inside a process that has torch's RCCL initialised
# (bench.py with one forced rank = the N > 1 flow without peers).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03e
mkdir -p $OUT
cd $REPO
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.txt
for S in 20 96; do
  echo "== steps $S"
  STEPS=$S BENCH_FLAGS="--no-profile --no-interactive" bash tools/variants_bench.sh old4wide default w4 2>&1 | tee -a $OUT/variants_$S.txt
done
echo "== forced single-rank torch.distributed + native gather"
PT_BENCH_FORCE_DIST=1 NCCL_DEBUG=WARN timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-interactive > $OUT/force_dist.json 2> $OUT/force_dist.err; echo "rc $?"; tail -c 600 $OUT/force_dist.json; grep -v "alt_rsmi\|^$" $OUT/force_dist.err | tail -15
