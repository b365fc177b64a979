#!/bin/bash
# GPU box: the round's evidence in one call on the final binary.  Order matters: the PMC passes and the instruction-mix ceilings first (their summaries are
# copied into profiles/ ON THE BOX, so that the bench lines that follow price their evidence fields from this binary's own counters), then the driver's command
# with every leg, the 256-step line, full lines of the other BASELINE configurations, the kernel-trace statistics of the driver's command, every rank's shard
# of N = 1, 2, 4, 8, the two-rank line on one device, the GPU test suite and smoke().      gpurun -- 'bash tools/final_evidence.sh r06'
# Everything lands in gpurun_out/<rNN>_final/ under the names it is committed with in profiles/ (copy by hand).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
R=${1:-r06}
T=${R}_final
O=$REPO/gpurun_out/$T; mkdir -p $O
bash tools/gpu_run.sh $T valumix pmc:96 binders:64 pmcwl:c2:32 pmcwl:c4:24 pmcwl:c5:24 pmcwl:c5:24:two
cp $O/valu_mix.txt $O/${R}_valu_mix_raw.txt 2>/dev/null
# the box's profiles/ take this binary's summaries (valu_mix.txt keeps the committed reading: only its measured lines are replaced when they parse)
for k in traffic valu cache binders; do [ -s $O/$k.json ] && cp $O/$k.json profiles/${R}_$k.json && cp $O/$k.json $O/${R}_$k.json; done
for w in c2 c4 c5 c5two; do for k in traffic valu binders; do [ -s $O/${k}_$w.json ] && cp $O/${k}_$w.json profiles/${R}_${k}_$w.json && cp $O/${k}_$w.json $O/${R}_${k}_$w.json; done; done
cp $O/pmc.txt $O/${R}_pmc_passes.txt 2>/dev/null
python - "$O/valu_mix.txt" "profiles/${R}_valu_mix.txt" "$O/${R}_valu_mix.txt" <<'PY'
import os, sys
fresh, committed, out = sys.argv[1:4]
if os.path.exists(fresh) and "k_shade mix (k_shade: 4)" in open(fresh).read():
    reading = ""
    if os.path.exists(committed):
        txt = open(committed).read()
        i = txt.find("\nReading")
        reading = txt[i:] if i >= 0 else ""
    new = open(fresh).read().rstrip("\n") + "\n" + reading
    open(committed, "w").write(new)
    open(out, "w").write(new)
PY
timeout 1200 python tools/shard_table.py --workload c3 --steps 20 128 256 > $O/${R}_shard_table_c3.json 2> $O/shard_c3.err; cp $O/${R}_shard_table_c3.json profiles/ 2>/dev/null
timeout 1200 python tools/shard_table.py --workload c4 --steps 20 128 > $O/${R}_shard_table_c4.json 2> $O/shard_c4.err; cp $O/${R}_shard_table_c4.json profiles/ 2>/dev/null
bash tools/gpu_run.sh $T bench20 bench256 prof20 line:c2:64 line:c4:20 line:c5:32 line:c5:32:two
PT_BENCH_SAME_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-profile --no-interactive > $O/${R}_two_ranks_one_gpu_bench_20.json 2> $O/two_ranks.err; echo "two ranks on one device: rc $?" | tee -a $O/log.txt
bash tools/gpu_run.sh $T tests smoke
