#!/bin/bash
# GPU box: the round's evidence in one call -- PMC passes on the final binary (-> profiles/rNN_{traffic,valu,cache}.json, which bench.py's evidence fields
# read), the driver's command line with every leg, the 256-step line, the kernel-trace statistics of the driver's command, every rank's shard of
# N = 1, 2, 4, 8 at the configurations' own spp, the GPU test suite.      gpurun -- 'bash tools/final_evidence.sh r05'
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
R=${1:-r05}
O=$REPO/gpurun_out/${R}_final; mkdir -p $O
bash tools/gpu_run.sh ${R}_final pmc:96
for k in traffic valu cache; do [ -s $O/$k.json ] && cp $O/$k.json profiles/${R}_$k.json && cp $O/$k.json $O/${R}_$k.json; done
cp $O/pmc.txt $O/${R}_final_pmc_passes.txt 2>/dev/null
bash tools/gpu_run.sh ${R}_final bench20 bench256 prof20
timeout 900 python tools/shard_table.py --workload c3 --steps 20 128 256 > $O/${R}_shard_table_c3.json 2> $O/shard_c3.err; tail -c 600 $O/${R}_shard_table_c3.json
timeout 900 python tools/shard_table.py --workload c4 --steps 20 128 > $O/${R}_shard_table_c4.json 2> $O/shard_c4.err; tail -c 400 $O/${R}_shard_table_c4.json
bash tools/gpu_run.sh ${R}_final tests smoke
