#!/bin/bash
# GPU box: ONE parameterised script for the round's GPU calls (replaces the per-call gpu_r03*.sh scripts of round 3).
#   gpurun -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'        output -> gpurun_out/<tag>/
# steps (run in the order given; each under its own `timeout`):
#   cold20              the driver's command with PT_TUNE=warm=0 and no evidence legs: what a first process on a fresh box does without the slot warm-up
#   bench20 | bench256  the driver's command line / the 256-step line (both with the CPU leg: parity in the line)
#   line:<workload>:<steps>[:two]  a full line of c2 / c4 / c5 (flat or two-level structure)
#   quick:<steps>       bench without evidence legs (rate only)
#   prof20              rocprofv3 --kernel-trace --stats over the driver's command line -> kernel_stats_bench20.csv
#   tune:<steps>:<A>;<B>;...    bench under each PT_TUNE string, two alternating rounds ("-" = the defaults)
#   libs:<steps>:<a>,<b>,...    bench with each libptmi variant of vk_raytrace_amd/variants/ ("default" = the product), two alternating rounds
#   libs5:<steps>:<a>,<b>,...   the same on the C5 stand-in with the two-level structure
#   tests[:<-k expression>]     pytest -m gpu
#   smoke               __graft_entry__.smoke()
#   pmcwl:<workload>:<frames>  reduced PMC passes on c2 / c4 / c5 -> traffic_<wl>.json, valu_<wl>.json, binders_<wl>.json
#   valumix             tools/valu_mix.hip: the VALU issue ceiling for the kernels' instruction mixes -> valu_mix.txt
#   pmc[:<frames>]      tools/pmc_passes.sh passes -> traffic / valu / cache json (copied to profiles/r04_*.json by hand)
#   binders[:<frames>[:<PT_TUNE>]]  tools/pmc_binders.sh passes -> binders.json (lane occupancy, I-cache, VMEM issue, waits per kernel)
#   shards:<wl>:<steps> every rank's shard of N = 1, 2, 4, 8 on this one GPU (bench.py --emulate-shard R/N); max over ranks per N
#   shardtune:<wl>:<steps>:<R/N>:<A>;<B>;...   one rank's shard under each PT_TUNE string, two alternating rounds
#   sh:<command>        anything else
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
TAG=$1; shift
O=$REPO/gpurun_out/$TAG; mkdir -p $O
val() { python -c 'import sys,json
try:
    d=json.loads(sys.stdin.readline()); r=d.get("repeats") or []
    print(round(d["value"],1), "Msamples/s  windows:", [round(x) for x in r], " parity:", (d.get("parity") or {}).get("l2"))
except Exception as e: print("unreadable", e)'; }
for step in "$@"; do
  IFS=: read -r kind a b <<< "$step"
  echo "=== $step" | tee -a $O/log.txt
  case $kind in
    cold20)  PT_TUNE=warm=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-interactive > $O/cold20.json 2> $O/cold20.err; val < $O/cold20.json | tee -a $O/log.txt ;;
    bench20) timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "rc $?" | tee -a $O/log.txt; val < $O/bench_20.json | tee -a $O/log.txt ;;
    bench256) timeout 900 python bench.py --gpus 1 --steps 256 --warmup 8 > $O/bench_256.json 2> $O/bench_256.err; val < $O/bench_256.json | tee -a $O/log.txt ;;
    line)    # line:<workload>:<steps>[:two]  a full line (parity, cpu_baseline, roofline) of another BASELINE configuration
             IFS=: read -r ln_steps ln_accel <<< "$b"; ACC=flat; SUF=""; [ "$ln_accel" = "two" ] && ACC=two && SUF=_two
             timeout 1500 python bench.py --gpus 1 --workload $a --accel $ACC --steps $ln_steps --warmup 5 > $O/bench_${a}_${ln_steps}$SUF.json 2> $O/bench_${a}_${ln_steps}$SUF.err; echo "rc $?" | tee -a $O/log.txt; val < $O/bench_${a}_${ln_steps}$SUF.json | tee -a $O/log.txt ;;
    quick)   timeout 300 python bench.py --steps $a --warmup 8 --no-cpu-baseline --no-profile --no-interactive > $O/quick_$a.json 2> $O/quick_$a.err; val < $O/quick_$a.json | tee -a $O/log.txt ;;
    prof20)  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof20 -o p -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-interactive > $O/bench_20_profiled.json 2> $O/prof20.err)
             find $O/prof20 -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_bench20.csv \; ; rm -rf $O/prof20; head -12 $O/kernel_stats_bench20.csv | cut -c1-150 | tee -a $O/log.txt ;;
    tune)    for round in 1 2; do IFS=';' read -ra TS <<< "$b"; for t in "${TS[@]}"; do tt=$t; [ "$t" = "-" ] && tt=""
               echo -n "steps $a PT_TUNE=$tt : " | tee -a $O/log.txt
               PT_TUNE=$tt timeout 300 python bench.py --steps $a --warmup 8 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | val | tee -a $O/log.txt; done; done ;;
    libs)    for round in 1 2; do IFS=',' read -ra VS <<< "$b"; for v in "${VS[@]}"; do L=""; [ "$v" != default ] && L="$REPO/vk_raytrace_amd/variants/libptmi_$v.so"
               echo -n "steps $a lib $v : " | tee -a $O/log.txt
               PT_LIB=$L timeout 300 python bench.py --steps $a --warmup 8 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | val | tee -a $O/log.txt; done; done ;;
    libs5)   for round in 1 2; do IFS=',' read -ra VS <<< "$b"; for v in "${VS[@]}"; do L=""; [ "$v" != default ] && L="$REPO/vk_raytrace_amd/variants/libptmi_$v.so"
               echo -n "c5 two-level steps $a lib $v : " | tee -a $O/log.txt
               PT_LIB=$L timeout 300 python bench.py --workload c5 --accel two --steps $a --warmup 8 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | val | tee -a $O/log.txt; done; done ;;
    tests)   if [ -n "$a" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$a" > $O/gputest.txt 2>&1; else timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.txt 2>&1; fi; tail -5 $O/gputest.txt | tee -a $O/log.txt ;;
    smoke)   timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee -a $O/log.txt ;;
    pmc)     PMC_FRAMES=${a:-96} timeout 1500 bash tools/pmc_passes.sh $TAG > $O/pmc.txt 2>&1; for k in traffic valu cache; do [ -s gpurun_out/pmc_$TAG/$k.json ] && cp gpurun_out/pmc_$TAG/$k.json $O/$k.json; done; tail -4 $O/pmc.txt | tee -a $O/log.txt ;;
    pmcwl)   # pmcwl:<workload>:<frames>[:two]  the reduced pass set (traffic, instruction mix, cycles, lane occupancy) on another BASELINE configuration -> <kind>_<workload>.json
             IFS=: read -r pw_frames pw_accel <<< "$b"; PACC=flat; PSUF=$a; [ "$pw_accel" = "two" ] && PACC=two && PSUF=${a}two
             PMC_WORKLOAD=$a PMC_ACCEL=$PACC PMC_FRAMES=${pw_frames:-32} PMC_SET=min timeout 1500 bash tools/pmc_passes.sh ${TAG}_$PSUF > $O/pmc_$PSUF.txt 2>&1
             PMC_WORKLOAD=$a PMC_ACCEL=$PACC PMC_FRAMES=${pw_frames:-32} PMC_SET=min timeout 900 bash tools/pmc_binders.sh ${TAG}_$PSUF >> $O/pmc_$PSUF.txt 2>&1
             for k in traffic valu binders; do [ -s gpurun_out/pmc_${TAG}_$PSUF/$k.json ] && cp gpurun_out/pmc_${TAG}_$PSUF/$k.json $O/${k}_$PSUF.json; done; tail -6 $O/pmc_$PSUF.txt | tee -a $O/log.txt ;;
    binders) PMC_FRAMES=${a:-64} PMC_TUNE=${b:-inflight=1,warm=0} timeout 1500 bash tools/pmc_binders.sh $TAG > $O/binders.txt 2>&1; cp gpurun_out/pmc_$TAG/binders.json $O/binders.json 2>/dev/null; tail -14 $O/binders.txt | tee -a $O/log.txt ;;
    shards)  for n in 1 2 4 8; do worst=0; for ((r = 0; r < n; r++)); do
               timeout 300 python bench.py --workload $a --emulate-shard $r/$n --steps $b --warmup 5 --no-cpu-baseline --no-profile --no-interactive > $O/shard_${a}_${r}of${n}_$b.json 2>/dev/null
               ms=$(python -c "import json; print(json.loads(open('$O/shard_${a}_${r}of${n}_$b.json').readline())['ms_per_step'])" 2>/dev/null || echo 0)
               worst=$(python -c "print(max($worst, $ms))"); done
               echo "$a steps $b N=$n: slowest rank $worst ms/frame" | tee -a $O/log.txt; done ;;
    shardtune) IFS=: read -r st_steps st_rn st_tunes <<< "$b"
             for round in 1 2; do IFS=';' read -ra TS <<< "$st_tunes"; for t in "${TS[@]}"; do tt=$t; [ "$t" = "-" ] && tt=""
               echo -n "$a shard $st_rn steps $st_steps PT_TUNE=$tt : " | tee -a $O/log.txt
               PT_TUNE=$tt timeout 300 python bench.py --workload $a --emulate-shard $st_rn --steps $st_steps --warmup 5 --no-cpu-baseline --no-profile --no-interactive 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), "ms/frame  windows:", [round(x) for x in d["repeats"]])' | tee -a $O/log.txt; done; done ;;
    valumix) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/valu_mix.hip -o /tmp/valu_mix 2>/dev/null && timeout 300 /tmp/valu_mix | tee $O/valu_mix.txt | tee -a $O/log.txt ;;
    sh)      timeout 900 bash -c "${step#sh:}" 2>&1 | tail -20 | tee -a $O/log.txt ;;
    *)       echo "unknown step $step" | tee -a $O/log.txt ;;
  esac
done
