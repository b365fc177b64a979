#!/bin/bash
# GPU box: kernel timeline of a bench run -> how much of the timed region has 0 / 1 / 2+ kernels in flight, per-kernel busy share
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- python $REPO/bench.py --no-cpu-baseline --no-profile "$@" > $OUT/bench.json 2> $OUT/bench.err
find $OUT/raw -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/raw
python3 - <<PY
import csv
rows=[(r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open("$OUT/kernel_trace.csv"))]
rows=[r for r in rows if r[0].startswith("k_") and r[0] not in ("k_world_tris","k_gather","k_refit","k_emit","k_collapse","k_morton")]
# timed region ~ last 90 % of render kernels: take from the first k_generate after warmup (skip first) to the last k_accumulate
gens=sorted(r[1] for r in rows if r[0]=="k_generate")
t0=gens[1] if len(gens)>1 else gens[0]
t1=max(r[2] for r in rows)
ev=[]
for n,s,e in rows:
    if e<=t0: continue
    ev.append((max(s,t0),1)); ev.append((e,-1))
ev.sort()
cur=0; last=t0; hist={}
for t,d in ev:
    hist[cur]=hist.get(cur,0)+(t-last); last=t; cur+=d
tot=t1-t0
print("window %.1f ms" % (tot/1e6))
for k in sorted(hist): print("  %d kernels in flight: %5.1f %%" % (k, 100.0*hist[k]/tot))
busy={}
for n,s,e in rows:
    if e>t0: busy[n]=busy.get(n,0)+(e-max(s,t0))
for n,v in sorted(busy.items(), key=lambda x:-x[1]): print("  %-14s sum of durations %.1f ms (%.2f x window)" % (n, v/1e6, v/tot))
PY
