cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O
for round in 1 2; do
for t in fuse=1 fuse=0; do
PT_TUNE=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $O/i_$t.json 2>$O/i_$t.err
python - <<PY
import json
d=json.loads(open("$O/i_$t.json").readline())
i=d["interactive"]
print("$t", "20 steps", round(d["value"],1), "interactive ms/frame", round(i["ms_per_frame"],3), "pipelined", round(i["pipelined_ms_per_frame"],3))
PY
for sh in 0/8 3/4; do
PT_TUNE=$t timeout 300 python bench.py --emulate-shard $sh --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-interactive > $O/s_$t.json 2>$O/s_$t.err
python - <<PY
import json
d=json.loads(open("$O/s_$t.json").readline())
print("$t", "shard $sh 20 steps ms/frame", round(d["ms_per_step"],4), [round(x) for x in d["repeats"]])
PY
done
PT_TUNE=$t timeout 300 python bench.py --emulate-shard 0/8 --steps 128 --warmup 5 --no-cpu-baseline --no-profile --no-interactive > $O/s_$t.json 2>$O/s_$t.err
python - <<PY
import json
d=json.loads(open("$O/s_$t.json").readline())
print("$t", "shard 0/8 128 steps ms/frame", round(d["ms_per_step"],4), [round(x) for x in d["repeats"]])
PY
done; done
