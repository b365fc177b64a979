cd $GRAFT_REPO_ROOT; O=gpurun_out/r05f; mkdir -p $O
run() { # tune, extra args, label
PT_TUNE=$1 timeout 300 python bench.py $2 --no-cpu-baseline --no-profile > $O/t.json 2>$O/t.err
python - <<PY
import json
try:
    d=json.loads(open("$O/t.json").readline())
    i=d.get("interactive") or {}
    print("$3 PT_TUNE=$1 :", round(d["value"],1), "ms/frame", round(d["ms_per_step"],4), [round(x) for x in d["repeats"]], "interactive", round(i.get("ms_per_frame",0),3), "pipelined", round(i.get("pipelined_ms_per_frame",0),3))
except Exception as e: print("$3 $1 failed", e, open("$O/t.err").read()[-300:])
PY
}
for round in 1 2; do
for t in wave=0 wave=1 wave=1,tail=262144 wave=1,tail=1048576 wave=1,tail=4194304; do
run $t "--steps 20 --warmup 5" full20
done
for t in wave=0 wave=1 wave=1,tail=262144 wave=1,tail=1048576; do
run $t "--steps 20 --warmup 5 --emulate-shard 0/8 --no-interactive" shard0of8
done
done
for t in wave=0 wave=1 wave=1,tail=1048576; do
run $t "--steps 96 --warmup 8 --no-interactive" full96
done
