cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "path_traced or tiny or punctual or sun_and or samples_per or rtx_pipeline or any_hit or sponza_like or c1_quad" 2>&1 | tail -3
run() { # tune, extra args, label
PT_TUNE=$1 timeout 90 python bench.py $2 --no-cpu-baseline --no-profile > $O/t.json 2>$O/t.err
python - <<PY
import json
try:
    d=json.loads(open("$O/t.json").readline())
    i=d.get("interactive") or {}
    print("$3 PT_TUNE=$1 :", round(d["value"],1), "ms/frame", round(d["ms_per_step"],4), [round(x) for x in d["repeats"]], "interactive", round(i.get("ms_per_frame",0),3), "pipelined", round(i.get("pipelined_ms_per_frame",0),3))
except Exception as e: print("$3 $1 failed", e, open("$O/t.err").read()[-300:])
PY
}
for t in wave=0 wave=1 wave=1,tail=262144 wave=1,tail=1048576; do
run $t "--steps 20 --warmup 5" full20
done
for t in wave=0 wave=1 wave=1,tail=262144; do
run $t "--steps 20 --warmup 5 --emulate-shard 0/8 --no-interactive" shard0of8
done
