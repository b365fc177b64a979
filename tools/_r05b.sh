cd $GRAFT_REPO_ROOT; O=gpurun_out/r05b; mkdir -p $O
for t in fuse=1 fuse=0; do
PT_TUNE=$t timeout 300 python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-interactive > $O/b96_$t.json 2>$O/b96_$t.err
python - <<PY
import json
d=json.loads(open("$O/b96_$t.json").readline())
s=d["serialised"]
print("$t", round(d["value"],1), {k:round(v,2) for k,v in s["stage_ms"].items()}, "wall", round(s["wall_ms"],1), "launches", s["launches_per_stage"], s.get("launches_fused"), s["launches_tail"])
PY
done
