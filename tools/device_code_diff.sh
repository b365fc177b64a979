#!/bin/bash
# No GPU: is the DEVICE code of the working tree the same as at a given revision?  Compiles every .hip file of both trees to gfx950 assembly
# (hipcc --cuda-device-only -S, the product's flags) and diffs it, ignoring comments and the per-compile unit id.  Used to show that measurement-build
# scaffolding added after a round's last GPU call (code behind -D flags that default to off) left the shipped kernels byte-identical.
#   tools/device_code_diff.sh <git revision>
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:?usage: device_code_diff.sh <revision>}
W=$(mktemp -d)
(cd "$ROOT" && git archive "$REV" vk_raytrace_amd/csrc include | tar -x -C "$W")
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -DSTACK_LDS=24 --cuda-device-only -S"
rc=0
for f in pt_render pt_accel pt_capi pt_sah; do
  (cd "$W/vk_raytrace_amd/csrc" && /opt/rocm/bin/hipcc $F $f.hip -o "$W/$f.old.s" 2>/dev/null) &
  (cd "$ROOT/vk_raytrace_amd/csrc" && /opt/rocm/bin/hipcc $F $f.hip -o "$W/$f.new.s" 2>/dev/null)
  wait
  d=$(diff <(grep -v '^\s*;\|\.file\|\.ident\|__hip_cuid' "$W/$f.old.s") <(grep -v '^\s*;\|\.file\|\.ident\|__hip_cuid' "$W/$f.new.s") | wc -l)
  echo "$f.hip: $(wc -l < "$W/$f.new.s") lines of assembly, $d differing"
  [ "$d" = 0 ] || rc=1
done
rm -rf "$W"
exit $rc
