#!/bin/bash
# GPU box: sweep launch-policy knobs, print per-stage ms/frame
for T in "$@"; do
  echo "== PT_TUNE=$T"; PT_TUNE=$T PT_PROF=${PT_PROF:-0} python tools/gpu_stats.py ${FRAMES:-16} 2>&1 | grep -E "WALL|ms/frame"
done
