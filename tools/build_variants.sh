#!/bin/bash
# Builds measurement variants of libptmi.so into vk_raytrace_amd/variants/ (git-ignored, travels with gpurun).
#   tools/build_variants.sh name1 "EXTRA flags" name2 "EXTRA flags" ...
# `base` as a name with flags "HEAD" builds the committed sources instead of the working tree.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/vk_raytrace_amd/variants"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  W=$(mktemp -d)
  mkdir -p "$W/vk_raytrace_amd/csrc" "$W/include"
  if [ "$flags" = "HEAD" ]; then
    (cd "$ROOT" && git archive HEAD vk_raytrace_amd/csrc include | tar -x -C "$W")
    flags=""
  else
    cp "$ROOT"/vk_raytrace_amd/csrc/*.h "$ROOT"/vk_raytrace_amd/csrc/*.hip "$ROOT"/vk_raytrace_amd/csrc/*.cpp "$ROOT"/vk_raytrace_amd/csrc/Makefile "$W/vk_raytrace_amd/csrc/"
    cp "$ROOT"/include/* "$W/include/"
  fi
  make -s -C "$W/vk_raytrace_amd/csrc" -j4 EXTRA="$flags" >/dev/null 2>&1 || { echo "variant $name failed"; make -C "$W/vk_raytrace_amd/csrc" EXTRA="$flags" 2>&1 | grep -E "error" | head; exit 1; }
  cp "$W/vk_raytrace_amd/libptmi.so" "$ROOT/vk_raytrace_amd/variants/libptmi_$name.so"
  rm -rf "$W"
  echo "built variant $name ($flags)"
done
