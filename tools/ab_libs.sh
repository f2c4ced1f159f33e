#!/bin/bash
# A/B of two builds of the library in ONE GPU session (numbers from different gpurun boxes differ by a few %):
# usage: tools/ab_libs.sh libA.so libB.so rounds workload [workload...]   (paths relative to the repo root)
A="$PWD/$1"; B="$PWD/$2"; R="$3"; shift 3
for w in "$@"; do
  for i in $(seq 1 "$R"); do
    for L in "$A" "$B"; do
      printf "%s %s " "$w" "$(basename "$L")"
      FR_LIB_PATH="$L" python tools/sweep_opts.py "$w" 15 "${AB_OPTS:-}" 2>/dev/null | tail -1
    done
  done
done
