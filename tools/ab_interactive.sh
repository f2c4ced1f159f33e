#!/bin/bash
# A/B of two library builds on tools/interactive_time.py (fp32 and fp64 frames at interactive sizes), interleaved.
# usage: tools/ab_interactive.sh libA.so libB.so rounds
A="$PWD/$1"; B="$PWD/$2"; R="$3"
for i in $(seq 1 "$R"); do
  for L in "$A" "$B"; do
    echo "== $(basename "$L")"
    FR_LIB_PATH="$L" python tools/interactive_time.py 2>/dev/null | grep -v amdgpu.ids
  done
done
