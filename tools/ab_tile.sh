#!/bin/bash
# A/B of library builds on the per-launch durations of staged renders (tools/tile_time.py), interleaved in ONE GPU session.
# usage: tools/ab_tile.sh rounds "workload ..." libA.so libB.so [...]   (paths relative to the repo root)
R="$1"; WL="$2"; shift 2
for rep in 1 2; do
  for L in "$@"; do
    echo "== $(basename "$L")"
    FR_LIB_PATH="$PWD/$L" python3 tools/tile_time.py "$R" $WL 2>/dev/null
  done
done
