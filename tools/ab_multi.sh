#!/bin/bash
# interleaved A/B/C... of several library builds.  usage (GPU box): tools/ab_multi.sh rounds "opts" "workloads" lib1.so lib2.so ...
R="$1"; OPTS="$2"; WL="$3"; shift 3
for w in $WL; do
  for i in $(seq 1 "$R"); do
    for L in "$@"; do
      printf "%s %s " "$w" "$(basename "$L")"
      FR_LIB_PATH="$PWD/$L" python tools/sweep_opts.py "$w" 15 "$OPTS" 2>/dev/null | tail -1
    done
  done
done
