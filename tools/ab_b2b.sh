#!/bin/bash
# A/B of library builds on back-to-back frames (tools/b2b_time.py), interleaved in ONE GPU session.
# usage: tools/ab_b2b.sh frames reps "workload ..." libA.so libB.so [...]
F="$1"; R="$2"; WL="$3"; shift 3
for rep in 1 2 3; do
  for L in "$@"; do FR_LIB_PATH="$PWD/$L" python3 tools/b2b_time.py "$F" "$R" $WL 2>/dev/null; done
done
