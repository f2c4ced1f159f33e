#!/bin/bash
# A/B of library builds on the cost of 8 one-eighth shares of C2 rendered one after the other (tools/shard_cost.py), interleaved
# usage: tools/ab_share.sh "extra shard_cost args" libA.so libB.so ...
X="$1"; shift
for rep in 1 2 3; do
  for L in "$@"; do
    printf "%-12s " "$(basename "$L")"
    FR_LIB_PATH="$PWD/$L" python tools/shard_cost.py --opt periodicity=-1 --only 8 $X 2>/dev/null | grep "plane=rgba"
  done
done
