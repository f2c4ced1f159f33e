#!/bin/bash
# A/B of two library builds over the round's workloads, periodicity off (the headline) and default.
# usage (GPU box): tools/ab_round.sh build/ab/base.so build/ab/work.so [rounds] [workloads...]
A="$1"; B="$2"; R="${3:-3}"; shift 3
WL="${@:-c2 c3 c5 reset hd1k}"
for opts in "periodicity=-1" ""; do
  echo "== options: ${opts:-defaults}"
  AB_OPTS="$opts" tools/ab_libs.sh "$A" "$B" "$R" $WL
done
