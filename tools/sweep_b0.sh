#!/bin/bash
# Tile-pass budget (stage_first) sweep on the workloads whose tile pass is a large share of the frame; periodicity off.
# usage (on the GPU box): tools/sweep_b0.sh [rounds]
R="${1:-8}"
for w in c2 c3 c5 reset hd1k; do
  case $w in
    c3) V="32 48 64 80 96 128 160 256";;
    c5) V="32 48 64 96 128 144 192 256";;
    *)  V="16 32 48 64 96 128";;
  esac
  args=("periodicity=-1")
  for b in $V; do args+=("periodicity=-1,stage_first=$b"); done
  python3 tools/sweep_opts.py $w $R "${args[@]}" 2>/dev/null
done
