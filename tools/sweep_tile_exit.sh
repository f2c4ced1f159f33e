#!/bin/bash
# occupancy exit of the lean tile pass: cost x from, per workload (tools/sweep_opts.py interleaves and shuffles)
cd "$(dirname "$0")/.."
R=${R:-15}
V=("tile_exit=1" "" "tile_exit=24" "tile_exit=32" "tile_exit=64" "tile_exit=96" "tile_exit=48,tile_exit_from=16" "tile_exit=48,tile_exit_from=48" "tile_exit=48,tile_exit_from=64" "tile_exit=96,tile_exit_from=64" "tile_exit=24,tile_exit_from=16")
for w in ${WL:-c2 c3 c5 hd1k uhd1k c2_reset}; do
  for per in ${PER:--1 1}; do
    args=()
    for v in "${V[@]}"; do args+=("periodicity=$per${v:+,$v}"); done
    python3 tools/sweep_opts.py $w $R "${args[@]}" || exit 1
  done
done
