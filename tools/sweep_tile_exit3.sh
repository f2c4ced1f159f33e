#!/bin/bash
# with the occupancy exit in place: does a longer tile budget pay now?
cd "$(dirname "$0")/.."
R=${R:-13}
for w in ${WL:-c5 c2 c3 uhd1k}; do
  args=()
  for b in 0 64 96 128 192 256 384; do args+=("periodicity=-1,stage_first=$b"); done
  args+=("periodicity=-1,tile_exit=1" "periodicity=-1,tile_exit=1,stage_first=256")
  python3 tools/sweep_opts.py $w $R "${args[@]}" || exit 1
done
