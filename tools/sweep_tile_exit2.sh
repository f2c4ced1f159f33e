#!/bin/bash
# finer sweep around the adopted rule (cost 48, from 64)
cd "$(dirname "$0")/.."
R=${R:-15}
for w in ${WL:-c2 c5}; do
  args=("periodicity=-1,tile_exit=1")
  for c in 32 48 64; do for f in 48 64 80 96; do args+=("periodicity=-1,tile_exit=$c,tile_exit_from=$f"); done; done
  python3 tools/sweep_opts.py $w $R "${args[@]}" || exit 1
done
python3 tools/sweep_opts.py c4 3 "periodicity=-1,tile_exit=1" "periodicity=-1" "periodicity=1,tile_exit=1" "periodicity=1" || exit 1
python3 tools/sweep_opts.py c1 15 "tile_exit=1" "" || exit 1
