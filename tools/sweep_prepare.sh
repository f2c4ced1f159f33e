#!/bin/bash
# control block + coordinate tables: a launch of their own (prepare=1) against the prologue of the lean tile pass (default)
cd "$(dirname "$0")/.."
R=${R:-25}
for w in ${WL:-c1 hd hd1k uhd uhd1k c3 c2 c5}; do
  python3 tools/sweep_opts.py $w $R "prepare=1" "" || exit 1
done
