#!/bin/bash
# one against two survivor records per lane of the lane pool, periodicity off and on.  usage (GPU box): tools/ab_pool.sh [rounds]
R="${1:-8}"
for w in c2 c3 c5 reset hd1k; do
  python3 tools/sweep_opts.py $w $R "periodicity=-1,pool_records=1" "periodicity=-1,pool_records=2" "pool_records=1" "pool_records=2" 2>/dev/null
done
python3 tools/sweep_opts.py c4 2 "periodicity=-1,pool_records=1" "periodicity=-1,pool_records=2" "pool_records=1" "pool_records=2" 2>/dev/null
