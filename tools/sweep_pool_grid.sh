#!/bin/bash
# lane-pool grid (workgroups per CU) and refill threshold sweeps, periodicity off.  usage (GPU box): tools/sweep_pool_grid.sh [rounds]
R="${1:-8}"
for w in c5 c3 c2; do
  python3 tools/sweep_opts.py $w $R "periodicity=-1" "periodicity=-1,stream_workgroups_per_cu=5" "periodicity=-1,stream_workgroups_per_cu=7" "periodicity=-1,stream_workgroups_per_cu=8" \
     "periodicity=-1,pool_refill_at=16" "periodicity=-1,pool_refill_at=32" "periodicity=-1,pool_refill_at=40" "periodicity=-1,stream_run_max=1" "periodicity=-1,stream_run_max=4" 2>/dev/null
done
