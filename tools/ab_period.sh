#!/bin/bash
# cycle closing on (default) against off, interleaved.  usage (GPU box): tools/ab_period.sh [rounds]
R="${1:-10}"
for w in c3 c5 c2 reset hd1k; do python3 tools/sweep_opts.py $w $R "periodicity=-1" "" 2>/dev/null; done
python3 tools/sweep_opts.py c4 2 "periodicity=-1" "" 2>/dev/null
