#!/bin/bash
# round 4: what the event pair around every render ("timing", off by default since 1.1) costs back-to-back frames (bench.py's timed loop)
for w in ${1:-c2 c3 hd1k hd c1}; do for i in 1 2 3; do for o in "timing=1" "timing=0"; do printf "%-6s %-10s " $w $o
python bench.py --workload $w --no-cpu-baseline --no-periodicity --steps 200 --warmup 20 --options "$o" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms/frame %.4f' % d['ms_per_step'])"; done; done; done
