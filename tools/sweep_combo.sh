#!/bin/bash
R="${1:-8}"
P="periodicity=-1"
python3 tools/sweep_opts.py c2 $R "$P" "$P,stream_run_max=1" "$P,stream_run_max=1,pool_refill_at=16" "$P,stream_run_max=1,stage_first=96" "$P,stream_run_max=1,pool_refill_at=16,stage_first=96" "$P,stream_run_max=1,stage_first=64" 2>/dev/null
python3 tools/sweep_opts.py c3 $R "$P" "$P,stream_run_max=1" "$P,stream_run_max=1,pool_refill_at=16" "$P,stage_first=96" "$P,stream_run_max=1,stage_first=96" "$P,pool_refill_at=16,stage_first=96" 2>/dev/null
python3 tools/sweep_opts.py c5 $R "$P" "$P,stream_run_max=1" "$P,stream_run_max=1,pool_refill_at=16" "$P,pool_refill_at=16,stage_first=192" "$P,pool_refill_at=12" "$P,pool_refill_at=16,stream_workgroups_per_cu=7" 2>/dev/null
python3 tools/sweep_opts.py reset $R "$P" "$P,stream_run_max=1" "$P,stream_run_max=1,pool_refill_at=16" "$P,stream_run_max=1,stage_first=96" 2>/dev/null
python3 tools/sweep_opts.py hd1k $R "$P" "$P,stream_run_max=1" "$P,stream_run_max=1,pool_refill_at=16" 2>/dev/null
python3 tools/sweep_opts.py c4 2 "$P" "$P,stream_run_max=1" "$P,stream_run_max=1,pool_refill_at=16" 2>/dev/null
for w in c2 c5 reset; do python3 tools/sweep_opts.py $w $R "" "stream_run_max=1" "stream_run_max=1,pool_refill_at=16" 2>/dev/null; done
