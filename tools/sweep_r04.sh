#!/bin/bash
# round 4: tile-pass budget (b0) and refill threshold re-swept with the deferring lane pool
out=gpurun_out/r04_b0_sweep.txt; : > $out
P="periodicity=-1"
python tools/sweep_opts.py c5 7 "$P" "$P,stage_first=32" "$P,stage_first=48" "$P,stage_first=64" "$P,stage_first=96" "$P,stage_first=128" "$P,stage_first=256" >> $out 2>&1
python tools/sweep_opts.py c3 15 "$P" "$P,stage_first=16" "$P,stage_first=32" "$P,stage_first=48" "$P,stage_first=64" "$P,stage_first=112" >> $out 2>&1
python tools/sweep_opts.py c2 15 "$P" "$P,stage_first=16" "$P,stage_first=32" "$P,stage_first=48" "$P,stage_first=64" "$P,stage_first=128" >> $out 2>&1
python tools/sweep_opts.py hd1k 21 "$P" "$P,stage_first=16" "$P,stage_first=48" "$P,stage_first=64" "$P,stage_first=96" >> $out 2>&1
python tools/sweep_opts.py uhd1k 15 "$P" "$P,stage_first=16" "$P,stage_first=32" "$P,stage_first=64" "$P,stage_first=96" >> $out 2>&1
python tools/sweep_opts.py c2 15 "" "stage_first=16" "stage_first=48" "stage_first=64" "stage_first=96" >> $out 2>&1
python tools/sweep_opts.py c5 7 "$P" "$P,pool_refill_at=8" "$P,pool_refill_at=12" "$P,pool_refill_at=24" "$P,pool_refill_at=32" >> $out 2>&1
python tools/sweep_opts.py c3 15 "$P" "$P,pool_refill_at=8" "$P,pool_refill_at=16" "$P,pool_refill_at=32" "$P,pool_refill_at=40" >> $out 2>&1
python tools/sweep_opts.py c2 15 "$P" "$P,pool_refill_at=8" "$P,pool_refill_at=12" "$P,pool_refill_at=24" "$P,pool_refill_at=32" >> $out 2>&1
grep -v amdgpu.ids $out
