#!/bin/bash
# The numbers DESIGN.md quotes for a round, from ONE gpurun session: every bench workload (one frame at a time, headline leg
# with periodicity off + the periodicity leg), per-launch durations of the staged workloads against the general tile kernel
# with 8 shards, and the fused launch against the two-launch schedule.  usage: tools/round_numbers.sh r02
R="$1"; cd /root/repo || exit 1
OUT=gpurun_out/${R}_numbers; mkdir -p "$OUT"
: > "$OUT/all_workloads.jsonl"
for w in c1 c2 c2_reset c3 c4 c5 hd hd1k uhd uhd1k deepzoom trap stripes colorize export8 export16; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep '"metric"' >> "$OUT/all_workloads.jsonl"
done
python3 - "$OUT/all_workloads.jsonl" <<'P'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); p = d.get("periodicity") or {}
    rv = d.get("roofline_valu") or {}
    print(f"{d['metric']:28s} {d['ms_per_step']:9.4f} ms {d['value']:10.1f} Mpx/s  issue frac {rv.get('frac')}  8d frac {(rv.get('survey_8d') or {}).get('frac')}  iter/px {rv.get('mean_iterations_per_pixel')}  periodicity: {p.get('ms_per_step')} ms {p.get('value')}")
P
echo "== per-launch durations (tools/tile_time.py): defaults | general tile kernel, 8 shards"
python3 tools/tile_time.py 15 c2 c2:tile_kernel=1,shards=8 c3 c3:tile_kernel=1,shards=8 c5 c5:tile_kernel=1,shards=8 far far:tile_kernel=1,shards=8 far:plane=iter far:plane=iter,tile_kernel=1,shards=8 uhd1k uhd1k:tile_kernel=1,shards=8 2>/dev/null
