#!/bin/bash
# All rocprofv3 collections of a round in one gpurun call: kernel stats + separate PMC passes per workload
# (tools/collect_profiles.sh), plus the kernel stats of the DEFAULT bench command.  usage: tools/collect_round.sh r02
R="$1"
cd /root/repo || exit 1
for w in c2 c3 c5 c4 deepzoom trap stripes colorize export8 export16; do
  echo "=== collecting ${R}_$w"; date +%T
  tools/collect_profiles.sh "${R}_$w" --workload "$w" 2>&1 | tail -2 || echo "collection of $w failed"
done
echo "=== default command"; date +%T
OUT=/root/repo/gpurun_out/profiles_${R}_default; mkdir -p "$OUT"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 /root/repo/bench.py > "$OUT/stats.log" 2>&1)
grep -h '"metric"' "$OUT/stats.log" | cut -c1-300
find /root/repo/gpurun_out -name "*_kernel_trace.csv" -size +2M -delete
find /root/repo/gpurun_out -name "*_counter_collection.csv" -size +4M -delete
du -sh /root/repo/gpurun_out | tail -1
