#!/bin/bash
# SQ instruction-mix counters of one bench workload under a given library build: tools/pmc_lib.sh <tag> <lib.so> <workload>
set -u
TAG="$1"; LIB="$2"; WL="$3"
OUT=/root/repo/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp FR_LIB_PATH="/root/repo/$LIB"
BENCH="python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-periodicity --workload $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
  --output-format csv -d "$OUT/pmc_sq_b" -- $BENCH > "$OUT/pmc_sq_b.log" 2>&1 || exit 1
python3 /root/repo/tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
grep "fr::" "$OUT"/stats/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-160 >> "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +200k -delete
cat "$OUT/summary.txt"
