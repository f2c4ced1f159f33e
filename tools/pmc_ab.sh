#!/bin/bash
# SQ counter passes (no tracing domains mixed with --pmc) of one bench workload under an option set.
# usage: tools/pmc_ab.sh <tag> <workload> "<options>"     -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG="$1"; WL="$2"; OPTS="$3"
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-periodicity --workload $WL --options $OPTS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/pmc_sq_a" -- $BENCH > "$OUT/pmc_sq_a.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 \
  --output-format csv -d "$OUT/pmc_sq_b" -- $BENCH > "$OUT/pmc_sq_b.log" 2>&1 || exit 1
python3 /root/repo/tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cat {} \; >> "$OUT/summary.txt"
grep -h '"metric"' "$OUT"/stats.log | cut -c1-400 >> "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +200k -delete
cat "$OUT/summary.txt"
