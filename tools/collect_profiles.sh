#!/bin/bash
# Runs the rocprofv3 passes whose summaries are committed under profiles/ (on the GPU box, via gpurun).
# Counters are collected in their own passes (no tracing domains mixed with --pmc).
# usage: tools/collect_profiles.sh <tag> [bench args...]
set -u
TAG="$1"; shift
OUT=/root/repo/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-periodicity $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/pmc_sq_a" -- $BENCH > "$OUT/pmc_sq_a.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 \
  --output-format csv -d "$OUT/pmc_sq_b" -- $BENCH > "$OUT/pmc_sq_b.log" 2>&1 || exit 1
grep -h '"metric"' "$OUT"/*.log | head -5
echo collected "$OUT"
