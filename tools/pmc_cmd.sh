#!/bin/bash
# rocprofv3 --pmc passes (one per counter group, groups separated by "/") of an arbitrary python command; per-kernel means.
# usage: tools/pmc_cmd.sh <tag> "C1 C2 / C3 C4 ..." <python script and args ...>   (run on the GPU box via gpurun)
TAG="$1"; GROUPS_="$2"; shift 2
OUT=/root/repo/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
IFS='/' read -ra GS <<< "$GROUPS_"
for g in "${GS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g --output-format csv -d "$OUT/g$i" -- python3 "$@" > "$OUT/g$i.log" 2>&1 || { echo "group $i failed"; tail -5 "$OUT/g$i.log"; }
done
python3 /root/repo/tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +200k -delete
cat "$OUT/summary.txt"
