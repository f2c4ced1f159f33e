#!/bin/bash
# PMC instruction mix of the tile pass's per-pixel skeleton: the far-exterior view (every pixel escapes at i <= 1).
# usage (on the GPU box): tools/pmc_far_exterior.sh [plane]
PLANE="${1:-iter}"
OUT=/root/repo/gpurun_out/pmc_far_$PLANE
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/a" -- python3 /root/repo/tools/far_exterior.py "$PLANE" 6 > "$OUT/a.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 \
  --output-format csv -d "$OUT/b" -- python3 /root/repo/tools/far_exterior.py "$PLANE" 6 > "$OUT/b.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG \
  --output-format csv -d "$OUT/c" -- python3 /root/repo/tools/far_exterior.py "$PLANE" 6 > "$OUT/c.log" 2>&1 || echo "pass c failed (counter names?)"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in "abc":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fr::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in sorted(cs.items()):
            print("%-62s %-24s %.4g" % (k, c, sum(v[1:]) / max(1, len(v) - 1)))
PY
