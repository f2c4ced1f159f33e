#!/usr/bin/env python3
"""Turns gpurun_out/profiles_<tag>/ (rocprofv3 CSVs) into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_pmc.json              per-counter mean over the timed escape_kernel dispatches
  profiles/pmc_traffic.json            HBM bytes per launch, read by bench.py ("roofline.traffic")
usage: tools/summarize_profiles.py <tag> <workload>"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(src + "/stats/**/*_kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
pmc = {}
for d in ("pmc_write", "pmc_fetch", "pmc_sq_a", "pmc_sq_b"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{src}/{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "escape_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count")}
    for k, v in acc.items():
        vv = v[1:]      # first dispatch of bench.py also writes the iter plane: drop it
        pmc[k] = {"mean": sum(vv) / len(vv), "min": min(vv), "max": max(vv), "dispatches": len(vv)}
pmc["_kernel"] = meta
line = [l for l in open(os.path.join(src, "stats.log")) if l.startswith('{"metric"')]
if line:
    pmc["_bench_line_under_kernel_trace"] = json.loads(line[0])
json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
# HBM traffic per launch, as MI355X_MICROARCH.md prescribes: WRITE_SIZE is exact for 16 B/lane streaming
# stores; FETCH_SIZE reports half the bytes on gfx950 -> doubled; both are in KiB.
w, f = pmc["WRITE_SIZE"]["mean"], pmc["FETCH_SIZE"]["mean"]
traffic_path = os.path.join(dst, "pmc_traffic.json")
t = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
t[workload] = {"hbm_bytes_per_launch": int(round((w + 2.0 * f) * 1024)), "WRITE_SIZE_KiB": w, "FETCH_SIZE_KiB_raw": f,
               "source": f"profiles/{tag}_pmc.json", "note": "WRITE_SIZE + 2*FETCH_SIZE (gfx950 FETCH_SIZE counts half), KiB -> bytes"}
json.dump(t, open(traffic_path, "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read()[:600])
print(json.dumps({k: v["mean"] for k, v in pmc.items() if not k.startswith("_")}, indent=1))
print(t[workload])
