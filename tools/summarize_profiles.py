#!/usr/bin/env python3
"""Turns gpurun_out/profiles_<tag>/ (rocprofv3 CSVs) into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_pmc.json              per kernel and counter: mean over the timed dispatches; per-frame sums
  profiles/pmc_traffic.json            HBM bytes per frame (all kernels of one render), read by bench.py
usage: tools/summarize_profiles.py <tag> <workload>"""
import collections, csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """gpurun merges new files into gpurun_out/ without removing those of earlier collections: take the latest"""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)
stats = newest(src + "/stats/**/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
pmc = collections.defaultdict(dict)
frame = collections.defaultdict(float)
for d in ("pmc_write", "pmc_fetch", "pmc_sq_a", "pmc_sq_b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in [newest(f"{src}/{d}/**/*_counter_collection.csv")]:
        for r in csv.DictReader(open(f)):
            m = re.search(r"fr::(\w+)(?:<([^>]*)>)?", r["Kernel_Name"])
            if not m or m.group(1) == "clear_words_kernel":
                continue
            kn = f"{m.group(1)}<{m.group(2)}>" if m.group(2) is not None else m.group(1)
            acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
            pmc[kn]["_launch"] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count")}
    for kn, cs in acc.items():
        for k, v in cs.items():
            vv = v[1:] if len(v) > 1 else v     # bench.py's first render also writes the iter plane: drop it
            pmc[kn][k] = {"mean": sum(vv) / len(vv), "min": min(vv), "max": max(vv), "dispatches": len(vv)}
            frame[k] += sum(vv) / len(vv)
out = {"kernels": pmc, "per_frame_sum_over_kernels": dict(frame)}
line = [l for l in open(os.path.join(src, "stats.log")) if l.startswith('{"metric"')]
if line:
    out["bench_line_under_kernel_trace"] = json.loads(line[0])
ks = list(csv.DictReader(open(stats)))
out["kernel_trace_avg_ns"] = {r["Name"][:80]: float(r["AverageNs"]) for r in ks if "fr::" in r["Name"] and "clear_words" not in r["Name"]}
# kernel workloads (colorize / export): the render that produces their input runs once, outside the timed region -- only the
# kernel under test counts as "the frame"
if workload in ("colorize", "export8", "export16"):
    key = {"colorize": "colorize_kernel", "export8": "export_rgb8_kernel", "export16": "export_rgb16_kernel"}[workload]
    out["kernel_trace_avg_ns"] = {k: v for k, v in out["kernel_trace_avg_ns"].items() if key in k}
    frame = collections.defaultdict(float, {k: v["mean"] for kn, cs in pmc.items() if key in kn for k, v in cs.items() if isinstance(v, dict) and "mean" in v})
    out["per_frame_sum_over_kernels"] = dict(frame)
out["kernel_trace_frame_ms"] = sum(out["kernel_trace_avg_ns"].values()) / 1e6
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
# HBM traffic per frame, as MI355X_MICROARCH.md prescribes: WRITE_SIZE is exact for 16 B/lane streaming
# stores; FETCH_SIZE reports half the bytes on gfx950 -> doubled; both are in KiB.
w, f = frame["WRITE_SIZE"], frame["FETCH_SIZE"]
traffic_path = os.path.join(dst, "pmc_traffic.json")
t = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
t[workload] = {"hbm_bytes_per_launch": int(round((w + 2.0 * f) * 1024)), "WRITE_SIZE_KiB": w, "FETCH_SIZE_KiB_raw": f,
               "source": f"profiles/{tag}_pmc.json",
               "note": "per frame = tile pass + lane-pool pass; WRITE_SIZE + 2*FETCH_SIZE (gfx950 FETCH_SIZE counts half), KiB -> bytes"}
# VALU busy from the PMC passes, per kernel and per frame: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the waves,
# GRBM_GUI_ACTIVE counts cycles summed over the 8 XCDs; 1024 SIMDs (256 CUs x 4)
def busy(active_valu, gui):
    return active_valu * 4.0 / (1024.0 * gui / 8.0) if gui else None
vb = {kn: busy(v["SQ_ACTIVE_INST_VALU"]["mean"], v["GRBM_GUI_ACTIVE"]["mean"]) for kn, v in pmc.items()
      if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v}
t[workload]["valu_busy"] = {"frame": busy(frame.get("SQ_ACTIVE_INST_VALU", 0.0), frame.get("GRBM_GUI_ACTIVE", 0.0)),
                            "per_kernel": vb, "clock_ghz": frame.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (out["kernel_trace_frame_ms"] * 1e6) if out["kernel_trace_frame_ms"] else None,
                            "note": "SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)"}
json.dump(t, open(traffic_path, "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read()[:500])
print("frame ms (sum of kernel averages):", out["kernel_trace_frame_ms"])
print(json.dumps(out["per_frame_sum_over_kernels"], indent=1))
print(t[workload])
