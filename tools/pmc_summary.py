#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs under a directory: per counter, mean over the
escape_kernel dispatches (first dispatch dropped)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "escape_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d)
    for k, v in sorted(acc.items()):
        vv = v[1:] if len(v) > 1 else v
        print(f"  {k:28s} {sum(vv)/len(vv):16.1f}  (n={len(vv)})")
