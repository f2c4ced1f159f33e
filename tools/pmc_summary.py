#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs under directories: per kernel and counter, the mean
over dispatches (first dispatch of each kernel dropped)."""
import csv, glob, sys, collections, re
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"fr::(\w+)<([^>]*)>", r["Kernel_Name"])
            if m:
                acc[m.group(1) + "<" + m.group(2) + ">"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d)
    for kn, cs in sorted(acc.items()):
        print(" ", kn)
        for k, v in sorted(cs.items()):
            vv = v[len(v)//4:] if len(v) > 4 else v
            print(f"    {k:28s} {sum(vv)/len(vv):16.1f}  (n={len(vv)})")
