#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import csv, glob, sys, os, collections
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("#", os.path.relpath(f, d))
    for k, cs in agg.items():
        if "m17dev" not in k: continue
        print(k, "dispatches", max(len(v) for v in cs.values()))
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} mean/dispatch {sum(v)/len(v):.4g}")
