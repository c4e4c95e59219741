#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if pat and pat not in k:
                continue
            acc[k[:60]][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
for k, cs in acc.items():
    print(k)
    for c, vals in sorted(cs.items()):
        per = defaultdict(float)
        for d, v in vals:
            per[d] += v
        xs = list(per.values())
        print(f"   {c:36s} {sum(xs)/len(xs):18.1f}  (n={len(xs)})")
