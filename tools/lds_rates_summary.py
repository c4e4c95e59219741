#!/usr/bin/env python3
"""profiles/lds_bank_conflicts.json from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass: per kernel, conflict cycles / LDS-array cycles (sums over all dispatches)."""
import csv
import glob
import json
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("frcnn::", "")
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
rates = {}
for k, c in sorted(acc.items()):
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    if act > 0 and not k.startswith("at::"):
        rates[k] = {"bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0.0), "lds_active_cycles": act,
                    "rate": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / act}
json.dump({"source": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 tools/lds_rates.py "
                     "(600x1000 workload, B = 64, training path, both top-k variants)",
           "rates": rates}, open(out, "w"), indent=1)
for k, v in rates.items():
    print(f"{v['rate']:.4f}  {k}")
