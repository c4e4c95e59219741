#!/usr/bin/env python3
"""profiles/lds_bank_conflicts.json from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass: per kernel, conflict cycles / LDS-array cycles (sums over all dispatches)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the hash of the kernel sources the record belongs to)

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
           "csrc_sha": bench.sources_sha(),
           "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                                    text=True).stdout.strip(),
           "rates": rates}, open(out, "w"), indent=1)
for k, v in rates.items():
    print(f"{v['rate']:.4f}  {k}")
