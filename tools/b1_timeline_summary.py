#!/usr/bin/env python3
"""Per-kernel timeline of the last replay in a rocprofv3 kernel trace of tools/b1_timeline.py:
start (us from the replay's first kernel), duration, gap to the previous kernel's end."""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a replay starts at the first pass of the sort
starts = [i for i, r in enumerate(rows) if ("ts_hist_kernel<0>" in r["Kernel_Name"] or "tm_runs_kernel" in r["Kernel_Name"])]
seq = rows[starts[-1]:]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-48:]
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name}")
    prev_end = e
print(f"total {(prev_end - t0) / 1e3:.1f} us")
