#!/usr/bin/env python3
"""Print one step of a `rocprofv3 --kernel-trace` run of bench.py: every kernel between two
consecutive RoI forward launches with its start (us after the first one's start), duration,
queue and grid.  usage: tools/step_timeline.py <dir with *_kernel_trace.csv> [which step]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
roi = [i for i, r in enumerate(rows) if "roi_fwd_band_kernel" in r["Kernel_Name"]
       or "roi_fwd_stage_kernel" in r["Kernel_Name"]]
i0, i1 = roi[which], roi[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1000
    e = (int(r["End_Timestamp"]) - t0) / 1000
    print(f"{s:9.1f} {e - s:8.1f}  q{r['Queue_Id']} {r['Kernel_Name'][:48]:48s} "
          f"grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}")
