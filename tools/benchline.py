#!/usr/bin/env python3
"""Reads bench.py's JSON line on stdin and prints the fields used when comparing builds."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(" ".join(sys.argv[1:]), f"images/s {d['value']:.0f}  ms/step {d['ms_per_step']:.3f}  "
      f"roi ms {d['roofline']['avg_launch_ms']:.3f}  nms us/img alone {d.get('nms_us_per_image', 0):.1f} "
      f"in step {d.get('nms_us_per_image_in_step', 0):.1f}")
