#!/usr/bin/env python3
"""One image through the captured path, a few replays: run under `rocprofv3 --kernel-trace` and
feed the kernel trace to tools/b1_timeline_summary.py for the per-kernel start/end of a replay."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from objectdetection_amd.pipeline import RegionProposalPath  # noqa: E402

dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
fmap, score, coord, gt, label = bench.make_inputs(cfg, 1, 512, dev, 1000)
path = RegionProposalPath(cfg, 1, 512, dev, train=False)
for _ in range(3):
    path.run(fmap, score, coord)
torch.cuda.synchronize()
g = path.capture(fmap, score, coord)
for _ in range(6):
    g.replay()
    torch.cuda.synchronize()
