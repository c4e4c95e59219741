#!/usr/bin/env python3
"""RoI forward for a few images: the per-RoI kernel against the band kernel with the image's RoIs
split into ranges (FRCNN_ROI_BAND_SPLIT) -- HIP events, 600x1000 workload."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from objectdetection_amd import ops, _lib  # noqa: E402
from objectdetection_amd.pipeline import RegionProposalPath  # noqa: E402

dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
for B in (1, 2, 4, 8):
    fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 512, dev, 1000)
    path = RegionProposalPath(cfg, B, 512, dev, train=False)
    path.run(fmap, score, coord)
    ref = path.pooled.clone()
    res = []
    for min_wg, split in ((1 << 30, 0), (1, 1), (1, 2), (1, 3), (1, 5), (1, 7)):
        _lib.set_tuning("roi_band_min_wg", min_wg)
        _lib.set_tuning("roi_band_split", split)
        fn = lambda: ops.roi_crop_pool_fwd(fmap, path.proposed, cfg.crop, cfg.pool, out=path.pooled,
                                           workspace=path.roi_ws)
        ms = bench.timed_alone(fn, dev, reps=30, skip=10)
        same = bool(torch.equal(path.pooled, ref))
        res.append(f"{'per-RoI' if split == 0 else 'band/' + str(split)} {ms * 1e3:.0f} us{'' if same else ' MISMATCH'}")
    print(f"B={B}: " + "  ".join(res), flush=True)
