#!/usr/bin/env python3
"""RoI backward on a 1x1 feature map (every sample lands on the one cell): prints the device
result beside float32 sums taken in several orders."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from objectdetection_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for R in (1, 2, 3):
    fm = rng.standard_normal((1, 1, 1, 4)).astype(np.float32)
    prop = rng.uniform(0, 1, (1, R, 4)).astype(np.float32)
    out, win = ops.roi_crop_pool_fwd(torch.from_numpy(fm).to(dev), torch.from_numpy(prop).to(dev),
                                     (14, 14), 2, want_winner=True)
    dY = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    got = ops.roi_crop_pool_bwd(torch.from_numpy(dY).to(dev), win, torch.from_numpy(prop).to(dev),
                                fm.shape, (14, 14), 2).cpu().numpy().ravel()
    seq = np.zeros(4, np.float32)
    per = np.zeros(4, np.float32)
    for r in range(R):
        a = np.zeros(4, np.float32)
        for p in range(7):
            for q in range(7):
                seq = (seq + dY[0, r, p, q]).astype(np.float32)
                a = (a + dY[0, r, p, q]).astype(np.float32)
        per = (per + a).astype(np.float32)
    print("R", R, "winner", np.unique(win.cpu().numpy()))
    print(" gpu", [float(v).hex() for v in got])
    print(" seq", [float(v).hex() for v in seq])
    print(" per", [float(v).hex() for v in per])
