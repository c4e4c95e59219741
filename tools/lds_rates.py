#!/usr/bin/env python3
"""Profiling target for the LDS bank-conflict pass: one pass of the whole path (training
stages included) at the bench workload, with both top-k variants, so that every kernel that
uses LDS shows up once per image batch.  Run under
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 tools/lds_rates.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from objectdetection_amd import ops  # noqa: E402
from objectdetection_amd.pipeline import RegionProposalPath  # noqa: E402

B = int(os.environ.get("ROI_B", "64"))
dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 512, dev, 1000)
for topk in ("sort", "lds"):
    path = RegionProposalPath(cfg, B, 512, dev, train=True, topk=topk)
    for i in range(2):
        path.run(fmap, score, coord, gt, label, seed=i)
    torch.cuda.synchronize()
    del path
print("lds pass done")
