#!/usr/bin/env python3
"""Profiling target: the bench workload's RoI kernel alone (run under rocprofv3).
Runs one full pass to obtain realistic proposals, then REPS launches of
frcnn_roi_crop_pool_fwd on them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from objectdetection_amd import ops  # noqa: E402
from objectdetection_amd.pipeline import RegionProposalPath  # noqa: E402

B = int(os.environ.get("ROI_B", "64"))
REPS = int(os.environ.get("ROI_REPS", "5"))
WINNER = os.environ.get("ROI_WINNER", "0") == "1"
BAND = os.environ.get("ROI_BAND", "1") == "1"       # frcnn_roi_crop_pool_fwd_ws (band kernel)
PLANAR = os.environ.get("ROI_PLANAR", "0")          # 1: [B,C,H,W] map + workspace; 2: no workspace
dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 512, dev, 1000)
path = RegionProposalPath(cfg, B, 512, dev, train=False)
path.run(fmap, score, coord)
torch.cuda.synchronize()
ws = torch.empty(ops.roi_fwd_workspace_bytes(B, cfg.post_n), dtype=torch.uint8, device=dev) if BAND else None
if PLANAR != "0":
    fmap = fmap.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
    ws = torch.empty(ops.roi_fwd_workspace_bytes(B, cfg.post_n, (cfg.H, cfg.W, 512)),
                     dtype=torch.uint8, device=dev) if PLANAR == "1" else None
ROI_R = int(os.environ.get("ROI_R", "0"))           # > 0: only the first ROI_R proposals per image
proposed, pooled = path.proposed, path.pooled
if ROI_R > 0:
    proposed = proposed[:, :ROI_R].contiguous()
    pooled = torch.empty((B, ROI_R) + tuple(pooled.shape[2:]), dtype=pooled.dtype, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(REPS):
    ev0.record()
    ops.roi_crop_pool_fwd(fmap, proposed, cfg.crop, cfg.pool, want_winner=WINNER,
                          out=pooled, workspace=ws)
    ev1.record()
    torch.cuda.synchronize()
    ts.append(ev0.elapsed_time(ev1))
alg = bench.algorithmic_bytes_roi(cfg, 512, B)
if ROI_R > 0:
    alg = alg * ROI_R // cfg.post_n
best = min(ts)
print(f"roi fwd B={B} band={int(BAND)} planar={PLANAR}: min {best:.3f} ms  mean {sum(ts)/len(ts):.3f} ms  "
      f"{alg/best/1e6:.1f} GB/s algorithmic ({alg/best/1e6/8000*100:.1f}% of 8 TB/s)")
