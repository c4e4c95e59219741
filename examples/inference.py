#!/usr/bin/env python3
"""The reference's per-epoch detection check (Faster_RCNN/model_train.py:740-760) on the HIP
path: RPN forward -> process_fmap (top-k, NMS, crop + pool) -> head -> class argmax,
generate_coord, final NMS, score cut.  Synthetic image, random weights.

  python examples/inference.py [--size 600 1000]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from objectdetection_amd.config import PathConfig                     # noqa: E402
from objectdetection_amd.module import make_anchor, process_fmap, detect   # noqa: E402
from objectdetection_amd.train import RPNModel, FRCNModel             # noqa: E402


def run(size=(500, 500), batch=1, width=1.0, hidden=4096, dev="cuda:0", score_threshold=0.0):
    dev = torch.device(dev)
    H, W = size[0] // 16, size[1] // 16
    n_anchor = H * W * 9
    cfg = PathConfig(H=H, W=W, base_w=float(size[1]), base_h=float(size[0]),
                     pre_k=min(6000, n_anchor), post_n=min(1500, n_anchor // 4))
    torch.manual_seed(0)
    rpn = RPNModel(cfg.A, width).to(dev).eval()
    frcn = FRCNModel(rpn.channels, cfg.crop[0] // cfg.pool, hidden, cfg.num_classes).to(dev).eval()
    anchor_box = make_anchor(cfg, dev)                                # model_train.py:601
    image = torch.rand((batch, 3, H * 16 + 8, W * 16 + 8), device=dev)
    with torch.no_grad():
        fmap, pred_reg, pred_obj = rpn(image)                         # :741
        crop_fmap, proposed = process_fmap(fmap, pred_obj, pred_reg, anchor_box, cfg,
                                           fuse_pool=True)            # :743 (+ head layer 1, model.py:30)
        pred_reg_valid, pred_obj_valid = frcn(crop_fmap)              # :745
        dets = detect(proposed, pred_reg_valid, pred_obj_valid, cfg,
                      score_threshold=score_threshold)               # :747-759
    return cfg, proposed, dets


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=(500, 500))
    a = ap.parse_args()
    cfg, proposed, dets = run(tuple(a.size))
    n = int(dets["count"][0])
    print(f"{proposed.shape[1]} proposals -> {n} detections")
    for i in range(n):
        print("  class", int(dets["classes"][0, i]), "score %.3f" % float(dets["scores"][0, i]),
              [round(float(v), 3) for v in dets["boxes"][0, i]])
