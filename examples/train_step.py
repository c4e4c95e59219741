#!/usr/bin/env python3
"""The reference's joint training loop (Faster_RCNN/model_train.py:609-675) on the HIP path.

  python examples/train_step.py --steps 3 --batch 2
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_step.py

Synthetic images and boxes (no dataset, no pretrained weights: there is no network here).  One
process per GPU; gradients are all-reduced over RCCL in buckets during the backward."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from objectdetection_amd.config import PathConfig                     # noqa: E402
from objectdetection_amd.module import random_augmentation            # noqa: E402
from objectdetection_amd.train import RPNModel, FRCNModel, HipPathOps, TrainStep   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU (model_train.py:610 uses 2)")
    ap.add_argument("--size", type=int, nargs=2, default=(500, 500), help="image height width")
    ap.add_argument("--width", type=float, default=1.0, help="channel multiplier of the backbone")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    H, W = args.size[0] // 16, args.size[1] // 16                     # 31 x 31 for 500 x 500
    n_anchor = H * W * 9
    cfg = PathConfig(H=H, W=W, base_w=float(args.size[1]), base_h=float(args.size[0]),
                     pre_k=min(6000, n_anchor), post_n=min(1500, n_anchor // 4))
    torch.manual_seed(0)                                              # same weights on every rank
    rpn = RPNModel(cfg.A, args.width).to(dev)
    frcn = FRCNModel(rpn.channels, cfg.crop[0] // cfg.pool, 4096, cfg.num_classes).to(dev)
    step = TrainStep(rpn, frcn, cfg, HipPathOps(cfg, args.batch, rpn.channels, dev), dist, rank, world)
    g = torch.Generator(device=dev)
    g.manual_seed(100 + rank)
    for it in range(args.steps):
        images = torch.rand((args.batch, H * 16 + 8, W * 16 + 8, 3), generator=g, device=dev)
        # ragged boxes per image -> random flip of image and boxes, padding to G rows
        # (module/data.py:36-52)
        bboxes, labels = [], []
        for _ in range(args.batch):
            n = int(torch.randint(1, 8, (1,)))
            c = torch.rand((n, 2)) * 0.6 + 0.2
            half = torch.rand((n, 2)) * 0.15 + 0.05
            bboxes.append(torch.cat([c - half, c + half], 1))
            labels.append(torch.randint(0, 20, (n,)))
        batch = random_augmentation(images, bboxes, labels, cfg, seed=it,
                                    image_base=rank * args.batch)
        images, gt, label = batch["image"].permute(0, 3, 1, 2), batch["bbox"], batch["label"]
        losses = step.step(images, gt, label, seed=it)
        if rank == 0:
            l = losses.tolist()
            print(f"step {it}: rpn 2*cls {l[0]:.4f} reg {l[1]:.4f} | head 2*cls {l[2]:.4f} reg {l[3]:.4f}")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
