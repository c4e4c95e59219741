#!/usr/bin/env python3
"""Which small kernel costs the band-resident RoI kernel how much: the RoI kernel runs on one
stream while ONE other stage is launched over and over on a second stream."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from objectdetection_amd import ops
from objectdetection_amd.pipeline import RegionProposalPath

B = 64
dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 512, dev, 1000)
path = RegionProposalPath(cfg, B, 512, dev, train=True)
path.run(fmap, score, coord, gt, label)
torch.cuda.synchronize()
ws = path.roi_ws
a = path.anchor.reshape(-1, 4)
N, K, R = cfg.N, cfg.pre_k, cfg.post_n
stages = {
    "none": None,
    "iou_assign": lambda: ops.iou_assign(a, gt, cfg.rpn_pos_thr, cfg.rpn_neg_thr),
    "rpn_sample": lambda: ops.rpn_sample(path.anchor, gt, path.best_gt, path.flags, seed=1,
                                         batch=cfg.rpn_batch, max_pos=cfg.rpn_max_pos),
    "topk_sort": lambda: ops.topk(score.reshape(B, -1), K, a, coord, "sort", workspace=path.topk_ws),
    "nms": lambda: ops.nms(path.cand, cfg.nms_thr, R, workspace=path.nms_ws),
    "label+sample proposals": lambda: (ops.label_proposals(path.proposed, gt, label, cfg.fg_thr, cfg.iou_floor,
                                                           cfg.bg_label, cfg.num_classes),
                                       ops.sample_proposals(path.p_label, cfg.bg_label, seed=1,
                                                            batch=cfg.head_batch, max_pos=cfg.head_max_pos)),
}
side = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)
for name, fn in stages.items():
    ts = []
    for rep in range(14):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        go = torch.cuda.Event()
        go.record(main)
        e0.record(main)
        ops.roi_crop_pool_fwd(fmap, path.proposed, cfg.crop, cfg.pool, out=path.pooled, workspace=ws)
        e1.record(main)
        n = 0
        if fn is not None:
            with torch.cuda.stream(side):
                side.wait_event(go)
                # about 3 ms worth of this stage back to back
                for _ in range(int(os.environ.get("REPEAT", "4"))):
                    fn(); n += 1
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = ts[4:]
    print(f"{name:26s} RoI kernel {sum(ts)/len(ts):.3f} ms (min {min(ts):.3f}) beside {n} launches of the stage", flush=True)
