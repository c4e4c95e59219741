#!/usr/bin/env python3
"""Times the hot path on each benchmark configuration (HIP events, one GPU) and writes one
JSON object: B=1 latency of the 600x1000 workload with its stages, the ship configuration at
batch 16, the 100,000-candidate NMS and the G=200 anchor assignment of the dense
configuration, and a training step's RoI part (forward with winner + backward) at batch 8.

  python tools/config_table.py [out.json]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from objectdetection_amd import ops  # noqa: E402
from objectdetection_amd.config import PathConfig  # noqa: E402
from objectdetection_amd.pipeline import RegionProposalPath  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=10, warm=2):
    """Mean and minimum HIP-event time of fn() in ms."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return dict(mean_ms=float(np.mean(ts)), min_ms=float(np.min(ts)))


def path_times(cfg, B, C, train, seed):
    fmap, score, coord, gt, label = bench.make_inputs(cfg, B, C, dev, seed)
    path = RegionProposalPath(cfg, B, C, dev, train=train, depth=1)
    run = (lambda: path.run(fmap, score, coord, gt, label, seed=1)) if train else \
          (lambda: path.run(fmap, score, coord))
    out = dict(whole_path=timed(run))
    ws = path.topk_ws
    out["topk_sort"] = timed(lambda: ops.topk(score.reshape(B, -1), cfg.pre_k, anchor_flat=path.anchor.reshape(-1, 4),
                                              deltas=coord.reshape(B, -1, 4), method="sort",
                                              workspace=ws))
    out["nms"] = timed(lambda: ops.nms(path.cand, cfg.nms_thr, cfg.post_n, workspace=path.nms_ws))
    out["roi_fwd"] = timed(lambda: ops.roi_crop_pool_fwd(fmap, path.proposed, cfg.crop, cfg.pool,
                                                         out=path.pooled))
    if train:
        out["iou_assign"] = timed(lambda: ops.iou_assign(path.anchor.reshape(-1, 4), gt))
    graph = path.capture(fmap, score, coord, gt, label, seed=1) if train else \
        path.capture(fmap, score, coord)
    out["whole_path_graph_replay"] = timed(graph.replay)
    out["images_per_s"] = B / (out["whole_path"]["mean_ms"] * 1e-3)
    out["mean_kept_proposals"] = float(path.keep_count.float().mean().item())
    return out


def main():
    res = {}
    cfg1 = bench.workload_cfg()
    res["config1_600x1000_B1_train"] = path_times(cfg1, 1, 512, True, 1000)
    res["config1_600x1000_B1_inference"] = path_times(cfg1, 1, 512, False, 1000)

    ship = PathConfig(H=48, W=48, scales=(32, 64, 128, 256), base_w=500.0, base_h=500.0, G=50,
                      pre_k=6000, post_n=1500)
    res["config2_ship_768x768_B16_train"] = path_times(ship, 16, 512, True, 2000)

    # dense stress: NMS over 100,000 candidates -> 2,000; anchors x 200 GT
    rng = np.random.default_rng(303)
    K = 100_000
    cy, cx = rng.uniform(0, 1, K), rng.uniform(0, 1, K)
    h = np.exp(rng.uniform(np.log(0.01), np.log(0.3), K))
    w = np.exp(rng.uniform(np.log(0.01), np.log(0.3), K))
    boxes = np.clip(np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], 1), 0, 1)
    boxes = torch.from_numpy(boxes.astype(np.float32)[None]).to(dev)
    nws = torch.empty((ops.nms_workspace_bytes(1, K, 2000),), dtype=torch.uint8, device=dev)
    res["config3_nms_100k_to_2000"] = timed(lambda: ops.nms(boxes, 0.7, 2000, workspace=nws))
    _, _, cnt = ops.nms(boxes, 0.7, 2000, workspace=nws)
    res["config3_nms_100k_to_2000"]["kept"] = int(cnt[0])
    a3 = ops.anchors(50, 83, (64, 128, 256, 384, 512), (0.5, 1, 2), 1333.0, 800.0, dev)
    g3 = torch.rand((1, 200, 2), device=dev) * 0.7
    gt3 = torch.cat([g3, g3 + 0.05 + torch.rand((1, 200, 2), device=dev) * 0.25], -1).clamp(0, 1)
    res["config3_iou_assign_62250x200"] = timed(lambda: ops.iou_assign(a3.reshape(-1, 4), gt3))

    # training step, RoI part: forward with winner index + backward, batch 8
    B = 8
    fmap, score, coord, gt, label = bench.make_inputs(cfg1, B, 512, dev, 3000)
    path = RegionProposalPath(cfg1, B, 512, dev, train=True, want_winner=True, depth=1)
    path.run(fmap, score, coord, gt, label, seed=1)
    dY = torch.randn_like(path.pooled)
    res["config5_B8_path_forward_with_winner"] = timed(
        lambda: path.run(fmap, score, coord, gt, label, seed=1), reps=5)
    res["config5_B8_roi_backward"] = timed(
        lambda: ops.roi_crop_pool_bwd(dY, path.winner, path.proposed, tuple(fmap.shape),
                                      cfg1.crop, cfg1.pool), reps=3, warm=1)
    # the steps either side of the path, batch 64 of the 600x1000 workload
    B = 64
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    R, NC, S = cfg1.post_n, cfg1.num_classes, cfg1.head_batch
    prop = torch.rand((B, R, 2), generator=g, device=dev) * 0.6
    prop = torch.cat([prop, prop + 0.05 + torch.rand((B, R, 2), generator=g, device=dev) * 0.3], -1).clamp(0, 1)
    cls_prob = torch.softmax(torch.randn((B, R, NC), generator=g, device=dev) * 3, -1)
    reg = torch.randn((B, R, NC * 4), generator=g, device=dev) * 0.3
    res["next_detect_postprocess_B64"] = timed(lambda: ops.detect_postprocess(prop, cls_prob, reg))
    H, W, A, Sr = cfg1.H, cfg1.W, cfg1.A, cfg1.rpn_batch
    pobj = torch.rand((B, H, W, A), generator=g, device=dev)
    preg = torch.randn((B, H, W, A, 4), generator=g, device=dev)
    pos = torch.stack([torch.randint(0, H, (B, Sr), generator=g, device=dev),
                       torch.randint(0, W, (B, Sr), generator=g, device=dev),
                       torch.randint(0, A, (B, Sr), generator=g, device=dev)], -1)
    lab = (torch.rand((B, Sr), generator=g, device=dev) < 0.5).float()
    rgt = torch.randn((B, Sr, 4), generator=g, device=dev)
    res["next_rpn_loss_with_grads_B64"] = timed(lambda: ops.rpn_loss(pobj, preg, pos, lab, rgt))
    tindex = torch.randint(0, R, (B, S), generator=g, device=dev)
    hl = torch.nn.functional.one_hot(torch.randint(0, NC, (B, S), generator=g, device=dev), NC).float()
    gm = (torch.rand((B, S), generator=g, device=dev) < 0.3).int()
    gc = torch.randn((B, S, 4), generator=g, device=dev)
    res["next_head_loss_with_grads_B64"] = timed(
        lambda: ops.head_loss(cls_prob, reg, tindex, hl, gm, gc))
    counts = torch.randint(1, cfg1.G + 1, (B,), generator=g, device=dev)
    offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), counts.cumsum(0)]).int()
    total = int(offs[-1])
    rb = torch.rand((total, 4), generator=g, device=dev)
    rl = torch.randint(0, 20, (total,), generator=g, device=dev)
    res["next_pad_gt_with_flip_B64"] = timed(
        lambda: ops.pad_gt(rb, rl, offs, cfg1.G, use_flip=True, seed=3))
    imgs = torch.rand((B, 600, 1000, 3), generator=g, device=dev)
    t = timed(lambda: ops.flip_images(imgs, seed=3), reps=5)
    t["GBps_read_plus_write"] = 2 * imgs.numel() * 4 / (t["mean_ms"] * 1e-3) / 1e9
    res["next_flip_images_B64_600x1000x3"] = t
    del imgs
    # the head's first dense layer on the pooled tensor of a batch-8 step: [16,000, 25,088] x
    # [25,088, 4,096], float32 matrix cores (peak 157.3 TFLOP/s); torch.matmul beside it
    Mr, Kf, Nf = 8 * cfg1.post_n, 7 * 7 * 512, 4096
    xf = torch.relu(torch.randn((Mr, Kf), generator=g, device=dev))
    wf = torch.randn((Kf, Nf), generator=g, device=dev) * 0.01
    bf = torch.randn((Nf,), generator=g, device=dev)
    of = torch.empty((Mr, Nf), device=dev)
    t = timed(lambda: ops.fc1_fwd(xf, wf, bf, True, out=of), reps=3, warm=1)
    t["tflops"] = 2.0 * Mr * Kf * Nf / (t["mean_ms"] * 1e-3) / 1e12
    t["frac_of_f32_mfma_peak"] = t["tflops"] / 157.3
    res["next_fc1_16000x25088x4096"] = t
    t = timed(lambda: torch.relu_(torch.addmm(bf, xf, wf, out=of)), reps=3, warm=1)
    t["tflops"] = 2.0 * Mr * Kf * Nf / (t["mean_ms"] * 1e-3) / 1e12
    res["next_fc1_torch_addmm_relu"] = t
    txt = json.dumps(res, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
