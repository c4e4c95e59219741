"""Joint RPN + head training step around the HIP hot path, data parallel over RCCL.

Counterpart of the reference's joint loop, Faster_RCNN/model_train.py:623-675:

    fmap, pred_reg, pred_obj = rpn_model(images)                 :626   (construct_rpn, module/model.py:10-27)
    RPN loss on the patch_batch mini-batch                       :627-635
    making_frcnn_input(gt, label, fmap, pred_reg, pred_obj)      :637   (the hot path)
    pred_reg, pred_obj = frcn_model(crop_fmap)                   :646   (construct_frcn, module/model.py:29-41)
    head loss on the sampled proposals                           :647-657
    two optimizer steps                                          :661-664

The networks are stand-ins of the reference's shapes with random weights (VGG16 feature
stack, 3x3 conv + two 1x1 heads, fc4096 x 2 + softmax 21 + reg 84): they stay plain PyTorch-ROCm
modules.  Everything between them is the HIP path (RegionProposalPath, the fused RoI crop +
pool with its backward, the loss-assembly kernels).

Differences from the reference, on purpose:
  * the crop is differentiable (`roi_grad=True`): the reference stops the gradient at the crop
    (module/model.py:71); here the head's loss also reaches the backbone through
    frcnn_roi_crop_pool_bwd;
  * one process per GPU, images sharded per rank.  Gradients are all-reduced (sum) in buckets
    that are launched while the backward is still running; the two losses are normalised by the
    GLOBAL sample and positive counts, so N ranks x B images give the gradients of one rank on
    the N*B images (up to float summation order).
"""
from __future__ import annotations

import json
import time
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .config import PathConfig

VGG16 = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)


class RPNModel(nn.Module):
    """construct_rpn (module/model.py:10-27): VGG16 up to block5_conv3 (stride 16), 3x3 conv,
    sigmoid objectness (A) and box deltas (4A).  `width` scales the channel counts (tests)."""

    def __init__(self, A: int = 9, width: float = 1.0):
        super().__init__()
        layers: List[nn.Module] = []
        c_in = 3
        for v in VGG16:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                c = max(4, int(v * width)) // 4 * 4
                layers += [nn.Conv2d(c_in, c, 3, padding=1), nn.ReLU(inplace=True)]
                c_in = c
        self.features = nn.Sequential(*layers)
        self.channels = c_in
        self.conv = nn.Conv2d(c_in, c_in, 3, padding=1)
        self.cls = nn.Conv2d(c_in, A, 1)
        self.reg = nn.Conv2d(c_in, 4 * A, 1)

    def forward(self, images: torch.Tensor):
        """images [B,3,H,W] -> fmap [B,h,w,C], pred_reg [B,h,w,4A], pred_obj [B,h,w,A]
        (NHWC like the reference's outputs; channels_last memory, so the permutes are views)."""
        x = images.contiguous(memory_format=torch.channels_last)
        f = self.features(x)
        h = torch.relu(self.conv(f))
        obj = torch.sigmoid(self.cls(h))
        reg = self.reg(h)
        nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
        return nhwc(f), nhwc(reg), nhwc(obj)


class FRCNModel(nn.Module):
    """construct_frcn (module/model.py:29-41) without its first layer (the 2x2 max-pool is fused
    into the crop): flatten (h,w,c) -> fc -> fc -> softmax classes, 4 x classes box deltas."""

    def __init__(self, channels: int = 512, pooled: int = 7, hidden: int = 4096,
                 num_classes: int = 21):
        super().__init__()
        self.fc1 = nn.Linear(pooled * pooled * channels, hidden)
        self.fc2 = nn.Linear(hidden, hidden)
        self.cls = nn.Linear(hidden, num_classes)
        self.reg = nn.Linear(hidden, 4 * num_classes)

    def forward(self, pooled: torch.Tensor):
        """pooled [B,R,ph,pw,C] -> pred_reg [B,R,4K], pred_obj [B,R,K]."""
        B, R = pooled.shape[:2]
        x = pooled.reshape(B * R, -1)
        x = torch.relu(self.fc2(torch.relu(self.fc1(x))))
        return self.reg(x).reshape(B, R, -1), torch.softmax(self.cls(x), -1).reshape(B, R, -1)


class _KernelLoss(torch.autograd.Function):
    """(2*cls, reg) whose gradients the loss-assembly kernels have already computed: the first
    scalar depends on the class scores only, the second on the box deltas only."""

    @staticmethod
    def forward(ctx, pred_cls, pred_reg, two_cls, reg, d_cls, d_reg):
        ctx.save_for_backward(d_cls, d_reg)
        return two_cls.clone(), reg.clone()

    @staticmethod
    def backward(ctx, g_cls, g_reg):
        d_cls, d_reg = ctx.saved_tensors
        return g_cls * d_cls, g_reg * d_reg, None, None, None, None


class HipPathOps:
    """The hot path on the HIP kernels: targets, differentiable crop + pool, the two losses."""

    def __init__(self, cfg: PathConfig, batch: int, channels: int, device):
        from .pipeline import RegionProposalPath
        self.cfg = cfg
        self.path = RegionProposalPath(cfg, batch, channels, device, train=True, depth=1,
                                       alloc_pooled=False)

    def targets(self, fmap, pred_reg, pred_obj, gt, label, seed, image_base):
        cfg = self.cfg
        p = self.path.run(fmap, pred_obj.contiguous(), pred_reg.contiguous(), gt, label, seed=seed,
                          image_base=image_base, skip_roi=True)
        ti = p.tindex
        return dict(
            rpn_label=p.rpn["label"], rpn_pos=p.rpn["pos"], rpn_reg=p.rpn["reg"],
            proposed=p.proposed, tindex=ti,
            gt_label2=torch.gather(p.gt_onehot, 1, ti.unsqueeze(-1).expand(-1, -1, cfg.num_classes)),
            gt_mask2=torch.gather(p.gt_mask, 1, ti),
            gt_coord2=torch.gather(p.gt_coord, 1, ti.unsqueeze(-1).expand(-1, -1, 4)))

    def roi_pool(self, fmap, proposed):
        from .module.model import roi_pool_autograd
        return roi_pool_autograd(fmap, proposed, self.cfg)

    def rpn_loss(self, pred_obj, pred_reg, t):
        from . import ops
        # negatives carry -inf / nan targets (module/data.py:147-148); only label-1 rows are read
        out = ops.rpn_loss(pred_obj.detach(), pred_reg.detach(), t["rpn_pos"], t["rpn_label"],
                           t["rpn_reg"])
        two_cls, reg = _KernelLoss.apply(pred_obj, pred_reg, 2.0 * out["loss"][0], out["loss"][1],
                                         out["d_obj"], out["d_reg"].reshape(pred_reg.shape))
        return two_cls, reg, out["partial"][:, 2].sum()

    def head_loss(self, cls_prob, reg, t):
        from . import ops
        out = ops.head_loss(cls_prob.detach(), reg.detach(), t["tindex"], t["gt_label2"],
                            t["gt_mask2"], t["gt_coord2"])
        two_cls, lreg = _KernelLoss.apply(cls_prob, reg, 2.0 * out["loss"][0], out["loss"][1],
                                          out["d_cls"], out["d_reg"].reshape(reg.shape))
        return two_cls, lreg, out["partial"][:, 2].sum()


class GradBuckets:
    """Gradient all-reduce in buckets, launched during the backward.

    Every parameter's .grad is a view into one flat buffer per bucket (buckets are filled in
    reverse parameter order, the order the backward produces gradients in).  A hook counts the
    parameters of a bucket whose gradient is complete; the last one launches that bucket's
    asynchronous SUM all-reduce, so communication of the head's and the last conv blocks'
    gradients overlaps the rest of the backward.  finish() waits for the handles.  On xGMI an
    all-reduce is bound per link, so buckets are large (64 MB by default): a few big messages
    instead of many small ones."""

    def __init__(self, params, dist, bucket_bytes: int = 64 << 20):
        self.dist = dist
        self.params = [p for p in params if p.requires_grad]
        self.buckets: List[Dict] = []
        cur: List[torch.nn.Parameter] = []
        size = 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self._close(cur)
                cur, size = [], 0
        if cur:
            self._close(cur)
        self.handles: List = []

    def _close(self, ps):
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=ps[0].dtype, device=ps[0].device)
        b = dict(flat=flat, n=len(ps), ready=0, launched=False)
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            p.register_post_accumulate_grad_hook(lambda _p, b=b: self._ready(b))
        self.buckets.append(b)

    def _ready(self, b):
        b["ready"] += 1
        if b["ready"] == b["n"]:
            self._launch(b)

    def _launch(self, b):
        b["launched"] = True
        if self.dist is not None:
            self.handles.append(self.dist.all_reduce(b["flat"], op=self.dist.ReduceOp.SUM,
                                                     async_op=True))

    def zero(self):
        for b in self.buckets:
            b["flat"].zero_()
            b["ready"] = 0
            b["launched"] = False
        self.handles = []

    def finish(self):
        """Waits for the all-reduces.  A bucket with a parameter that received no gradient in
        this step (a frozen or unused layer) never fills: it is reduced here, zeros included, in
        bucket order -- every rank has the same parameters, so every rank issues the same
        collectives in the same order."""
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        for h in self.handles:
            h.wait()
        self.handles = []

    @property
    def nbytes(self) -> int:
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)


class TrainStep:
    """One joint step (model_train.py:623-664) on this rank's images."""

    def __init__(self, rpn: RPNModel, frcn: FRCNModel, cfg: PathConfig, path_ops, dist=None,
                 rank: int = 0, world: int = 1, lr: float = 1e-5, roi_grad: bool = True,
                 bucket_bytes: int = 64 << 20):
        self.rpn, self.frcn, self.cfg, self.ops = rpn, frcn, cfg, path_ops
        self.dist = dist if world > 1 else None
        self.rank, self.world, self.roi_grad = rank, world, roi_grad
        params = list(rpn.parameters()) + list(frcn.parameters())
        self.buckets = GradBuckets(params, self.dist, bucket_bytes)
        # model_train.py:590.  The reference applies the RPN and the head gradients with two
        # apply_gradients calls on ONE Adam per iteration (:661-664), so its step counter (bias
        # correction) advances by two per iteration; here one opt.step() advances it by one.
        # Per-parameter moments are the same (each variable is updated once per iteration).
        self.opt = torch.optim.Adam(params, lr=lr)

    def _global(self, t: torch.Tensor) -> torch.Tensor:
        """Sum of a small device tensor over the ranks (stays on the device: no host sync)."""
        if self.dist is None:
            return t
        t = t.clone()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def forward_backward(self, images, gt, label, seed: int = 0):
        B = images.shape[0]
        self.buckets.zero()
        fmap, pred_reg, pred_obj = self.rpn(images)
        t = self.ops.targets(fmap.detach(), pred_reg.detach(), pred_obj.detach(), gt, label, seed,
                             self.rank * B)
        rpn_cls2, rpn_reg, rpn_pos = self.ops.rpn_loss(pred_obj, pred_reg, t)
        pooled = self.ops.roi_pool(fmap if self.roi_grad else fmap.detach(), t["proposed"])
        head_reg_out, head_cls_out = self.frcn(pooled)
        head_cls2, head_reg, head_pos = self.ops.head_loss(head_cls_out, head_reg_out, t)
        # global normalisation: the class terms are means over world*B*S samples, the box terms
        # means over the positives of ALL ranks
        pos = torch.stack([rpn_pos, head_pos]).float()
        gpos = self._global(pos)
        w_reg = torch.where(gpos > 0, pos / gpos.clamp(min=1.0), torch.zeros_like(pos))
        local = (rpn_cls2 + head_cls2) / self.world + w_reg[0] * rpn_reg + w_reg[1] * head_reg
        local.backward()
        self.buckets.finish()
        losses = torch.stack([rpn_cls2 / self.world, w_reg[0] * rpn_reg, head_cls2 / self.world,
                              w_reg[1] * head_reg]).detach()
        return self._global(losses)        # [2*rpn_cls, rpn_reg, 2*head_cls, head_reg], global

    def step(self, images, gt, label, seed: int = 0):
        losses = self.forward_backward(images, gt, label, seed)
        self.opt.step()
        return losses


def bench_train(args, cfg, wl, B, C, device, dist, rank, world, scaling):
    """`bench.py --mode train`: the training step at the configured size, images per GPU = B
    (config 5 of the benchmark list: 8 per GPU).  Reports images/s over all ranks plus the
    times of the step's hot-path pieces (HIP events)."""
    import numpy as np
    if args.batch is None and args.global_batch is None:
        B = 8
    torch.manual_seed(1234)                      # same initial weights on every rank
    rpn = RPNModel(cfg.A).to(device)
    frcn = FRCNModel(rpn.channels, cfg.crop[0] // cfg.pool, 4096, cfg.num_classes).to(device)
    fh, fw = cfg.H, cfg.W
    img_h, img_w = fh * 16 + 8, fw * 16 + 8       # 600 x 1000 for the 37 x 62 map
    ops_ = HipPathOps(cfg, B, rpn.channels, device)
    step = TrainStep(rpn, frcn, cfg, ops_, dist, rank, world)
    g = torch.Generator(device=device)
    g.manual_seed(77 + rank)
    images = torch.rand((B, 3, img_h, img_w), generator=g, device=device)
    import bench
    _, _, _, gt, label = bench.make_inputs(cfg, B, 4, device, seed=2000 + rank, gt_side=wl["gt_side"])

    def sync():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    for w in range(args.warmup):
        step.step(images, gt, label, seed=w)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        losses = step.step(images, gt, label, seed=100 + k)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        from . import ops
        from bench import timed_alone, algorithmic_bytes_roi, HBM_PEAK_GBS
        p = ops_.path
        fmap = torch.relu(torch.randn((B, fh, fw, rpn.channels), device=device))
        out, win = ops.roi_crop_pool_fwd(fmap, p.proposed, cfg.crop, cfg.pool, want_winner=True)
        fwd_ms = timed_alone(lambda: ops.roi_crop_pool_fwd(fmap, p.proposed, cfg.crop, cfg.pool,
                                                           out=out, winner=win), device)
        dY = torch.randn_like(out)
        bwd_ms = timed_alone(lambda: ops.roi_crop_pool_bwd(dY, win, p.proposed, tuple(fmap.shape),
                                                           cfg.crop, cfg.pool), device)
        n_out = out.numel()
        bwd_bytes = n_out * 5 + fmap.numel() * 4
        rec = {
            "metric": "images/sec, joint RPN + head training step (forward, RoI backward, "
                      "gradient all-reduce, Adam)",
            "value": world * B * args.steps / elapsed, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["text"] + f"; {img_h}x{img_w} images through a VGG16-shaped "
                                   "backbone with random weights, fc4096 x 2 head on all "
                                   f"{cfg.post_n} RoIs, differentiable crop",
                       "images_per_gpu_per_step": B, "global_batch": B * world,
                       "gradient_bytes_all_reduced": step.buckets.nbytes,
                       "gradient_buckets": len(step.buckets.buckets)},
            "losses_last_step": [float(v) for v in losses.tolist()],
            "roofline": {"kernel": "roi_bwd (frcnn_roi_crop_pool_bwd)", "bound": "hbm",
                         "achieved": bwd_bytes / (bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s",
                         "frac": bwd_bytes / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": bwd_ms, "algorithmic_bytes_per_launch": bwd_bytes},
            "roi_fwd_with_winner_ms": fwd_ms,
            "roi_fwd_frac": (algorithmic_bytes_roi(cfg, rpn.channels, B) + n_out)
            / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }
        assert all(np.isfinite(rec["losses_last_step"])), rec["losses_last_step"]
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
