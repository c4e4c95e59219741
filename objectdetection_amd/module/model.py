"""process_fmap and the head's RoI max-pool -- counterparts of
Faster_RCNN/module/model.py:43-57 and :30."""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from ..config import PathConfig, REFERENCE


def propose(score: torch.Tensor, coord: torch.Tensor, anchor_box: torch.Tensor,
            cfg: PathConfig = REFERENCE, workspace: Optional[torch.Tensor] = None):
    """Proposal creator (module/model.py:44-52): top-k -> decode -> NMS.
    Returns dict(proposed [B,post_n,4], keep_pos, counts, top_scores, top_idx, cand)."""
    B = score.shape[0]
    a = anchor_box.reshape(-1, 4)
    N = a.shape[0]
    vals, idx, cand = ops.topk(score.reshape(B, N), cfg.pre_k, a, coord.reshape(B, N, 4))
    proposed, keep, count = ops.nms(cand, cfg.nms_thr, cfg.post_n, workspace)
    return dict(proposed=proposed, keep_pos=keep, counts=count, top_scores=vals, top_idx=idx,
                cand=cand)


def _is_planar(fmap: torch.Tensor) -> bool:
    """True for the permuted [B,H,W,C] view of a contiguous [B,C,H,W] tensor."""
    B, H, W, C = fmap.shape
    return C > 1 and tuple(fmap.stride()) == (C * H * W, W, 1, H * W)


def _path_for(fmap, score, anchor_box, cfg, train, fuse_pool, inputs_ready=None):
    """The kept buffer set for a call of these dimensions, its anchors bound to `anchor_box`."""
    from ..pipeline import cached_path
    B, H, W, C = fmap.shape
    if (H, W) != (cfg.H, cfg.W) or tuple(anchor_box.shape) != (cfg.H, cfg.W, cfg.A, 4):
        raise ValueError(f"fmap {tuple(fmap.shape)} / anchor_box {tuple(anchor_box.shape)} do not "
                         f"match cfg ({cfg.H}x{cfg.W}x{cfg.A}): pass the PathConfig of this grid")
    path = cached_path(cfg, B, C, fmap.device, train=train, fuse_pool=fuse_pool,
                       fmap_planar=_is_planar(fmap))
    path.anchor = anchor_box.to(torch.float32).contiguous()
    return path


def process_fmap(fmap: torch.Tensor, score: torch.Tensor, coord: torch.Tensor,
                 anchor_box: torch.Tensor, cfg: PathConfig = REFERENCE, fuse_pool: bool = False,
                 workspace: Optional[torch.Tensor] = None, cache: bool = True,
                 inputs_ready=None):
    """Counterpart of module/model.py:43-57.

    fmap [B,H,W,C] (NHWC, as the reference's block5_conv3 output), score
    [B,H,W,A], coord [B,H,W,4A], anchor_box [H,W,A,4].  Returns (crop_fmap,
    proposed) with crop_fmap [B,post_n,14,14,C] like the reference, or -- with
    fuse_pool=True -- the pooled [B,post_n,7,7,C] tensor that the head's first
    layer (module/model.py:30) would make of it.

    cache=True (default): the call runs on a buffer set kept per (cfg, B, C, device) --
    radix-sort top-k, NMS and band-resident RoI kernel with their workspaces, nothing allocated
    per call.  The returned tensors belong to that set: `crop_fmap` is overwritten by the next
    call with the same dimensions, `proposed` by the one after it; clone what must live longer
    (pipeline.clear_path_cache() frees the sets).  `inputs_ready`: an event after which the inputs
    may be read -- with it the proposals of this call are computed beside the RoI kernel of the
    call before.  cache=False allocates every output per call, as TensorFlow does.
    """
    if not cache or workspace is not None:
        pr = propose(score, coord, anchor_box, cfg, workspace)
        pool = cfg.pool if fuse_pool else 1
        crop_fmap, _ = ops.roi_crop_pool_fwd(fmap, pr["proposed"], cfg.crop, pool)
        return crop_fmap, pr["proposed"]
    path = _path_for(fmap, score, anchor_box, cfg, False, fuse_pool)
    path.run(fmap, score.contiguous(), coord.contiguous(), inputs_ready=inputs_ready)
    return path.pooled, path.proposed


def roi_pool(fmap: torch.Tensor, proposed: torch.Tensor, cfg: PathConfig = REFERENCE,
             want_winner: bool = False):
    """crop_and_resize (module/model.py:54) + MaxPool2D (module/model.py:30) in one kernel."""
    return ops.roi_crop_pool_fwd(fmap, proposed, cfg.crop, cfg.pool, want_winner)


class _RoICropPool(torch.autograd.Function):
    """Differentiable fused crop + max-pool.  The reference stops the gradient here
    (module/model.py:71); this is the exact adjoint w.r.t. the feature map, computed by
    frcnn_roi_crop_pool_bwd (no atomics, fixed summation order).  Boxes get no gradient."""

    @staticmethod
    def forward(ctx, fmap, proposed, crop, pool):
        out, win = ops.roi_crop_pool_fwd(fmap, proposed, crop, pool, want_winner=pool > 1)
        ctx.save_for_backward(proposed, win if win is not None else proposed.new_empty(0))
        ctx.meta = (tuple(fmap.shape), crop, pool)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        proposed, win = ctx.saved_tensors
        shape, crop, pool = ctx.meta
        g = ops.roi_crop_pool_bwd(grad_out.contiguous(), win if pool > 1 else None, proposed, shape,
                                  crop, pool)
        return g, None, None, None


def roi_pool_autograd(fmap: torch.Tensor, proposed: torch.Tensor, cfg: PathConfig = REFERENCE):
    """roi_pool with a backward pass to the feature map (for end-to-end training)."""
    return _RoICropPool.apply(fmap, proposed, tuple(cfg.crop), int(cfg.pool))


def fc1(pooled: torch.Tensor, kernel: torch.Tensor, bias: Optional[torch.Tensor] = None,
        relu: bool = True) -> torch.Tensor:
    """The first two layers of the head behind the RoI pool (module/model.py:31-32):
    TimeDistributed(Flatten) + TimeDistributed(Dense(4096, activation='relu')).

    pooled [B,R,7,7,C] as process_fmap(fuse_pool=True) / roi_pool return it: a RoI's block is
    its (h,w,c)-flattened row already, so Flatten is a view.  kernel [7*7*C, units] and bias
    [units] in the Keras layout.  Returns [B,R,units] float32, computed by frcnn_fc1_fwd."""
    return ops.fc1_fwd(pooled, kernel, bias, relu)
