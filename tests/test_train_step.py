"""The joint training step (objectdetection_amd/train.py; reference loop
Faster_RCNN/model_train.py:623-664) and its data-parallel machinery.

CPU part: two gloo ranks with one image each must produce the gradients of one rank on both
images (global loss normalisation + bucketed sum all-reduce), with the hot path supplied by the
CPU oracle and torch formulas of the losses.  GPU part: the same step on the HIP kernels gives
the oracle-backed step's losses and gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import frcnn_oracle as O
from objectdetection_amd.config import PathConfig
from objectdetection_amd.train import RPNModel, FRCNModel, TrainStep, GradBuckets

CFG = PathConfig(H=6, W=7, scales=(24, 40, 64), ratios=(0.5, 1, 2), base_w=120.0, base_h=104.0,
                 G=5, rpn_batch=32, rpn_max_pos=16, pre_k=120, post_n=24, head_batch=12,
                 head_max_pos=4, num_classes=21)
EPS = 1e-7


class _CropPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap, proposed, crop, pool):
        prop = proposed.cpu().numpy()
        out, win = O.roi_crop_pool(fmap.detach().cpu().numpy(), prop, crop, pool)
        ctx.meta = (win, prop, tuple(fmap.shape), crop, pool)
        return torch.from_numpy(out).to(fmap.device)

    @staticmethod
    def backward(ctx, g):
        win, prop, shape, crop, pool = ctx.meta
        return torch.from_numpy(O.roi_crop_pool_bwd(g.contiguous().cpu().numpy(), win, shape, prop,
                                                    crop, pool)).to(g.device), None, None, None


def _huber(t, p):
    e = t - p
    return torch.where(e.abs() < 1, e * e / 2, e.abs() - 0.5)


class OraclePathOps:
    """The hot path from the CPU oracle (tests only), losses as torch formulas."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.anchor = O.make_anchor(cfg.H, cfg.W, cfg.scales, cfg.ratios, cfg.base_w, cfg.base_h)

    def targets(self, fmap, pred_reg, pred_obj, gt, label, seed, image_base):
        cfg = self.cfg
        g, l = gt.cpu().numpy(), label.cpu().numpy()
        B = g.shape[0]
        rl, rp, rr = [], [], []
        for b in range(B):
            ref = O.patch_batch(dict(image=None, bbox=g[b], label=l[b]), self.anchor, seed=seed,
                                image_index=image_base + b, batch=cfg.rpn_batch,
                                max_pos=cfg.rpn_max_pos)
            rl.append(ref[1].reshape(-1)); rp.append(ref[5]); rr.append(ref[6])
        pc = O.proposal_creator(pred_obj.cpu().numpy(), pred_reg.cpu().numpy(), self.anchor,
                                cfg.pre_k, cfg.post_n, cfg.nms_thr)
        lab = O.label_proposals(pc["proposed"], g, l, cfg.fg_thr, cfg.iou_floor, cfg.bg_label,
                                cfg.num_classes)
        ti = O.sample_proposals(lab["p_iou"], seed=seed, image_base=image_base,
                                batch=cfg.head_batch, max_pos=cfg.head_max_pos,
                                bg_label=cfg.bg_label)
        T = lambda a: torch.from_numpy(a).to(fmap.device)
        return dict(rpn_label=T(np.stack(rl)), rpn_pos=T(np.stack(rp)), rpn_reg=T(np.stack(rr)),
                    proposed=T(pc["proposed"]), tindex=T(ti),
                    gt_label2=T(np.take_along_axis(lab["gt_label"], ti[..., None], 1)),
                    gt_mask2=T(np.take_along_axis(lab["gt_mask"], ti, 1)),
                    gt_coord2=T(np.take_along_axis(lab["gt_coord"], ti[..., None], 1)))

    def roi_pool(self, fmap, proposed):
        return _CropPool.apply(fmap, proposed, tuple(self.cfg.crop), self.cfg.pool)

    def rpn_loss(self, pred_obj, pred_reg, t):
        B, H, W, A = pred_obj.shape
        S = t["rpn_label"].shape[1]
        bi = torch.arange(B, device=pred_obj.device).repeat_interleave(S)
        y, x, a = t["rpn_pos"].reshape(-1, 3).T
        p = pred_obj[bi, y, x, a]
        lab = t["rpn_label"].reshape(-1)
        pc = p.clamp(EPS, 1 - EPS)
        bce = -(lab * torch.log(pc + EPS) + (1 - lab) * torch.log(1 - pc + EPS))
        m = lab == 1
        hub = _huber(t["rpn_reg"].reshape(-1, 4)[m], pred_reg.reshape(B, H, W, A, 4)[bi, y, x, a][m])
        P = m.sum()
        reg = hub.sum() / (4 * P) if int(P) else hub.sum() * 0
        return 2 * bce.mean(), reg, P.float()

    def head_loss(self, cls_prob, reg, t):
        B, R, K = cls_prob.shape
        S = t["tindex"].shape[1]
        bi = torch.arange(B, device=cls_prob.device).repeat_interleave(S)
        r = t["tindex"].reshape(-1)
        lab = t["gt_label2"].reshape(B * S, K)
        p = cls_prob[bi, r]
        q = (p / p.sum(-1, keepdim=True)).clamp(EPS, 1 - EPS)
        cce = -(lab * torch.log(q)).sum(-1)
        m = t["gt_mask2"].reshape(-1) != 0
        c = lab.argmax(-1)
        gtc = t["gt_coord2"].reshape(-1, 4).clamp(-100, 100)
        hub = _huber(gtc[m], reg.reshape(B, R, K, 4)[bi, r, c][m])
        P = m.sum()
        lreg = hub.sum() / (4 * P) if int(P) else hub.sum() * 0
        return 2 * cce.mean(), lreg, P.float()


def _models(seed=3):
    torch.manual_seed(seed)
    rpn = RPNModel(CFG.A, width=1 / 16)
    frcn = FRCNModel(rpn.channels, CFG.crop[0] // CFG.pool, 64, CFG.num_classes)
    return rpn, frcn


def _data(n):
    rng = np.random.default_rng(5)
    images = torch.from_numpy(rng.random((n, 3, CFG.H * 16 + 8, CFG.W * 16 + 8), dtype=np.float32))
    d = O.synth_inputs(seed=9, B=n, H=CFG.H, W=CFG.W, A=CFG.A, C=4, G=CFG.G, gt_side=(0.2, 0.6),
                       with_fmap=False)
    return images, torch.from_numpy(d["gt"]), torch.from_numpy(d["label"])


def _grads(step):
    return torch.cat([b["flat"] for b in step.buckets.buckets]).clone()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        rpn, frcn = _models()
        images, gt, label = _data(world)
        step = TrainStep(rpn, frcn, CFG, OraclePathOps(CFG), dist, rank, world,
                         bucket_bytes=64 << 10)
        assert len(step.buckets.buckets) > 2                 # several buckets in flight
        losses = step.forward_backward(images[rank:rank + 1], gt[rank:rank + 1],
                                       label[rank:rank + 1], seed=4)
        if rank == 0:
            torch.save(dict(grads=_grads(step), losses=losses), os.path.join(out_dir, "dp.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_give_the_one_rank_gradients(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = torch.load(tmp_path / "dp.pt", weights_only=True)
    rpn, frcn = _models()
    images, gt, label = _data(world)
    step = TrainStep(rpn, frcn, CFG, OraclePathOps(CFG), None, 0, 1, bucket_bytes=64 << 10)
    losses = step.forward_backward(images, gt, label, seed=4)
    ref = _grads(step)
    assert ref.abs().max() > 0
    np.testing.assert_allclose(got["losses"].numpy(), losses.numpy(), rtol=1e-5)
    # float summation order differs (two partial sums added by the all-reduce)
    np.testing.assert_allclose(got["grads"].numpy(), ref.numpy(), rtol=2e-4,
                               atol=1e-6 * float(ref.abs().max()))


def test_grad_buckets_are_views_and_zeroing_keeps_them():
    lin = torch.nn.Linear(8, 4)
    gb = GradBuckets(lin.parameters(), None, bucket_bytes=1 << 20)
    lin(torch.ones(2, 8)).sum().backward()
    flat = gb.buckets[0]["flat"]
    assert flat.abs().sum() > 0 and lin.weight.grad.data_ptr() >= flat.data_ptr()
    gb.zero()
    assert not lin.weight.grad.any() and not lin.bias.grad.any()
    lin(torch.ones(2, 8)).sum().backward()
    assert lin.bias.grad.tolist() == [2.0] * 4


class _RecordingDist:
    """Stands in for torch.distributed in one process: records the buffers it is asked to sum."""

    class ReduceOp:
        SUM = "sum"

    class _Done:
        def wait(self):
            pass

    def __init__(self):
        self.calls = []

    def all_reduce(self, t, op=None, async_op=False):
        self.calls.append(t.data_ptr())
        return self._Done()


def test_a_bucket_with_an_unused_parameter_is_still_reduced():
    """A parameter that receives no gradient leaves its bucket short of its count; finish() must
    reduce that bucket all the same (in bucket order), or the ranks would step on different
    gradients without any error."""
    a, b = torch.nn.Linear(8, 4), torch.nn.Linear(8, 4)          # b takes no part in the loss
    d = _RecordingDist()
    gb = GradBuckets(list(a.parameters()) + list(b.parameters()), d, bucket_bytes=1)
    assert len(gb.buckets) == 4
    gb.zero()
    a(torch.ones(2, 8)).sum().backward()
    assert len(d.calls) == 2                                    # a's two buckets, from the hooks
    gb.finish()
    assert sorted(d.calls) == sorted(bk["flat"].data_ptr() for bk in gb.buckets)
    assert d.calls[2:] == [bk["flat"].data_ptr() for bk in gb.buckets if bk["ready"] != bk["n"]]
    d.calls.clear()
    gb.zero()
    (a(torch.ones(2, 8)).sum() + b(torch.ones(2, 8)).sum()).backward()
    gb.finish()
    assert len(d.calls) == 4                                    # nothing is reduced twice


def _worker_hip(rank, world, port, out_dir, per_rank):
    """One rank of the HIP-backed training step; the ranks share cuda:0 and talk over gloo."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from objectdetection_amd.train import HipPathOps
        dev = torch.device("cuda:0")
        rpn, frcn = _models()
        rpn, frcn = rpn.to(dev), frcn.to(dev)
        images, gt, label = _data(world * per_rank)
        sl = slice(rank * per_rank, (rank + 1) * per_rank)
        step = TrainStep(rpn, frcn, CFG, HipPathOps(CFG, per_rank, rpn.channels, dev), dist, rank,
                         world, bucket_bytes=64 << 10)
        losses = step.forward_backward(images[sl].to(dev), gt[sl].to(dev), label[sl].to(dev), seed=4)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save(dict(grads=_grads(step).cpu(), losses=losses.cpu()),
                       os.path.join(out_dir, "dp_hip.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_hip_step_two_ranks_give_the_one_rank_gradients(dev, tmp_path):
    """The HIP-backed training step data parallel: two ranks (sharing the one GPU, gloo) with two
    images each against one rank on the four images -- same sampled targets (counter keys follow
    the global image index), global loss normalisation, bucketed gradient all-reduce."""
    from objectdetection_amd.train import HipPathOps
    world, per_rank = 2, 2
    mp.spawn(_worker_hip, args=(world, _free_port(), str(tmp_path), per_rank), nprocs=world,
             join=True)
    got = torch.load(tmp_path / "dp_hip.pt", weights_only=True)
    rpn, frcn = _models()
    rpn, frcn = rpn.to(dev), frcn.to(dev)
    images, gt, label = _data(world * per_rank)
    step = TrainStep(rpn, frcn, CFG, HipPathOps(CFG, world * per_rank, rpn.channels, dev), None, 0,
                     1, bucket_bytes=64 << 10)
    losses = step.forward_backward(images.to(dev), gt.to(dev), label.to(dev), seed=4).cpu()
    ref = _grads(step).cpu()
    assert ref.abs().max() > 0
    np.testing.assert_allclose(got["losses"].numpy(), losses.numpy(), rtol=1e-5)
    np.testing.assert_allclose(got["grads"].numpy(), ref.numpy(), rtol=2e-4,
                               atol=1e-6 * float(ref.abs().max()))


@pytest.mark.gpu
def test_hip_step_matches_the_oracle_backed_step(dev):
    """One step on the HIP path (RegionProposalPath, fused crop + pool with its backward, loss
    kernels) against the same step with the oracle supplying the hot path: same sampled
    targets, losses within 1e-5, gradients within float tolerance."""
    from objectdetection_amd.train import HipPathOps
    images, gt, label = _data(2)
    rpn, frcn = _models()
    rpn, frcn = rpn.to(dev), frcn.to(dev)           # same networks, same device: same outputs
    ref_step = TrainStep(rpn, frcn, CFG, OraclePathOps(CFG), None, 0, 1)
    ref_losses = ref_step.forward_backward(images.to(dev), gt.to(dev), label.to(dev), seed=4).cpu()
    ref = _grads(ref_step).cpu()
    rpn2, frcn2 = _models()
    rpn2, frcn2 = rpn2.to(dev), frcn2.to(dev)
    step = TrainStep(rpn2, frcn2, CFG, HipPathOps(CFG, 2, rpn2.channels, dev), None, 0, 1)
    losses = step.forward_backward(images.to(dev), gt.to(dev), label.to(dev), seed=4)
    np.testing.assert_allclose(losses.cpu().numpy(), ref_losses.numpy(), rtol=2e-4)
    got = _grads(step).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=5e-3,
                               atol=2e-5 * float(ref.abs().max()))
    step.opt.step()                                          # and the optimizer runs on the views


@pytest.mark.gpu
def test_examples_run(dev):
    """The two caller loops shipped as scripts: the inference chain (model_train.py:740-760) and
    the training loop (model_train.py:609-675), at small sizes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    import inference
    cfg, proposed, dets = inference.run((500, 500), batch=2, width=1 / 8, hidden=128)
    assert proposed.shape == (2, cfg.post_n, 4)
    n = dets["count"].cpu().numpy()
    assert (n >= 0).all() and (n <= 30).all()
    b = dets["boxes"].cpu().numpy()
    assert ((b >= 0) & (b <= 1)).all()
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "train_step.py"), "--steps",
                          "2", "--batch", "2", "--size", "500", "500", "--width", "0.125"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("step ") == 2 and "nan" not in out.stdout
