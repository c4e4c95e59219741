"""Loss-assembly oracle (next row f2) against an independent torch float64 statement of the
Keras formulas, values and autograd gradients.  Parity unpinned (no TensorFlow here): the
formulas are tf.keras.backend.binary_crossentropy / categorical_crossentropy (TF 2.3, eager
path) and Faster_RCNN/module/loss.py:9-14."""
import numpy as np
import pytest
import torch

from oracle import frcnn_oracle as O

EPS = 1e-7


def _inputs(seed=0, B=3, H=5, W=6, A=4, S=32, R=40, K=7, dup=False):
    rng = np.random.default_rng(seed)
    pred_obj = rng.random((B, H, W, A), dtype=np.float32)
    pred_reg = rng.normal(size=(B, H, W, A, 4)).astype(np.float32) * 2
    pos = np.stack([np.stack(np.unravel_index(rng.permutation(H * W * A)[:S], (H, W, A)), -1)
                    for _ in range(B)]).astype(np.int64)
    for i, v in enumerate([0.0, 1.0, 5e-8]):                         # outside the clip range
        pred_obj[(0,) + tuple(pos[0, i])] = v
    label = (rng.random((B, S)) < 0.4).astype(np.float32)
    reg_gt = rng.normal(size=(B, S, 4)).astype(np.float32) * 2
    cls_prob = rng.random((B, R, K), dtype=np.float32) + 1e-3
    cls_prob /= cls_prob.sum(-1, keepdims=True)
    reg = rng.normal(size=(B, R, K, 4)).astype(np.float32) * 2
    tindex = np.stack([rng.permutation(R)[:S] for _ in range(B)]).astype(np.int64)
    cls_prob[0, tindex[0, 0]] = 0
    cls_prob[0, tindex[0, 0], 1] = 1                                 # zeros: clipped, no gradient
    cls = rng.integers(0, K, (B, S))
    onehot = np.eye(K, dtype=np.float32)[cls]
    gt_mask = (cls != K - 1).astype(np.int32)
    gt_coord = rng.normal(size=(B, S, 4)).astype(np.float32) * 60    # some beyond +-100
    if dup:
        # repeated gather positions, as the RPN mini-batch really holds them (a negative's
        # position is a rank in the valid-anchor list and can equal a positive's anchor id,
        # module/data.py:108-118): one anchor as label 1 and label 0, one anchor twice as
        # label 1, one three times; one proposal row twice with different classes
        for b in range(B):
            pos[b, 5] = pos[b, 2]; label[b, 2], label[b, 5] = 1.0, 0.0
            pos[b, 9] = pos[b, 7]; label[b, 7] = label[b, 9] = 1.0
            pos[b, 20] = pos[b, 11]; pos[b, 30] = pos[b, 11]
            tindex[b, 6] = tindex[b, 3]; cls[b, 3], cls[b, 6] = 0, 1
            tindex[b, 12] = tindex[b, 10]; cls[b, 10] = cls[b, 12] = 2
        onehot = np.eye(K, dtype=np.float32)[cls]
        gt_mask = (cls != K - 1).astype(np.int32)
    return dict(pred_obj=pred_obj, pred_reg=pred_reg, pos=pos, label=label, reg_gt=reg_gt,
                cls_prob=cls_prob, reg=reg, tindex=tindex, onehot=onehot, gt_mask=gt_mask,
                gt_coord=gt_coord)


def _huber_t(t, p):
    e = t - p
    return torch.where(e.abs() < 1, e * e / 2, e.abs() - 0.5)


def test_smooth_l1_known_values():
    v, g = O.loss_bbr(np.float32([0, 0, 0, 0]), np.float32([0.5, -0.5, 2.0, -3.0]))
    np.testing.assert_allclose(v, [0.125, 0.125, 1.5, 2.5])
    np.testing.assert_allclose(g, [0.5, -0.5, 1.0, -1.0])


def test_bce_known_values():
    v, g = O.keras_bce(np.float32([1, 0, 1]), np.float32([0.5, 0.5, 0.0]))
    np.testing.assert_allclose(v[:2], -np.log(0.5 + EPS), rtol=1e-6)
    np.testing.assert_allclose(v[2], -np.log(2 * EPS), rtol=1e-5)   # clipped to eps, + eps
    assert g[2] == 0                                                # outside the clip range


@pytest.mark.parametrize("dup", [False, True])
def test_rpn_loss_matches_torch_autograd(dup):
    d = _inputs(dup=dup)
    (cls, reg, tot), d_obj, d_reg = O.rpn_loss(d["pred_obj"], d["pred_reg"], d["pos"], d["label"],
                                               d["reg_gt"])
    po = torch.tensor(d["pred_obj"], dtype=torch.float64, requires_grad=True)
    pr = torch.tensor(d["pred_reg"], dtype=torch.float64, requires_grad=True)
    B, S = d["label"].shape
    bi = torch.arange(B).repeat_interleave(S)
    y, x, a = torch.tensor(d["pos"]).reshape(-1, 3).T
    p = po[bi, y, x, a]
    lab = torch.tensor(d["label"], dtype=torch.float64).reshape(-1)
    pc = p.clamp(EPS, 1 - EPS)
    bce = -(lab * torch.log(pc + EPS) + (1 - lab) * torch.log(1 - pc + EPS))
    m = lab == 1
    hub = _huber_t(torch.tensor(d["reg_gt"], dtype=torch.float64).reshape(-1, 4)[m], pr[bi, y, x, a][m])
    t_cls, t_reg = bce.mean(), hub.mean()
    total = 2 * t_cls + t_reg
    total.backward()
    np.testing.assert_allclose([cls, reg, tot], [t_cls.item(), t_reg.item(), total.item()], rtol=2e-6)
    np.testing.assert_allclose(d_obj, po.grad.numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(d_reg, pr.grad.numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("dup", [False, True])
def test_head_loss_matches_torch_autograd(dup):
    d = _inputs(1, dup=dup)
    (cls, reg, tot), d_cls, d_reg = O.head_loss(d["cls_prob"], d["reg"], d["tindex"], d["onehot"],
                                                d["gt_mask"], d["gt_coord"])
    cp = torch.tensor(d["cls_prob"], dtype=torch.float64, requires_grad=True)
    rg = torch.tensor(d["reg"], dtype=torch.float64, requires_grad=True)
    B, S = d["tindex"].shape
    bi = torch.arange(B).repeat_interleave(S)
    r = torch.tensor(d["tindex"]).reshape(-1)
    lab = torch.tensor(d["onehot"], dtype=torch.float64).reshape(B * S, -1)
    p = cp[bi, r]
    q = (p / p.sum(-1, keepdim=True)).clamp(EPS, 1 - EPS)
    cce = -(lab * torch.log(q)).sum(-1)
    m = torch.tensor(d["gt_mask"]).reshape(-1) != 0
    c = lab.argmax(-1)
    gt = torch.tensor(d["gt_coord"], dtype=torch.float64).reshape(-1, 4).clamp(-100, 100)
    hub = _huber_t(gt[m], rg[bi, r, c][m])
    t_cls, t_reg = cce.mean(), hub.mean()
    total = 2 * t_cls + t_reg
    total.backward()
    np.testing.assert_allclose([cls, reg, tot], [t_cls.item(), t_reg.item(), total.item()], rtol=2e-6)
    np.testing.assert_allclose(d_cls, cp.grad.numpy(), rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(d_reg, rg.grad.numpy(), rtol=1e-5, atol=1e-9)


def test_no_positive_gives_zero_regression_loss():
    d = _inputs(2)
    d["label"][:] = 0
    (cls, reg, tot), _, d_reg = O.rpn_loss(d["pred_obj"], d["pred_reg"], d["pos"], d["label"],
                                           d["reg_gt"])
    assert reg == 0.0 and tot == 2 * cls and not d_reg.any()


def test_rpn_minibatch_really_repeats_anchors():
    """patch_batch on the 31x31x9 grid: some images list one (y,x,a) as a positive and as a
    negative (quirk of module/data.py:108-118), so the gradient scatter must accumulate."""
    a = O.make_anchor()
    n_dup = 0
    for seed in range(12):
        d = O.synth_inputs(seed=seed, B=1, H=31, W=31, A=9, C=4, G=42, with_fmap=False)
        out = O.patch_batch(dict(image=None, bbox=d["gt"][0], label=d["label"][0]), a, seed=seed)
        pos = out[5]
        lin = (pos[:, 0] * 31 + pos[:, 1]) * 9 + pos[:, 2]
        n_dup += len(lin) - len(np.unique(lin))
    assert n_dup > 0
