"""Loss-assembly kernels (next row f2) against the oracle, through the C ABI.
Tolerance: 1e-5 relative on the scalars and on every gradient element (device logf and the
f32 reduction order differ from the float64 oracle sums by ULPs)."""
import numpy as np
import pytest
import torch

from oracle import frcnn_oracle as O
from tests.test_oracle_loss import _inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def T(a, dev):
    return torch.as_tensor(a, device=dev)


@pytest.mark.parametrize("shape", [dict(), dict(B=64, H=37, W=62, A=9, S=256, R=2000, K=21)])
def test_rpn_loss_parity(dev, shape):
    from objectdetection_amd import ops
    d = _inputs(3, **shape)
    (cls, reg, tot), d_obj, d_reg = O.rpn_loss(d["pred_obj"], d["pred_reg"], d["pos"], d["label"],
                                               d["reg_gt"])
    out = ops.rpn_loss(T(d["pred_obj"], dev), T(d["pred_reg"], dev), T(d["pos"], dev),
                       T(d["label"], dev), T(d["reg_gt"], dev))
    np.testing.assert_allclose(out["loss"].cpu().numpy(), [cls, reg, tot], rtol=1e-5)
    np.testing.assert_allclose(out["d_obj"].cpu().numpy(), d_obj, rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(out["d_reg"].cpu().numpy(), d_reg, rtol=1e-5, atol=1e-10)
    part = out["partial"].cpu().numpy()
    assert part[:, 3].sum() == 0
    np.testing.assert_array_equal(part[:, 2], (d["label"].astype(np.int64) == 1).sum(1))
    # forward only
    out2 = ops.rpn_loss(T(d["pred_obj"], dev), T(d["pred_reg"], dev), T(d["pos"], dev),
                        T(d["label"], dev), T(d["reg_gt"], dev), want_grad=False)
    np.testing.assert_array_equal(out2["loss"].cpu().numpy(), out["loss"].cpu().numpy())


@pytest.mark.parametrize("shape", [dict(), dict(B=64, H=4, W=4, A=2, S=128, R=2000, K=21)])
def test_head_loss_parity(dev, shape):
    from objectdetection_amd import ops
    d = _inputs(4, **shape)
    (cls, reg, tot), d_cls, d_reg = O.head_loss(d["cls_prob"], d["reg"], d["tindex"], d["onehot"],
                                                d["gt_mask"], d["gt_coord"])
    out = ops.head_loss(T(d["cls_prob"], dev), T(d["reg"], dev), T(d["tindex"], dev),
                        T(d["onehot"], dev), T(d["gt_mask"], dev), T(d["gt_coord"], dev))
    np.testing.assert_allclose(out["loss"].cpu().numpy(), [cls, reg, tot], rtol=1e-5)
    np.testing.assert_allclose(out["d_cls"].cpu().numpy(), d_cls, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(out["d_reg"].cpu().numpy(), d_reg, rtol=1e-5, atol=1e-10)
    assert out["partial"].cpu().numpy()[:, 3].sum() == 0


def test_no_positives_and_bad_indices(dev):
    from objectdetection_amd import ops
    d = _inputs(5)
    d["label"][:] = 0
    d["pos"][1, 3] = [99, 0, 0]                                      # out of range: counted, skipped
    out = ops.rpn_loss(T(d["pred_obj"], dev), T(d["pred_reg"], dev), T(d["pos"], dev),
                       T(d["label"], dev), T(d["reg_gt"], dev))
    loss = out["loss"].cpu().numpy()
    assert loss[1] == 0.0 and np.isfinite(loss).all()
    assert not out["d_reg"].cpu().numpy().any()
    assert out["partial"].cpu().numpy()[1, 3] == 1


def test_module_names(dev):
    """The reference-named front end (module/loss.py, model_train.py:626-657)."""
    from objectdetection_amd.module import Loss_bbr, rpn_losses, frcn_losses
    d = _inputs(6)
    B, H, W, A = d["pred_obj"].shape
    o, r, t, d_obj, d_reg = rpn_losses(T(d["pred_obj"], dev), T(d["pred_reg"].reshape(B, H, W, A * 4), dev),
                                       T(d["label"][..., None], dev), T(d["pos"], dev),
                                       T(d["reg_gt"], dev))
    (cls, reg, tot), _, ref_d_reg = O.rpn_loss(d["pred_obj"], d["pred_reg"], d["pos"], d["label"],
                                               d["reg_gt"])
    np.testing.assert_allclose([o.item(), r.item(), t.item()], [cls, reg, tot], rtol=1e-5)
    assert d_reg.shape == (B, H, W, A * 4)
    np.testing.assert_allclose(d_reg.cpu().numpy().reshape(ref_d_reg.shape), ref_d_reg, rtol=1e-5,
                               atol=1e-10)
    K = d["cls_prob"].shape[-1]
    R = d["cls_prob"].shape[1]
    o2, r2, t2, _, _ = frcn_losses(T(d["cls_prob"], dev), T(d["reg"].reshape(B, R, K * 4), dev),
                                   T(d["onehot"], dev), T(d["gt_mask"], dev), T(d["gt_coord"], dev),
                                   T(d["tindex"], dev))
    (c2, g2, s2), _, _ = O.head_loss(d["cls_prob"], d["reg"], d["tindex"], d["onehot"],
                                     d["gt_mask"], d["gt_coord"])
    np.testing.assert_allclose([o2.item(), r2.item(), t2.item()], [c2, g2, s2], rtol=1e-5)
    v = Loss_bbr()(T(np.float32([0, 0]), dev), T(np.float32([0.5, 2.0]), dev))
    np.testing.assert_allclose(v.cpu().numpy(), [0.125, 1.5])


def test_repeated_positions_accumulate(dev):
    """One anchor sampled as a positive and as a negative, anchors and proposal rows listed
    two and three times: gradients are the sums (as the gradient of a gather is), run twice
    to show the result does not depend on thread order."""
    from objectdetection_amd import ops
    d = _inputs(7, dup=True)
    (cls, reg, tot), d_obj, d_reg = O.rpn_loss(d["pred_obj"], d["pred_reg"], d["pos"], d["label"],
                                               d["reg_gt"])
    runs = [ops.rpn_loss(T(d["pred_obj"], dev), T(d["pred_reg"], dev), T(d["pos"], dev),
                         T(d["label"], dev), T(d["reg_gt"], dev)) for _ in range(2)]
    out = runs[0]
    np.testing.assert_allclose(out["loss"].cpu().numpy(), [cls, reg, tot], rtol=1e-5)
    np.testing.assert_allclose(out["d_obj"].cpu().numpy(), d_obj, rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(out["d_reg"].cpu().numpy(), d_reg, rtol=1e-5, atol=1e-10)
    assert torch.equal(runs[0]["d_obj"], runs[1]["d_obj"]) and torch.equal(runs[0]["d_reg"], runs[1]["d_reg"])
    (c2, g2, s2), d_cls, d_hreg = O.head_loss(d["cls_prob"], d["reg"], d["tindex"], d["onehot"],
                                              d["gt_mask"], d["gt_coord"])
    h = ops.head_loss(T(d["cls_prob"], dev), T(d["reg"], dev), T(d["tindex"], dev),
                      T(d["onehot"], dev), T(d["gt_mask"], dev), T(d["gt_coord"], dev))
    np.testing.assert_allclose(h["loss"].cpu().numpy(), [c2, g2, s2], rtol=1e-5)
    np.testing.assert_allclose(h["d_cls"].cpu().numpy(), d_cls, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(h["d_reg"].cpu().numpy(), d_hreg, rtol=1e-5, atol=1e-10)


def test_rpn_sample_output_feeds_the_loss(dev):
    """The real thing: frcnn_rpn_sample on the 31x31x9 grid (where repeated positions occur)
    into frcnn_rpn_loss, against the oracle's accumulating scatter."""
    from objectdetection_amd import ops
    B = 8
    d = O.synth_inputs(seed=21, B=B, H=31, W=31, A=9, C=4, G=42, with_fmap=False)
    a = ops.anchors(31, 31, (128, 256, 512), (0.5, 1, 2), 500.0, 500.0, dev)
    gt = T(d["gt"], dev)
    _, best, flags, _ = ops.iou_assign(a.reshape(-1, 4), gt)
    s = ops.rpn_sample(a, gt, best, flags, seed=21)
    pos = s["pos"].cpu().numpy()
    lin = (pos[..., 0] * 31 + pos[..., 1]) * 9 + pos[..., 2]
    assert any(len(np.unique(r)) < len(r) for r in lin), "no repeated anchor in 8 images"
    reg_gt = torch.nan_to_num(s["reg"], nan=0.0, posinf=0.0, neginf=0.0)
    pred_reg = T(d["coord"].reshape(B, 31, 31, 9, 4), dev)
    out = ops.rpn_loss(T(d["score"], dev), pred_reg, s["pos"], s["label"], reg_gt)
    (cls, reg, tot), d_obj, d_reg = O.rpn_loss(d["score"], d["coord"].reshape(B, 31, 31, 9, 4), pos,
                                               s["label"].cpu().numpy(), reg_gt.cpu().numpy())
    np.testing.assert_allclose(out["loss"].cpu().numpy(), [cls, reg, tot], rtol=1e-5)
    np.testing.assert_allclose(out["d_obj"].cpu().numpy(), d_obj, rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(out["d_reg"].cpu().numpy(), d_reg, rtol=1e-5, atol=1e-10)
