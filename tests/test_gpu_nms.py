"""NMS through the C ABI vs the oracle on identical boxes: ragged sizes, heavy
overlap (every chunk is walked), early exit, zero-area boxes, ties, thresholds."""
import numpy as np
import pytest
import torch

from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def random_boxes(rng, B, K, spread):
    cy, cx = rng.uniform(0, 1, (B, K)), rng.uniform(0, 1, (B, K))
    h, w = rng.uniform(0.02, spread, (B, K)), rng.uniform(0.02, spread, (B, K))
    b = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1)
    return np.clip(b, 0, 1).astype(np.float32)


def check(dev, boxes, thr, post_n):
    from objectdetection_amd import ops
    prop, keep, cnt = ops.nms(T(boxes, dev), thr, post_n)
    rp, rk, rc = O.combined_nms(boxes, None, post_n, thr)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
    np.testing.assert_array_equal(keep.cpu().numpy(), rk)
    np.testing.assert_array_equal(prop.cpu().numpy(), rp)
    if boxes.shape[0] <= ops.NMS_FUSED_MAX_B:              # the one-launch route: same results
        prop, keep, cnt = ops.nms_fused(T(boxes, dev), thr, post_n)
        np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
        np.testing.assert_array_equal(keep.cpu().numpy(), rk)
        np.testing.assert_array_equal(prop.cpu().numpy(), rp)
    return rc


@pytest.mark.parametrize("K,post_n,spread", [(1, 4, 0.5), (63, 10, 0.5), (64, 64, 0.3),
                                             (65, 100, 0.3), (1000, 300, 0.2), (3000, 3000, 0.6),
                                             (2500, 50, 0.1),
                                             (6000, 5000, 0.03)])   # > 2048 kept: words beyond the registers
def test_nms_sizes(dev, K, post_n, spread):
    rng = np.random.default_rng(K)
    check(dev, random_boxes(rng, 3, K, spread), 0.7, post_n)


def test_nms_heavy_overlap_walks_every_chunk(dev):
    # boxes jittered around 40 centres: few survivors, the scan reaches the last candidate
    rng = np.random.default_rng(1)
    K = 4000
    centres = rng.uniform(0.2, 0.8, (40, 2))
    idx = rng.integers(0, 40, (2, K))
    c = centres[idx] + rng.normal(0, 0.004, (2, K, 2))
    half = 0.1 + rng.normal(0, 0.002, (2, K, 2))
    boxes = np.concatenate([c - half, c + half], -1).astype(np.float32)
    rc = check(dev, boxes, 0.7, 1500)
    assert rc.max() < 1500


def test_nms_zero_area_ties_and_thresholds(dev):
    rng = np.random.default_rng(2)
    boxes = random_boxes(rng, 2, 700, 0.4)
    boxes[0, 10:40, 2] = boxes[0, 10:40, 0]                 # zero height
    boxes[0, 100:130] = boxes[0, 100]                       # identical boxes
    boxes[1, 50:60] = 0.0                                   # zero-padded rows
    boxes[1, 200, :] = [0.9, 0.1, 0.2, 0.8]                 # flipped corners
    for thr in (0.0, 0.3, 0.7, 1.0):
        check(dev, boxes, thr, 500)


def test_nms_exact_threshold_pair(dev):
    # IoU exactly 0.5 must not suppress at thr 0.5; just above must
    b = np.array([[[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 0.5], [0.0, 0.0, 1.0, 0.5000001]]],
                 np.float32)
    check(dev, b, 0.5, 3)
    check(dev, b, np.float32(0.49999997), 3)


def test_nms_workspace_checks(dev):
    from objectdetection_amd import ops, _lib
    boxes = T(random_boxes(np.random.default_rng(0), 1, 128, 0.3), dev)
    small = torch.empty(16, dtype=torch.uint8, device=dev)
    with pytest.raises(_lib.FrcnnError, match="workspace"):
        ops.nms(boxes, 0.7, 10, workspace=small)
