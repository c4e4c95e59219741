"""Input-side kernels (next row f3) against the oracle, through the C ABI: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _ragged(seed, B, max_n):
    rng = np.random.default_rng(seed)
    boxes, labels = [], []
    for b in range(B):
        n = int(rng.integers(0, max_n + 1))
        y1, x1 = rng.random((2, n), dtype=np.float32) * 0.5
        boxes.append(np.stack([y1, x1, y1 + 0.3, x1 + 0.4], 1).astype(np.float32).reshape(n, 4))
        labels.append(rng.integers(0, 20, n).astype(np.int64))
    return boxes, labels


@pytest.mark.parametrize("B,G,max_n", [(5, 42, 42), (64, 42, 60), (3, 24, 0)])
def test_pad_gt_and_flip(dev, B, G, max_n):
    from objectdetection_amd import ops
    boxes, labels = _ragged(B, B, max_n)
    off = np.zeros(B + 1, np.int32)
    off[1:] = np.cumsum([len(b) for b in boxes])
    flip = (np.arange(B) % 2).astype(np.uint8)
    cat = np.concatenate(boxes) if off[-1] else np.zeros((0, 4), np.float32)
    lab = np.concatenate(labels) if off[-1] else np.zeros(0, np.int64)
    gt, gl, cnt = ops.pad_gt(torch.tensor(cat, device=dev), torch.tensor(lab, device=dev),
                             torch.tensor(off, device=dev), G, torch.tensor(flip, device=dev), True)
    for b in range(B):
        _, rb, rl = O.random_augmentation(None, labels[b], boxes[b], G, flip=bool(flip[b]))
        np.testing.assert_array_equal(gt[b].cpu().numpy(), rb)
        np.testing.assert_array_equal(gl[b].cpu().numpy(), rl)
    np.testing.assert_array_equal(cnt.cpu().numpy(), np.diff(off))
    # counter-key decisions
    gt2, _, _ = ops.pad_gt(torch.tensor(cat, device=dev), None, torch.tensor(off, device=dev), G,
                           None, True, seed=11, image_base=100)
    for b in range(B):
        _, rb, _ = O.random_augmentation(None, None, boxes[b], G, seed=11, image_index=100 + b)
        np.testing.assert_array_equal(gt2[b].cpu().numpy(), rb)
    # no flip requested
    gt3, _, _ = ops.pad_gt(torch.tensor(cat, device=dev), None, torch.tensor(off, device=dev), G)
    for b in range(B):
        np.testing.assert_array_equal(gt3[b].cpu().numpy(), O.pad_gt(boxes[b], None, G)[0])


@pytest.mark.parametrize("shape", [(4, 7, 9, 3), (8, 500, 500, 3), (2, 5, 300, 1)])
def test_flip_images(dev, shape):
    from objectdetection_amd import ops
    rng = np.random.default_rng(1)
    img = rng.random(shape, dtype=np.float32)
    flip = (rng.random(shape[0]) < 0.5).astype(np.uint8)
    flip[0] = 1
    out = ops.flip_images(torch.tensor(img, device=dev), torch.tensor(flip, device=dev)).cpu().numpy()
    for b in range(shape[0]):
        ref = img[b][:, ::-1, :] if flip[b] else img[b]
        np.testing.assert_array_equal(out[b], ref)
    out2 = ops.flip_images(torch.tensor(img, device=dev), None, seed=5, image_base=3).cpu().numpy()
    for b in range(shape[0]):
        ref = img[b][:, ::-1, :] if O.flip_decision(5, 3 + b) else img[b]
        np.testing.assert_array_equal(out2[b], ref)


def test_module_names(dev):
    from objectdetection_amd.module import data_preprocess, random_augmentation
    boxes, labels = _ragged(9, 4, 10)
    img = torch.rand((4, 16, 16, 3), device=dev)
    d = data_preprocess(img, boxes, labels)
    assert d["bbox"].shape == (4, 42, 4) and d["label"].shape == (4, 42)
    np.testing.assert_array_equal(d["bbox"][1].cpu().numpy(), O.pad_gt(boxes[1], None, 42)[0])
    a = random_augmentation(img, boxes, labels, seed=3)
    for b in range(4):
        _, rb, rl = O.random_augmentation(None, labels[b], boxes[b], 42, seed=3, image_index=b)
        np.testing.assert_array_equal(a["bbox"][b].cpu().numpy(), rb)
        np.testing.assert_array_equal(a["label"][b].cpu().numpy(), rl)
        ref = img[b].flip(1) if O.flip_decision(3, b) else img[b]
        assert torch.equal(a["image"][b], ref)
