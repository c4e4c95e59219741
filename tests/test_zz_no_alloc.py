"""frcnn_nms_static (the entry point that allocates nothing: its workspace is a device array
built into the library) against frcnn_nms with a caller-owned workspace, through the C ABI.
Bit-exact on proposals, keep positions and counts; two calls in a row on the shared array,
the second with a smaller input, so stale contents of the first would show."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _boxes(seed, B, K, clustered):
    rng = np.random.default_rng(seed)
    if clustered:
        c = 0.5 + rng.normal(0, 0.02, (B, K, 2))
        half = 0.15 + rng.normal(0, 0.01, (B, K, 2))
        b = np.concatenate([c - half, c + half], -1)
    else:
        cy, cx = rng.uniform(0, 1, (B, K)), rng.uniform(0, 1, (B, K))
        h = np.exp(rng.uniform(np.log(0.03), np.log(0.5), (B, K)))
        w = np.exp(rng.uniform(np.log(0.03), np.log(0.5), (B, K)))
        b = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1)
    return np.clip(b, 0, 1).astype(np.float32)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _both(dev, boxes, thr, post_n):
    from objectdetection_amd import _lib
    L = _lib.lib
    b = torch.from_numpy(boxes).to(dev)
    B, K = b.shape[0], b.shape[1]
    st = torch.cuda.current_stream(dev).cuda_stream
    out = []
    for static in (False, True):
        prop = torch.full((B, post_n, 4), -7.0, dtype=torch.float32, device=dev)
        keep = torch.full((B, post_n), -7, dtype=torch.int32, device=dev)
        cnt = torch.full((B,), -7, dtype=torch.int32, device=dev)
        if static:
            rc = L.frcnn_nms_static(_p(b), B, K, thr, post_n, _p(prop), _p(keep), _p(cnt), st)
        else:
            ws = torch.empty(L.frcnn_nms_workspace_bytes(B, K, post_n), dtype=torch.uint8, device=dev)
            rc = L.frcnn_nms(_p(b), B, K, thr, post_n, _p(prop), _p(keep), _p(cnt), _p(ws),
                             ws.numel(), st)
        assert rc == 0, _lib.status_string(rc)
        torch.cuda.synchronize(dev)
        out.append((prop.cpu().numpy(), keep.cpu().numpy(), cnt.cpu().numpy()))
    return out


def test_static_nms_equals_workspace_nms_two_calls(dev):
    from oracle import frcnn_oracle as O
    # first call: 3 images, 5,000 clustered candidates (several chunks, long keep-tile lists);
    # second call: smaller in every dimension, on the array the first one left behind
    for seed, B, K, post_n, clustered in ((1, 3, 5000, 600, True), (2, 2, 1500, 300, False)):
        boxes = _boxes(seed, B, K, clustered)
        (p0, k0, c0), (p1, k1, c1) = _both(dev, boxes, 0.7, post_n)
        np.testing.assert_array_equal(c1, c0)
        np.testing.assert_array_equal(k1, k0)
        np.testing.assert_array_equal(p1, p0)
        rp, rk, rc = O.combined_nms(boxes[:1], None, post_n, 0.7)     # and the oracle, one image
        np.testing.assert_array_equal(k1[0], rk[0])
        np.testing.assert_array_equal(p1[0], rp[0])
        assert c1[0] == rc[0]


def test_static_nms_rejects_what_does_not_fit(dev):
    from objectdetection_amd import _lib
    L = _lib.lib
    one = torch.zeros(16, device=dev)
    B = 64                                                     # 64 images x 2,000 kept: 25 MB
    assert L.frcnn_nms_workspace_bytes(B, 12000, 2000) > _lib.NMS_STATIC_BYTES
    rc = L.frcnn_nms_static(_p(one), B, 12000, 0.7, 2000, _p(one), None, None, None)
    assert rc == _lib.E_STATIC_LIMIT
    assert "scratch" in _lib.status_string(rc)
