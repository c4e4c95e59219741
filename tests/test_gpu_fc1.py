"""f4: the head's first dense layer on the pooled tensor (module/model.py:31-32), through the C
ABI, against the C oracle's k-ordered fmaf chain -- bit for bit -- and against float64."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as OC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clib():
    return OC.load()


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("M,K,N,bias,relu", [
    (200, 25088, 256, True, True),        # the reference's K = 7*7*512; ragged M (two row tiles)
    (128, 512, 128, False, False),        # one tile, no epilogue
    (130, 36, 132, True, True),           # ragged in every dimension, K not a multiple of 32
    (1, 64, 4, True, False),              # a single row
    (1100, 1024, 1156, True, True),       # more than one super-tile in both directions
])
def test_fc1_matches_the_fmaf_chain(dev, clib, M, K, N, bias, relu):
    from objectdetection_amd import ops
    rng = np.random.default_rng(M + K + N)
    x = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if bias else None
    got = ops.fc1_fwd(T(x, dev), T(w, dev), None if b is None else T(b, dev), relu).cpu().numpy()
    ref = OC.fc1(clib, x, w, b, relu)
    np.testing.assert_array_equal(got, ref)
    # and both agree with float64 to the rounding of a K-term float32 chain
    # (tolerance: 4e-7 * sum |x w| per output, the bound the matrix instruction's chain keeps)
    ref64 = x.astype(np.float64) @ w.astype(np.float64) + (0 if b is None else b.astype(np.float64))
    if relu:
        ref64 = np.maximum(ref64, 0)
    bound = 4e-7 * (np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)) + 1e-6
    assert (np.abs(got - ref64) <= bound).all()


def test_fc1_on_the_pooled_tensor_is_flatten_then_dense(dev, clib):
    """module.model.fc1 takes the pooled [B,R,7,7,C] tensor as it stands: same values as
    flattening (h,w,c) on the host and multiplying."""
    from objectdetection_amd import ops
    from objectdetection_amd.module import model as M
    rng = np.random.default_rng(5)
    B, R, C, U = 2, 37, 64, 96
    fm = np.maximum(rng.standard_normal((B, 9, 11, C)), 0).astype(np.float32)
    y1 = rng.uniform(0, 0.6, (B, R)); x1 = rng.uniform(0, 0.6, (B, R))
    prop = np.stack([y1, x1, y1 + rng.uniform(0.05, 0.4, (B, R)), x1 + rng.uniform(0.05, 0.4, (B, R))],
                    -1).astype(np.float32)
    pooled, _ = ops.roi_crop_pool_fwd(T(fm, dev), T(prop, dev), (14, 14), 2)
    w = (rng.standard_normal((7 * 7 * C, U)) * 0.02).astype(np.float32)
    b = rng.standard_normal(U).astype(np.float32)
    got = M.fc1(pooled, T(w, dev), T(b, dev))
    assert tuple(got.shape) == (B, R, U)
    flat = pooled.cpu().numpy().reshape(B * R, -1)
    np.testing.assert_array_equal(got.cpu().numpy().reshape(B * R, U), OC.fc1(clib, flat, w, b, True))


def test_fc1_rejects_bad_shapes(dev):
    from objectdetection_amd import ops
    from objectdetection_amd._lib import FrcnnError
    x = torch.zeros((4, 10), device=dev)
    with pytest.raises(FrcnnError):
        ops.fc1_fwd(x, torch.zeros((10, 8), device=dev))        # K % 4 != 0
    with pytest.raises(ValueError):
        ops.fc1_fwd(torch.zeros((4, 12), device=dev), torch.zeros((16, 8), device=dev))
