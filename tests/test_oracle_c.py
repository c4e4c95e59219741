"""The C restatement must agree bit-for-bit with the numpy oracle and the fixtures."""
import os

import numpy as np
import pytest

from oracle import c_oracle, frcnn_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def lib():
    c_oracle.build()
    return c_oracle.load()


def test_c_matches_numpy_on_seeded_inputs(lib):
    H, W, A, C, G = 14, 17, 9, 16, 12
    a = O.make_anchor(H, W, base_w=600.0, base_h=400.0)
    d = O.synth_inputs(3, 1, H, W, A, C, G)
    st = O.rpn_assign(a, d["gt"][0])
    mi, best = c_oracle.iou_argmax(lib, a, d["gt"][0])
    np.testing.assert_array_equal(mi, st["max_iou"])
    np.testing.assert_array_equal(best, st["best"])
    s = d["score"].reshape(-1).copy()
    s[5:40] = s[7]
    v, i = c_oracle.topk(lib, s, 900)
    rv, ri = O.top_k(s[None], 900)
    np.testing.assert_array_equal(i, ri[0])
    np.testing.assert_array_equal(v, rv[0])
    cand = c_oracle.decode_gather(lib, a, d["coord"], i)
    ref = O.inverse_trans(a, d["coord"]).reshape(-1, 4)[i]
    np.testing.assert_allclose(cand, ref, rtol=1e-6, atol=1e-7)       # libm expf vs numpy exp
    prop, keep, n = c_oracle.nms(lib, ref, 0.7, 200)
    rp, rk, rc = O.combined_nms(ref[None], None, 200, 0.7)
    assert n == rc[0]
    np.testing.assert_array_equal(keep, rk[0])
    np.testing.assert_array_equal(prop, rp[0])
    rp[0, 0] = [0, 0, 1, 1]
    rp[0, 1] = [0.9, 0.2, 0.1, 0.8]
    rp[0, 2] = [-0.2, 0.0, 0.5, 1.1]
    for pool, crop in ((2, (14, 14)), (1, (5, 7))):
        out, win = c_oracle.roi_crop_pool(lib, d["fmap"][0], rp[0], crop, pool)
        ro, ra = O.roi_crop_pool(d["fmap"], rp, crop, pool)
        np.testing.assert_array_equal(out, ro[0])
        if pool > 1:
            np.testing.assert_array_equal(win, ra[0])


@pytest.mark.parametrize("name", ["ref_small", "ship_small"])
def test_c_matches_golden(lib, name):
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    for b in range(g["gt"].shape[0]):
        prop, keep, n = c_oracle.nms(lib, g["cand"][b], 0.7, int(g["post_n"]))
        assert n == g["counts"][b]
        np.testing.assert_array_equal(keep, g["keep_pos"][b])
        out, win = c_oracle.roi_crop_pool(lib, g["fmap"][b], g["proposed"][b, :24])
        np.testing.assert_array_equal(out, g["pooled24"][b])
        np.testing.assert_array_equal(win, g["winner24"][b])
        _, i = c_oracle.topk(lib, g["score"][b].reshape(-1), int(g["pre_k"]))
        np.testing.assert_array_equal(i, g["top_idx"][b])


@pytest.mark.parametrize("pool,crop", [(2, (14, 14)), (1, (5, 7)), (2, (4, 6))])
def test_c_backward_matches_numpy(lib, pool, crop):
    """RoI backward: the C scatter in the defined order equals the numpy statement bit for bit
    (boxes with shared cells, a flipped box, boxes partly outside the image, a zero box)."""
    rng = np.random.default_rng(5 + pool)
    H, W, C, R = 9, 11, 8, 14
    fm = np.maximum(rng.standard_normal((1, H, W, C)).astype(np.float32), 0)
    y1, x1 = rng.uniform(-0.05, 0.6, R), rng.uniform(-0.05, 0.6, R)
    prop = np.stack([y1, x1, y1 + rng.uniform(0.02, 0.6, R), x1 + rng.uniform(0.02, 0.6, R)],
                    1).astype(np.float32)[None]
    prop[0, 0] = [0, 0, 1, 1]
    prop[0, 1] = [0, 0, 0, 0]
    prop[0, 2] = [0.8, 0.1, 0.1, 0.9]
    ro, ra = O.roi_crop_pool(fm, prop, crop, pool)
    dY = rng.standard_normal(ro.shape).astype(np.float32)
    ref = O.roi_crop_pool_bwd(dY, ra, fm.shape, prop, crop, pool)
    got = c_oracle.roi_crop_pool_bwd(lib, dY[0], ra[0] if pool > 1 else None, prop[0], (H, W, C),
                                     crop, pool)
    np.testing.assert_array_equal(got, ref[0])
