"""RoI crop + pool forward on adversarial box sets, against the C restatement of the oracle
(bit-identical to the numpy oracle, tests/test_oracle_c.py): every walk of the staged kernel
is exercised -- narrow boxes (two pooled-column groups per thread), wide boxes (one group per
walk), boxes with samples exactly on feature rows / columns (rows listed twice), samples
outside the image (general walk), flipped and degenerate boxes (direct path), several channel
counts (one, two and three 64-channel slices per RoI).  Bit-exact."""
import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu


def _boxes(rng, n, H, W):
    out = []
    for _ in range(n):
        kind = rng.integers(0, 8)
        y1, x1 = rng.random(2)
        if kind == 0:                                   # tiny
            h, w = rng.random(2) * 0.03
        elif kind == 1:                                 # huge
            y1, x1 = rng.random(2) * 0.1
            h, w = 0.8 + rng.random(2) * 0.2
        elif kind == 2:                                 # corners on grid lines: aligned samples
            y1 = rng.integers(0, H - 1) / (H - 1); x1 = rng.integers(0, W - 1) / (W - 1)
            h = rng.integers(1, H) / (H - 1); w = rng.integers(1, W) / (W - 1)
        elif kind == 3:                                 # sticks out of the image
            y1, x1 = rng.random(2) * 1.4 - 0.3
            h, w = rng.random(2) * 0.8
            out.append([y1, x1, y1 + h, x1 + w]); continue
        elif kind == 4:                                 # wide and flat / tall and thin
            h, w = (0.02, 0.9) if rng.random() < 0.5 else (0.9, 0.02)
            y1, x1 = rng.random(2) * 0.08
        elif kind == 5:                                 # degenerate or flipped
            h, w = rng.choice([0.0, -0.2, 0.3], 2)
        else:
            h, w = rng.random(2) * 0.6
        out.append([y1, x1, min(y1 + h, 1.0) if kind != 5 else y1 + h,
                    min(x1 + w, 1.0) if kind != 5 else x1 + w])
    return np.array(out, np.float32)


class _Path:
    """One way through the forward entry points: how the feature map is laid out and which
    workspace goes with the call."""

    def __init__(self, kind):
        self.kind = kind

    def fmap(self, t):
        if self.kind != "planar":
            return t
        return t.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)     # [B,C,H,W] storage

    def ws(self, B, R, H, W, C, dev):
        from objectdetection_amd import ops
        if self.kind == "per_roi":
            return None
        n = ops.roi_fwd_workspace_bytes(B, R, (H, W, C) if self.kind == "planar" else None)
        return torch.empty(n, dtype=torch.uint8, device=dev)

    def forward(self, fm, rois, crop, pool, want_winner=False):
        """fm [B,H,W,C] contiguous on the device -> (pooled, winner) through this path."""
        from objectdetection_amd import ops
        B, H, W, C = fm.shape
        ws = self.ws(B, rois.shape[1], H, W, C, fm.device)
        prepared = False
        if self.kind == "band_prepared":
            prepared = ops.roi_fwd_prepare(rois, (B, H, W, C), crop, pool, want_winner, ws)
            assert prepared == (not (pool == 1 and want_winner))
        return ops.roi_crop_pool_fwd(self.fmap(fm), rois, crop, pool, want_winner=want_winner,
                                     workspace=ws, prepared=prepared)


@pytest.fixture(params=["per_roi", "band", "band_prepared", "band_split", "planar"])
def ws_for(request, tuning):
    """The forward entry points: frcnn_roi_crop_pool_fwd (one wave per RoI and slice),
    frcnn_roi_crop_pool_fwd_ws with its band kernel forced on at test sizes, and the same entry
    point on a planar [B,C,H,W] map (transposed into the workspace, then the per-RoI kernel)."""
    if request.param in ("band", "band_prepared", "band_split"):   # band_prepared: prepare, then run
        tuning("roi_band_min_wg", 1)
    if request.param == "band_split":                  # an image's RoIs in 7 ranges of workgroups
        tuning("roi_band_split", 7)
        return _Path("band")
    return _Path(request.param)


@pytest.mark.parametrize("C,H,W", [(64, 31, 31), (128, 37, 62), (192, 20, 45), (512, 37, 62),
                                   (32, 75, 100)])
def test_roi_forward_stress(dev, ws_for, C, H, W):
    from objectdetection_amd import ops
    lib = c_oracle.load()
    rng = np.random.default_rng(C + H)
    B, R = 2, 400
    fm = (rng.standard_normal((B, H, W, C)) * 3).astype(np.float32)
    rois = np.stack([_boxes(rng, R, H, W) for _ in range(B)])
    rois[0, 0] = [0, 0, 1, 1]
    rois[0, 1] = [np.nan, 0.1, 0.5, 0.6]
    rois[0, 2] = [0.2, 0.2, 0.2, 0.2]
    for pool, win in ((2, True), (2, False), (1, False)):
        out, w = ws_for.forward(torch.tensor(fm, device=dev), torch.tensor(rois, device=dev),
                                (14, 14), pool, want_winner=win)
        for b in range(B):
            ro, rw = c_oracle.roi_crop_pool(lib, fm[b], rois[b], (14, 14), pool, want_winner=win)
            got = out[b].cpu().numpy()
            bad = np.argwhere(~((got == ro) | (np.isnan(got) & np.isnan(ro))))
            assert len(bad) == 0, (pool, win, b, bad[:5], rois[b, bad[0][0]])
            if win:
                np.testing.assert_array_equal(w[b].cpu().numpy(), rw)


def test_roi_forward_other_crop_sizes(dev, ws_for):
    """crop 16x16 / pool 2 (pooled width 8: two full column groups), 12x12 / pool 1 (three
    groups), 7x7 / pool 1."""
    from objectdetection_amd import ops
    lib = c_oracle.load()
    rng = np.random.default_rng(5)
    H, W, C = 25, 40, 128
    fm = rng.standard_normal((1, H, W, C)).astype(np.float32)
    rois = _boxes(rng, 300, H, W)[None]
    for crop, pool in (((16, 16), 2), ((12, 12), 1), ((7, 7), 1), ((14, 8), 2)):
        out, _ = ws_for.forward(torch.tensor(fm, device=dev), torch.tensor(rois, device=dev), crop, pool)
        ro, _ = c_oracle.roi_crop_pool(lib, fm[0], rois[0], crop, pool, want_winner=False)
        got = out[0].cpu().numpy()
        assert ((got == ro) | (np.isnan(got) & np.isnan(ro))).all(), (crop, pool)


def test_nchw_to_nhwc(dev):
    from objectdetection_amd import ops
    rng = np.random.default_rng(8)
    for shape in ((2, 512, 37, 62), (1, 5, 3, 7), (3, 64, 1, 1), (1, 130, 9, 70)):
        x = torch.tensor(rng.standard_normal(shape).astype(np.float32), device=dev)
        assert torch.equal(ops.nchw_to_nhwc(x), x.permute(0, 2, 3, 1).contiguous())


def test_store_data_survives_the_next_row(dev):
    """Guard for the 16-byte store with a scalar row offset (a vector instruction of the next
    row must not overwrite the store's data registers while the store still reads them: seen as
    the x / y components of channel quads 12-15 of a 16-lane group carrying the NEXT pooled
    row's values).  Every feature row holds its own value range and the boxes sit on grid
    lines, so a value from another row is off by thousands, not by rounding."""
    from objectdetection_amd import ops
    lib = c_oracle.load()
    H, W, C = 29, 29, 128
    y = np.arange(H, dtype=np.float32)[:, None, None] * 1000.0
    x = np.arange(W, dtype=np.float32)[None, :, None] * 10.0
    c = np.arange(C, dtype=np.float32)[None, None, :] / 256.0
    fm = (y + x + c)[None].astype(np.float32)
    rois = np.array([[[0, 0, 1, 1], [0, 0, 13 / 28, 13 / 28], [7 / 28, 7 / 28, 1, 1],
                      [0.1, 0.2, 0.9, 0.7]]], np.float32)
    rois = np.tile(rois, (1, 50, 1))
    out, _ = ops.roi_crop_pool_fwd(torch.tensor(fm, device=dev), torch.tensor(rois, device=dev))
    ro, _ = c_oracle.roi_crop_pool(lib, fm[0], rois[0], (14, 14), 2, want_winner=False)
    got = out[0].cpu().numpy()
    bad = np.argwhere(got != ro)
    assert len(bad) == 0, ("first mismatches (roi, p, q, channel):", bad[:6],
                           "channel quads", sorted(set((bad[:, 3] // 4 % 16).tolist())),
                           "components", sorted(set((bad[:, 3] % 4).tolist())))


@pytest.mark.parametrize("B", [3, 5, 11])
def test_band_kernel_workgroup_order(dev, tuning, B):
    """(image, band) pairs are dealt to the 8 XCDs in rounds of eight; a count that is not a
    multiple of eight ends with pairs in plain order.  37x62 maps have three bands: B = 3, 5, 11
    give 9, 15 and 33 pairs."""
    from objectdetection_amd import ops
    tuning("roi_band_min_wg", 1)
    lib = c_oracle.load()
    rng = np.random.default_rng(B)
    H, W, C, R = 37, 62, 64, 60
    assert ops.roi_fwd_kernel_name(B, H, W, C, R, (14, 14), 2).startswith("roi_fwd_band")
    fm = rng.standard_normal((B, H, W, C)).astype(np.float32)
    rois = np.stack([_boxes(rng, R, H, W) for _ in range(B)])
    ws = torch.empty(ops.roi_fwd_workspace_bytes(B, R), dtype=torch.uint8, device=dev)
    out, w = ops.roi_crop_pool_fwd(torch.tensor(fm, device=dev), torch.tensor(rois, device=dev),
                                   (14, 14), 2, want_winner=True, workspace=ws)
    for b in range(B):
        ro, rw = c_oracle.roi_crop_pool(lib, fm[b], rois[b], (14, 14), 2, want_winner=True)
        got = out[b].cpu().numpy()
        assert ((got == ro) | (np.isnan(got) & np.isnan(ro))).all(), b
        np.testing.assert_array_equal(w[b].cpu().numpy(), rw)
