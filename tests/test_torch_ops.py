"""The entry points as PyTorch operators (objectdetection_amd/torch_ops.py): the operator table
covers the header, fake tensors propagate shapes without a GPU, and on the GPU the operators
return what the ctypes front-end returns, with frcnn_roi_crop_pool_bwd as the registered
gradient of the fused crop + pool."""
import numpy as np
import pytest
import torch

from test_abi import header_symbols


def test_operator_table_covers_the_header():
    from objectdetection_amd import torch_ops
    covered = {e for eps in torch_ops.OP_ENTRY_POINTS.values() for e in eps}
    helpers = set(torch_ops.HELPER_ENTRY_POINTS)
    syms = set(header_symbols())
    assert covered | helpers == syms, (sorted(syms - covered - helpers), sorted((covered | helpers) - syms))
    assert not (covered & helpers)
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.frcnn, name), name


def test_fake_tensors_propagate_shapes_without_a_gpu():
    from torch._subclasses.fake_tensor import FakeTensorMode
    from objectdetection_amd import torch_ops  # noqa: F401
    F = torch.ops.frcnn
    with FakeTensorMode():
        e = lambda *s, dt=torch.float32: torch.empty(s, dtype=dt, device="cuda")
        a = F.anchors(e(1), 37, 62, [128., 256., 512.], [0.5, 1., 2.], 1000.0, 600.0)
        assert a.shape == (37, 62, 9, 4) and a.device.type == "cuda"
        N = 37 * 62 * 9
        mi, best, flags = F.iou_assign(a.reshape(-1, 4), e(3, 42, 4), 0.3, 0.3)
        assert mi.shape == (3, N) and best.dtype == torch.int32 and flags.dtype == torch.uint8
        out = F.rpn_sample(a, e(3, 42, 4), best, flags, 1, 0, 256, 128)
        assert out[4].shape == (3, 256, 3) and out[4].dtype == torch.int64
        v, i, c = F.topk(e(3, N), 12000, a.reshape(-1, 4), e(3, N, 4))
        assert i.shape == (3, 12000) and c.shape == (3, 12000, 4)
        p, k, n = F.nms(c, 0.7, 2000)
        assert p.shape == (3, 2000, 4) and k.dtype == torch.int32 and n.shape == (3,)
        f = e(3, 37, 62, 512).requires_grad_(True)
        pooled, win = F.roi_crop_pool(f, p, 14, 14, 2)
        assert pooled.shape == (3, 2000, 7, 7, 512) and win.dtype == torch.uint8
        assert pooled.requires_grad and not win.requires_grad        # autograd is registered
        g = F.roi_crop_pool_backward(e(3, 2000, 7, 7, 512), win, p, 3, 37, 62, 512, 14, 14, 2)
        assert g.shape == f.shape
        lab = F.label_proposals(p, e(3, 42, 4), e(3, 42, dt=torch.int64), 0.5, 0.1, 20, 21)
        assert lab[2].shape == (3, 2000, 21) and lab[0].dtype == torch.int64
        ti, cnt = F.sample_proposals(lab[0], 20, 0, 0, 128, 32)
        assert ti.shape == (3, 128) and cnt.shape == (3, 4)
        d = F.detect_postprocess(p, e(3, 2000, 21), e(3, 2000, 84), 20, 30, 0.7, 0.9)
        assert d[0].shape == (3, 30, 4) and d[4].shape == (3,)
        assert F.decode_clip(a.reshape(-1, 4), e(3, N, 4), False).shape == (3, N, 4)
        assert F.nchw_to_nhwc(e(2, 64, 9, 11)).shape == (2, 9, 11, 64)
        l, do, dr = F.rpn_loss(e(3, 37, 62, 9), e(3, 37, 62, 36), out[4], out[0], out[5])
        assert l.shape == (3,) and dr.shape == (3, 37, 62, 9, 4)


@pytest.mark.gpu
def test_operators_match_the_ctypes_front_end_and_carry_the_gradient(dev):
    from objectdetection_amd import ops, torch_ops  # noqa: F401
    F = torch.ops.frcnn
    gen = torch.Generator(device=dev).manual_seed(5)
    B, H, W, C, R = 2, 12, 17, 64, 40
    fmap = torch.randn((B, H, W, C), generator=gen, device=dev)
    y1 = torch.rand((B, R), generator=gen, device=dev) * 0.6
    x1 = torch.rand((B, R), generator=gen, device=dev) * 0.6
    rois = torch.stack([y1, x1, y1 + 0.35, x1 + 0.3], -1).contiguous()
    ref_out, ref_win = ops.roi_crop_pool_fwd(fmap, rois, (14, 14), 2, want_winner=True)
    f = fmap.clone().requires_grad_(True)
    out, win = F.roi_crop_pool(f, rois, 14, 14, 2)
    assert torch.equal(out, ref_out) and torch.equal(win, ref_win)
    w = torch.randn(out.shape, generator=gen, device=dev)
    (out * w).sum().backward()
    assert torch.equal(f.grad, ops.roi_crop_pool_bwd(w, ref_win, rois, tuple(fmap.shape), (14, 14), 2))
    # the dispatcher-level checks PyTorch ships for custom operators
    torch.library.opcheck(F.roi_crop_pool.default, (fmap.clone().requires_grad_(True), rois, 14, 14, 2),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    boxes = rois[:, :, :].contiguous()
    torch.library.opcheck(F.nms.default, (boxes, 0.7, 16), test_utils=("test_schema", "test_faketensor"))
    p, k, n = F.nms(boxes, 0.7, 16)
    rp, rk, rn = ops.nms(boxes, 0.7, 16)
    assert torch.equal(p, rp) and torch.equal(k, rk) and torch.equal(n, rn)

    # traced by torch.compile without a graph break (the operators are opaque nodes)
    @torch.compile(backend="eager", fullgraph=True)
    def fn(fm, r):
        o, _ = torch.ops.frcnn.roi_crop_pool(fm, r, 14, 14, 2)
        return o * 2.0

    assert torch.equal(fn(fmap, rois), ref_out * 2.0)
