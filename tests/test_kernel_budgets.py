"""Register, LDS and scratch budgets of the hot kernels, read from the built library's code-object
metadata (no GPU needed).

Several kernels are written around a budget the compiler must keep: the band-resident RoI
forward pins its staging registers to v[64:95] and needs 96 VGPRs (five waves per SIMD leave one
wave slot and 23 KB of LDS to the other streams' kernels), the NMS mask kernel is capped at 64
VGPRs so that two of its waves fit beside a RoI wave, the scan at 128.  A compiler update that
spills or re-allocates would otherwise show up only as a slowdown on the GPU box."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def budgets():
    from kernel_budgets import kernel_budgets
    b = kernel_budgets()
    assert len(b) > 40
    return b


def test_band_forward_keeps_its_register_budget(budgets):
    for name in ("roi_fwd_band_kernel<2,false>", "roi_fwd_band_kernel<2,true>",
                 "roi_fwd_band_kernel<1,false>"):
        k = budgets[name]
        assert k["vgpr"] <= 96 and k["agpr"] == 0, (name, k)
        assert k["lds"] == 0, (name, k)                  # all of its LDS is the dynamic band block
    assert budgets["roi_fwd_band_kernel<2,false>"]["scratch"] == 0
    assert budgets["roi_fwd_band_kernel<1,false>"]["scratch"] == 0
    # the winner variant parks the four winner bytes of a lane in 20 bytes of scratch
    assert budgets["roi_fwd_band_kernel<2,true>"]["scratch"] <= 20


def test_per_roi_forward_budget(budgets):
    for name in ("roi_fwd_stage_kernel<2,false>", "roi_fwd_stage_kernel<1,false>"):
        k = budgets[name]
        assert k["vgpr"] <= 128 and k["scratch"] == 0 and k["lds"] <= 8192, (name, k)
    assert budgets["roi_fwd_stage_kernel<2,true>"]["vgpr"] <= 128


def test_nms_kernels_fit_beside_a_roi_wave(budgets):
    m, s = budgets["nms_mask_kernel"], budgets["nms_scan_kernel"]
    assert m["vgpr"] <= 64 and m["scratch"] <= 32 and m["lds"] <= 2048, m
    assert s["vgpr"] <= 128 and s["scratch"] == 0, s


def test_backward_kernels_do_not_spill(budgets):
    # the variants frcnn_roi_crop_pool_bwd launches by default (two cells per wave)
    for name, k in budgets.items():
        if name.startswith("roi_bwd_wave_kernel<") and name.endswith(",2>"):
            assert k["scratch"] == 0, (name, k)
    # C = 512: two channel quads per lane, crops up to 16 x 16
    assert budgets["roi_bwd_wave_kernel<2,2,16,2>"]["vgpr"] <= 128
    # patch-owner backward: the sums sit in pinned registers (v64 up), nothing may spill, and
    # the two default shapes must leave room for three waves per SIMD
    patch = {n: k for n, k in budgets.items() if n.startswith("roi_bwd_patch_kernel<")}
    assert patch, "roi_bwd_patch_kernel is missing from the library"
    for name, k in patch.items():
        assert k["scratch"] == 0 and k["lds"] == 0, (name, k)
    for name in ("roi_bwd_patch_kernel<4,4,3>", "roi_bwd_patch_kernel<3,4,3>"):
        assert patch[name]["vgpr"] <= 168, (name, patch[name])


def test_target_kernels_fit_beside_a_band_workgroup(budgets):
    # 256-thread samplers: one wave per SIMD and an LDS block that fits the 23 KB a band
    # workgroup leaves free on its compute unit
    for name in ("rpn_sample_kernel<256,256>", "sample_proposals_kernel<256,256>"):
        k = budgets[name]
        assert k["vgpr"] <= 128 and k["lds"] <= 9 * 1024 and k["scratch"] == 0, (name, k)
    # the two-launch top-k: 256-thread workgroups that fit the same 128 registers per lane and
    # 23 KB of LDS
    for name in ("tm_runs_kernel", "tm_rank_kernel"):
        k = budgets[name]
        assert k["vgpr"] <= 128 and k["lds"] <= 17 * 1024 and k["scratch"] == 0, (name, k)
    for name in ("iou_assign_kernel", "label_proposals_kernel", "ts_hist_kernel<0>",
                 "ts_scatter_kernel<0>", "ts_scatter_kernel<3>"):
        k = budgets[name]
        assert k["vgpr"] <= 64 and k["scratch"] == 0, (name, k)
