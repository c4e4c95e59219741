"""Two-deep software pipelining of consecutive batches gives the results of the plain path,
batch by batch, with no host synchronisation between the calls."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("top_idx", "cand", "proposed", "keep_pos", "keep_count", "tindex", "gt_coord", "p_label")


def test_depth2_equals_depth1(dev):
    from objectdetection_amd.config import PathConfig
    from objectdetection_amd.pipeline import RegionProposalPath
    g = np.load(os.path.join(GOLD, "ref_small.npz"), allow_pickle=False)
    cfg = PathConfig(H=int(g["H"]), W=int(g["W"]), scales=tuple(g["scales"].tolist()),
                     G=int(g["G"]), pre_k=int(g["pre_k"]), post_n=int(g["post_n"]))
    B, C = g["gt"].shape[0], int(g["C"])
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    gt, label = torch.from_numpy(g["gt"]).to(dev), torch.from_numpy(g["label"]).to(dev)
    batches = []
    for i in range(5):
        fmap = torch.rand((B, cfg.H, cfg.W, C), generator=gen, device=dev)
        score = torch.rand((B, cfg.H, cfg.W, cfg.A), generator=gen, device=dev)
        coord = torch.randn((B, cfg.H, cfg.W, 4 * cfg.A), generator=gen, device=dev) * 0.5
        batches.append((fmap, score, coord))
    plain = RegionProposalPath(cfg, B, C, dev, train=True)
    want = []
    for i, (f, s, c) in enumerate(batches):
        plain.run(f, s, c, gt, label, seed=i)
        torch.cuda.synchronize()
        want.append({k: getattr(plain, k).clone() for k in KEYS} | {"pooled": plain.pooled.clone(),
                                                                   "rpn_pos": plain.rpn["pos"].clone()})
    piped = RegionProposalPath(cfg, B, C, dev, train=True, depth=2)
    got = []
    for i, (f, s, c) in enumerate(batches):                  # no synchronisation in between
        piped.run(f, s, c, gt, label, seed=i)
        got.append({k: getattr(piped, k).clone() for k in KEYS} | {"pooled": piped.pooled.clone(),
                                                                  "rpn_pos": piped.rpn["pos"].clone()})
    torch.cuda.synchronize()
    for w, h in zip(want, got):
        for k in w:
            assert torch.equal(w[k], h[k]), k


@pytest.mark.parametrize("train", [False, True])
def test_graph_replay_equals_run(dev, train):
    """The path recorded into a HIP graph: a replay on new contents of the input tensors gives
    what run() gives on them (the training variant forks to and joins the side stream inside
    the capture)."""
    from objectdetection_amd.config import PathConfig
    from objectdetection_amd.pipeline import RegionProposalPath
    g = np.load(os.path.join(GOLD, "ref_small.npz"), allow_pickle=False)
    cfg = PathConfig(H=int(g["H"]), W=int(g["W"]), scales=tuple(g["scales"].tolist()),
                     G=int(g["G"]), pre_k=int(g["pre_k"]), post_n=int(g["post_n"]))
    B, C = g["gt"].shape[0], int(g["C"])
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    gt, label = torch.from_numpy(g["gt"]).to(dev), torch.from_numpy(g["label"]).to(dev)
    fmap = torch.rand((B, cfg.H, cfg.W, C), generator=gen, device=dev)
    score = torch.rand((B, cfg.H, cfg.W, cfg.A), generator=gen, device=dev)
    coord = torch.randn((B, cfg.H, cfg.W, 4 * cfg.A), generator=gen, device=dev) * 0.5
    path = RegionProposalPath(cfg, B, C, dev, train=train)
    args = (gt, label) if train else ()
    graph = path.capture(fmap, score, coord, *args, seed=5)
    keys = KEYS if train else KEYS[:5]
    for _ in range(2):
        fmap.copy_(torch.rand(fmap.shape, generator=gen, device=dev))
        score.copy_(torch.rand(score.shape, generator=gen, device=dev))
        coord.copy_(torch.randn(coord.shape, generator=gen, device=dev) * 0.5)
        graph.replay()
        torch.cuda.synchronize()
        got = {k: getattr(path, k).clone() for k in keys} | {"pooled": path.pooled.clone()}
        path.run(fmap, score, coord, *args, seed=5)
        torch.cuda.synchronize()
        for k, v in got.items():
            assert torch.equal(v, getattr(path, k)), k


def test_inputs_ready_from_another_stream_depth1(dev):
    """Inputs produced on a third stream: with depth 1 the caller's stream itself waits for the
    `inputs_ready` event before the proposal stage reads them."""
    from objectdetection_amd.config import PathConfig
    from objectdetection_amd.pipeline import RegionProposalPath
    g = np.load(os.path.join(GOLD, "ref_small.npz"), allow_pickle=False)
    cfg = PathConfig(H=int(g["H"]), W=int(g["W"]), scales=tuple(g["scales"].tolist()),
                     G=int(g["G"]), pre_k=int(g["pre_k"]), post_n=int(g["post_n"]))
    B, C = g["gt"].shape[0], int(g["C"])
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    src = [torch.rand((B, cfg.H, cfg.W, C), generator=gen, device=dev),
           torch.rand((B, cfg.H, cfg.W, cfg.A), generator=gen, device=dev),
           torch.randn((B, cfg.H, cfg.W, 4 * cfg.A), generator=gen, device=dev) * 0.5]
    path = RegionProposalPath(cfg, B, C, dev, train=False)
    path.run(*src)
    torch.cuda.synchronize()
    want = {k: getattr(path, k).clone() for k in KEYS[:5]} | {"pooled": path.pooled.clone()}
    ins = [torch.zeros_like(t) for t in src]
    big = torch.empty(64 << 20, device=dev)
    torch.cuda.synchronize()
    producer = torch.cuda.Stream(dev)
    ev = torch.cuda.Event()
    with torch.cuda.stream(producer):
        for _ in range(20):                                   # keeps the producer busy for a while
            big.mul_(1.0001)
        for t, s in zip(ins, src):
            t.copy_(s)
        ev.record(producer)
    path.run(*ins, inputs_ready=ev)
    torch.cuda.synchronize()
    for k, v in want.items():
        assert torch.equal(v, getattr(path, k)), k


def test_planar_feature_map_gives_the_same_path(dev):
    """A backbone that emits [B,C,H,W] maps: the path transposes them into its workspace and
    gives the channels-last run's results bit for bit."""
    from objectdetection_amd.config import PathConfig
    from objectdetection_amd.pipeline import RegionProposalPath
    g = np.load(os.path.join(GOLD, "ref_small.npz"), allow_pickle=False)
    cfg = PathConfig(H=int(g["H"]), W=int(g["W"]), scales=tuple(g["scales"].tolist()),
                     G=int(g["G"]), pre_k=int(g["pre_k"]), post_n=int(g["post_n"]))
    B, C = g["gt"].shape[0], 64
    gen = torch.Generator(device=dev)
    gen.manual_seed(6)
    fmap = torch.rand((B, cfg.H, cfg.W, C), generator=gen, device=dev)
    score = torch.rand((B, cfg.H, cfg.W, cfg.A), generator=gen, device=dev)
    coord = torch.randn((B, cfg.H, cfg.W, 4 * cfg.A), generator=gen, device=dev) * 0.5
    a = RegionProposalPath(cfg, B, C, dev, train=False, want_winner=True)
    a.run(fmap, score, coord)
    planar = fmap.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
    b = RegionProposalPath(cfg, B, C, dev, train=False, want_winner=True, fmap_planar=True)
    b.run(planar, score, coord)
    torch.cuda.synchronize()
    assert torch.equal(a.pooled, b.pooled) and torch.equal(a.winner, b.winner)


def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 --backend gloo` from a bare shell on a one-GPU box: the script
    spawns two ranks that share the card, each runs RegionProposalPath on its half of a global
    batch of 32 images (16 per rank: the band-resident RoI kernel), rank 0 reports the whole job
    and checks its last step against a plain rerun."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend",
                          "gloo", "--global-batch", "32", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["pipelining_verified"] is True
    assert rec["oracle_verified"] is True
    assert rec["config"]["global_batch"] == 32 and rec["config"]["images_per_gpu_per_step"] == 16
    assert rec["roofline"]["kernel"] == "roi_fwd_band_kernel<2,false>"


def test_reference_named_functions_run_on_the_kept_buffer_sets(dev):
    """making_frcnn_input / process_fmap at the benchmark size (64 images, 37x62x512, 12k -> 2k)
    through the reference's names: the calls run on a kept RegionProposalPath (band-resident RoI
    kernel, radix-sort top-k, workspaces; no per-call allocation of the big tensors) and return
    what a RegionProposalPath built by hand returns, bit for bit; a second call reuses the set."""
    import bench
    from objectdetection_amd import ops, pipeline
    from objectdetection_amd.module import making_frcnn_input, process_fmap
    from objectdetection_amd.pipeline import RegionProposalPath
    cfg = bench.workload_cfg("600x1000")
    B, C = 64, 512
    f, s, c, g, l = bench.make_inputs(cfg, B, C, dev, seed=808)
    path = RegionProposalPath(cfg, B, C, dev, train=True, depth=1)
    path.run(f, s, c, g, l, seed=3)
    torch.cuda.synchronize()
    pipeline.clear_path_cache()
    for rep in range(2):
        before = torch.cuda.memory_allocated(dev)
        crop, gl, gm, gc, proposed, tindex = making_frcnn_input(g, l, f, c, s, path.anchor, cfg,
                                                                seed=3, fuse_pool=True)
        torch.cuda.synchronize()
        if rep == 1:                     # nothing big is allocated by a repeated call
            assert torch.cuda.memory_allocated(dev) - before < 64 << 20
        assert crop.shape == (B, cfg.post_n, 7, 7, C)
        assert torch.equal(proposed, path.proposed) and torch.equal(tindex, path.tindex)
        assert torch.equal(crop, path.pooled)
        ti = path.tindex
        assert torch.equal(gm, torch.gather(path.gt_mask, 1, ti))
        assert torch.equal(gl, torch.gather(path.gt_onehot, 1, ti.unsqueeze(-1).expand(-1, -1, 21)))
    assert len(pipeline._PATH_CACHE) == 1
    kept = next(iter(pipeline._PATH_CACHE.values()))
    assert kept.topk == "sort" and not kept.rpn_targets
    assert ops.roi_fwd_kernel_name(B, cfg.H, cfg.W, C, cfg.post_n, cfg.crop, cfg.pool).startswith(
        "roi_fwd_band")
    crop2, prop2 = process_fmap(f, s, c, path.anchor, cfg, fuse_pool=True)
    torch.cuda.synchronize()
    assert torch.equal(prop2, path.proposed) and torch.equal(crop2, path.pooled)
    assert len(pipeline._PATH_CACHE) == 2              # the inference set is a second entry
    pipeline.clear_path_cache()


@pytest.mark.parametrize("B", [1, 2])
def test_single_image_path_uses_the_one_launch_nms(dev, B):
    """One 600x1000 image (the reference's inference call, model_train.py:741-743): the path
    picks frcnn_nms_fused for one image, the chunked NMS from two on; both routes give the same
    proposals, keep lists and pooled tensor, also as one HIP graph replay."""
    import bench
    from objectdetection_amd.pipeline import RegionProposalPath
    cfg = bench.workload_cfg("600x1000")
    C = 64
    f, s, c, _, _ = bench.make_inputs(cfg, B, C, dev, seed=31 + B)
    auto = RegionProposalPath(cfg, B, C, dev, train=False)
    assert auto.nms == ("fused" if B == 1 else "chunked")
    a = RegionProposalPath(cfg, B, C, dev, train=False, nms="fused")
    b = RegionProposalPath(cfg, B, C, dev, train=False, nms="chunked")
    a.run(f, s, c)
    b.run(f, s, c)
    torch.cuda.synchronize()
    for k in ("proposed", "keep_pos", "keep_count", "pooled"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert int(a.keep_count.min()) > 0
    g = a.capture(f, s, c)
    a.proposed.zero_(); a.keep_pos.zero_(); a.pooled.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for k in ("proposed", "keep_pos", "keep_count", "pooled"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
