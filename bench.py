#!/usr/bin/env python3
"""Headline benchmark: images/sec of the fused RPN path (anchor match + proposal NMS + RoI
crop/pool) on synthetic images.

  python bench.py --gpus N --steps K --warmup W [--config 600x1000|ship768] [--global-batch G]
                  [--mode path|train]

One step = one pass of the whole hot path (RegionProposalPath.run, training mode) over the
images a rank owns, inputs already resident in HBM.  Images are independent units: rank r works
on its own images, no data-path collective; the barrier and the max-over-ranks time are the only
communication (`--mode train` adds the backbone / head stand-ins and the gradient all-reduce).
`--batch B` fixes the images per GPU (weak scaling, the default); `--global-batch G` fixes the
total and splits it over the ranks (strong scaling: "64 images sharded over 1/2/4/8 GPUs").

With --gpus N > 1 the script expects to run under torch.distributed.run (one rank per GPU).
Started without a launcher it starts that launcher itself, before anything touches the GPU,
and relays rank 0's line.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
F32_MFMA_PEAK_TFLOPS = 157.3   # dense float32 matrix peak (v_mfma_f32_32x32x2_f32)

WORKLOADS = {
    # 600x1000 image, VGG16 stride 16 -> 37x62 map, 9 anchors (20,646), G=42, 12k -> 2k NMS at
    # 0.7, RoI 14 -> 7, C=512
    "600x1000": dict(cfg=dict(H=37, W=62, scales=(128, 256, 512), ratios=(0.5, 1, 2),
                              base_w=1000.0, base_h=600.0, G=42, pre_k=12000, post_n=2000,
                              nms_thr=0.7),
                     batch=64, gt_side=(0.05, 0.6),
                     text="600x1000 image -> 37x62x9 = 20,646 anchors, G=42, top-12000 -> NMS 0.7 "
                          "-> 2000, RoI crop 14 -> pool 7, C=512"),
    # ship detector (shipdetection/train.py:129-143): 768x768 -> 48x48 map, 12 anchors
    # (scales 32..256, still divided by 500), up to 50 small boxes, 6k -> 1.5k
    "ship768": dict(cfg=dict(H=48, W=48, scales=(32, 64, 128, 256), ratios=(0.5, 1, 2),
                             base_w=500.0, base_h=500.0, G=50, pre_k=6000, post_n=1500,
                             nms_thr=0.7),
                    batch=16, gt_side=(0.01, 0.06),
                    text="768x768 ship image -> 48x48x12 = 27,648 anchors, G=50 small boxes, "
                         "top-6000 -> NMS 0.7 -> 1500, RoI crop 14 -> pool 7, C=512"),
}


def workload_cfg(name="600x1000"):
    from objectdetection_amd.config import PathConfig
    return PathConfig(**WORKLOADS[name]["cfg"])


def make_inputs(cfg, B, C, device, seed, gt_side=(0.05, 0.6)):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    score = torch.sigmoid(torch.randn((B, cfg.H, cfg.W, cfg.A), generator=g, device=device) * 2.0)
    coord = torch.randn((B, cfg.H, cfg.W, 4 * cfg.A), generator=g, device=device) * 0.5
    fmap = torch.relu(torch.randn((B, cfg.H, cfg.W, C), generator=g, device=device))
    rng = np.random.Generator(np.random.PCG64(seed))
    gt = np.zeros((B, cfg.G, 4), np.float32)
    label = np.zeros((B, cfg.G), np.int64)
    lo, hi = np.log(gt_side[0]), np.log(gt_side[1])
    for b in range(B):
        n = int(rng.integers(1, cfg.G + 1))
        cy, cx = rng.uniform(0.05, 0.95, n), rng.uniform(0.05, 0.95, n)
        hh = np.exp(rng.uniform(lo, hi, n))
        ww = np.exp(rng.uniform(lo, hi, n))
        gt[b, :n] = np.clip(np.stack([cy - hh / 2, cx - ww / 2, cy + hh / 2, cx + ww / 2], 1), 0, 1)
        label[b, :n] = rng.integers(0, 20, n)
    return (fmap.contiguous(), score.contiguous(), coord.contiguous(),
            torch.from_numpy(gt).to(device), torch.from_numpy(label).to(device))


def clustered_boxes(B, K, device, seed):
    """Score-ordered candidates in a few tight clusters per image: greedy NMS at 0.7 keeps far
    fewer than post_n of them, so every chunk of the candidate list is walked."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nc = 40
    centre = rng.uniform(0.2, 0.8, (B, nc, 2))
    half = rng.uniform(0.05, 0.2, (B, nc, 2))
    which = rng.integers(0, nc, (B, K))
    c = np.take_along_axis(centre, which[..., None], 1) + rng.normal(0, 0.004, (B, K, 2))
    h = np.take_along_axis(half, which[..., None], 1) * (1 + rng.normal(0, 0.02, (B, K, 2)))
    b = np.clip(np.concatenate([c - h, c + h], -1), 0, 1).astype(np.float32)
    return torch.from_numpy(b).to(device)


def algorithmic_bytes_roi(cfg, C, B):
    """RoI kernel, per launch: feature map read once + pooled tensor written once."""
    ph, pw = cfg.crop[0] // cfg.pool, cfg.crop[1] // cfg.pool
    per_image = cfg.H * cfg.W * C * 4 + cfg.post_n * ph * pw * C * 4 + cfg.post_n * 16
    return per_image * B


def sources_sha():
    """Hash of the kernel sources the loaded library was built from (they travel with the tree):
    counter-derived records under profiles/ carry the hash of the sources they were collected on
    and are only quoted when it matches."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "objectdetection_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "objectdetection_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def profile_record(name):
    """(record, note): a counter record under profiles/ if it was collected on these very kernel
    sources, else (None, why not)."""
    f = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(f):
        return None, f"profiles/{name} is missing"
    with open(f) as fh:
        rec = json.load(fh)
    have, want = rec.get("csrc_sha"), sources_sha()
    if have != want:
        return None, (f"profiles/{name} was collected on kernel sources {have}, this tree is "
                      f"{want}: not quoted")
    return rec, (f"profiles/{name}: collected by the builder with rocprofv3 --pmc on kernel sources "
                 f"{have} (commit {rec.get('commit', '?')}), not measured in this run")


def verify_against_oracle(path, cfg, fmap, image):
    """Outside the timed region: one image of the last timed step against the CPU oracle (C
    restatement): its keep list and proposals from the device's own candidate boxes, and its
    whole pooled tensor from those proposals.  Bit-exact or SystemExit."""
    from oracle import c_oracle
    lib = c_oracle.load()
    torch.cuda.synchronize()
    cand = path.cand[image].cpu().numpy()
    rp, rk, rc = c_oracle.nms(lib, cand, cfg.nms_thr, cfg.post_n)
    if not (np.array_equal(path.keep_pos[image].cpu().numpy(), rk)
            and np.array_equal(path.proposed[image].cpu().numpy(), rp)
            and int(path.keep_count[image]) == int(rc)):
        raise SystemExit(f"bench: keep list of image {image} differs from the oracle's greedy NMS")
    ro, _ = c_oracle.roi_crop_pool(lib, fmap[image].cpu().numpy(), rp, cfg.crop, cfg.pool,
                                   want_winner=False)
    if not np.array_equal(path.pooled[image].cpu().numpy(), ro):
        raise SystemExit(f"bench: pooled tensor of image {image} differs from the oracle's crop + pool")
    return True


def single_image(cfg, C, device, seed=4242):
    """One image through the inference stages (the reference's per-epoch check runs one image,
    model_train.py:741-743): whole-path latency as plain launches and as one HIP graph replay,
    and the stages alone.  HIP events, idle chip, outside the timed region."""
    from objectdetection_amd import _lib, ops
    from objectdetection_amd.pipeline import RegionProposalPath
    fmap, score, coord, _, _ = make_inputs(cfg, 1, C, device, seed=seed)
    path = RegionProposalPath(cfg, 1, C, device, train=False, depth=1)
    run = lambda: path.run(fmap, score, coord)
    plain = timed_alone(run, device, reps=40, skip=10)
    graph = path.capture(fmap, score, coord)
    replay = timed_alone(graph.replay, device, reps=40, skip=10)
    a = path.anchor.reshape(-1, 4)
    topk_ms = timed_alone(lambda: ops.topk(score.reshape(1, -1), cfg.pre_k, a, coord, "sort",
                                           workspace=path.topk_ws), device, reps=40, skip=10)
    _lib.set_tuning("topk_merge_max_runs", 0)
    radix_ms = timed_alone(lambda: ops.topk(score.reshape(1, -1), cfg.pre_k, a, coord, "sort",
                                            workspace=path.topk_ws), device, reps=40, skip=10)
    _lib.set_tuning("topk_merge_max_runs", None)
    nms_call = ops.nms_fused if path.nms == "fused" else ops.nms
    nms_ms = timed_alone(lambda: nms_call(path.cand, cfg.nms_thr, cfg.post_n, workspace=path.nms_ws),
                         device, reps=40, skip=10)
    chunked_ms = timed_alone(lambda: ops.nms(path.cand, cfg.nms_thr, cfg.post_n), device, reps=40,
                             skip=10)
    roi_ms = timed_alone(lambda: ops.roi_crop_pool_fwd(fmap, path.proposed, cfg.crop, cfg.pool,
                                                       out=path.pooled, workspace=path.roi_ws),
                         device, reps=40, skip=10)
    return dict(latency_b1_ms=min(plain, replay), latency_b1_plain_launches_ms=plain,
                latency_b1_graph_replay_ms=replay, topk_us_single_image=topk_ms * 1e3,
                topk_route_single_image=ops.topk_sort_route(1, score.numel(), path.topk_ws.numel()),
                topk_us_single_image_radix_passes=radix_ms * 1e3,
                nms_us_single_image=nms_ms * 1e3, nms_route_single_image=path.nms,
                nms_us_single_image_chunked=chunked_ms * 1e3, roi_us_single_image=roi_ms * 1e3,
                kept=int(path.keep_count[0]),
                note="one 600x1000 image, inference stages (top-k, NMS, RoI crop + pool), idle chip")


def training_side(cfg, C, device, B=8, seed=5151):
    """The kernels a training step adds behind the forward path, alone on an idle chip at the
    per-GPU batch of the training configuration (8 images): RoI forward with the winner index,
    RoI backward (255.6 MB algorithmic per image: pooled gradient + winner read, feature-map
    gradient written) and the head's first dense layer on the pooled tensor (float32 MFMA)."""
    from objectdetection_amd import ops
    from objectdetection_amd.pipeline import RegionProposalPath
    fmap, score, coord, _, _ = make_inputs(cfg, B, C, device, seed=seed)
    path = RegionProposalPath(cfg, B, C, device, train=False, want_winner=True, depth=1)
    path.run(fmap, score, coord)
    fwd_ms = timed_alone(lambda: ops.roi_crop_pool_fwd(fmap, path.proposed, cfg.crop, cfg.pool,
                                                       out=path.pooled, winner=path.winner,
                                                       workspace=path.roi_ws), device, reps=12, skip=4)
    dY = torch.randn_like(path.pooled)
    dF = torch.empty_like(fmap)
    bwd_ms = timed_alone(lambda: ops.roi_crop_pool_bwd(dY, path.winner, path.proposed, tuple(fmap.shape),
                                                       cfg.crop, cfg.pool, out=dF), device, reps=8, skip=2)
    bwd_bytes = dY.numel() * 4 + path.winner.numel() + dF.numel() * 4
    del dY
    units = 4096
    Kf = path.pooled[0, 0].numel()
    w = torch.randn((Kf, units), device=device) * 0.01
    bias = torch.randn((units,), device=device)
    out = torch.empty((B * cfg.post_n, units), device=device)
    fc_ms = timed_alone(lambda: ops.fc1_fwd(path.pooled, w, bias, True, out=out), device, reps=6, skip=2)
    flops = 2.0 * B * cfg.post_n * Kf * units
    return dict(images=B,
                roi_fwd_with_winner_ms=fwd_ms,
                roi_bwd=dict(kernel="roi_bwd_patch_kernel", ms=bwd_ms, bytes=int(bwd_bytes), bound="hbm",
                             achieved=bwd_bytes / (bwd_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=bwd_bytes / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             note="vector-ALU bound by the defined summation (profiles/r03_pmc_roi_bwd.txt)"),
                fc1=dict(kernel="fc1_fwd_kernel", shape=[B * cfg.post_n, Kf, units], ms=fc_ms, bound="mfma",
                         achieved=flops / (fc_ms * 1e-3) / 1e12, peak=F32_MFMA_PEAK_TFLOPS,
                         unit="TFLOP/s", frac=flops / (fc_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                         dtype="f32"),
                note="alone on an idle chip, outside the timed region; not part of `value`")


def reference_api_step(cfg, B, C, device, fmap, score, coord, gt, label, anchor, steps, warmup):
    """The same step through the reference's function names (patch_batch for the batch +
    making_frcnn_input, objectdetection_amd/module): wall time per step with a host sync only at
    both ends."""
    from objectdetection_amd import pipeline
    from objectdetection_amd.module import data as D
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(device))

    def step(k):
        D.patch_batch_many(gt, anchor, cfg, seed=k, image_base=0)
        D.making_frcnn_input(gt, label, fmap, coord, score, anchor, cfg, seed=k, fuse_pool=True,
                             check=False, inputs_ready=ready)

    for w in range(warmup):
        step(w)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(steps):
        step(100 + k)
    torch.cuda.synchronize(device)
    ms = (time.perf_counter() - t0) / steps * 1e3
    pipeline.clear_path_cache()
    return ms


def secondary_workload(name, C, device, steps, warmup, rank):
    """A second workload's headline numbers (no CPU leg), same timing rules as the main line."""
    from objectdetection_amd.pipeline import RegionProposalPath
    wl, cfg = WORKLOADS[name], workload_cfg(name)
    B = wl["batch"]
    fmap, score, coord, gt, label = make_inputs(cfg, B, C, device, seed=3000 + rank,
                                                gt_side=wl["gt_side"])
    path = RegionProposalPath(cfg, B, C, device, train=True, depth=2)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(steps)]
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(device))
    for w in range(warmup):
        path.run(fmap, score, coord, gt, label, seed=w, inputs_ready=ready)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(steps):
        path.run(fmap, score, coord, gt, label, seed=100 + k, roi_events=ev[k], inputs_ready=ready)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    roi_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    ok = verify_against_oracle(path, cfg, fmap, B // 2)
    alg = algorithmic_bytes_roi(cfg, C, B)
    from objectdetection_amd import ops
    return {"metric": "images/sec (RPN anchor-match+proposal NMS+RoIPool) 768x768 ship"
            if name == "ship768" else name,
            "value": B * steps / elapsed, "unit": "images/s", "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "config": {"workload": wl["text"], "images_per_gpu_per_step": B,
                       "batches_in_flight": 2},
            "oracle_verified": ok,
            "roofline": {"kernel": ops.roi_fwd_kernel_name(B, cfg.H, cfg.W, C, cfg.post_n, cfg.crop,
                                                           cfg.pool),
                         "bound": "hbm", "achieved": alg / (roi_ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (roi_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "avg_launch_ms": roi_ms, "algorithmic_bytes_per_launch": alg}}


def verify_last_step(path, cfg, B, C, device, fmap, score, coord, gt, label, seed, image_base,
                     train, topk):
    """Outside the timed region: the last timed step's results (two batches in flight) must
    equal a plain one-batch-at-a-time rerun of the same step bit for bit.  Raises on a
    mismatch; returns True."""
    from objectdetection_amd.pipeline import RegionProposalPath
    torch.cuda.synchronize(device)
    last = {k: getattr(path, k).clone() for k in ("proposed", "keep_pos", "keep_count")}
    if train:
        last["tindex"] = path.tindex.clone()
    pooled = path.pooled[B // 2].clone()
    plain = RegionProposalPath(cfg, B, C, device, train=train, depth=1, topk=topk)
    plain.run(fmap, score, coord, gt, label, seed=seed, image_base=image_base)
    torch.cuda.synchronize(device)
    for k, v in last.items():
        if not torch.equal(v, getattr(plain, k)):
            raise SystemExit(f"bench: {k} of the last timed step differs from a plain rerun")
    if not torch.equal(pooled, plain.pooled[B // 2]):
        raise SystemExit("bench: pooled tensor of the last timed step differs from a plain rerun")
    del plain
    return True


def cpu_baseline(cfg, fmap, score, coord, gt, label, threads=16, target_s=12.0):
    """The CPU oracle on the same step: C restatement (oracle/frcnn_oracle_c.c, -O3, no
    FMA contraction) for IoU/argmax, top-k, decode, greedy NMS and crop+pool, numpy
    restatement for the two samplers and the proposal labels.  One image per task;
    `threads` host threads (the C calls release the GIL).  The batch is re-run until
    about `target_s` seconds of wall time have been measured."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle, frcnn_oracle as O
    lib = c_oracle.load()
    a = O.make_anchor(cfg.H, cfg.W, cfg.scales, cfg.ratios, cfg.base_w, cfg.base_h)
    n = fmap.shape[0]
    f, s, c = fmap.cpu().numpy(), score.cpu().numpy(), coord.cpu().numpy()
    g, l = gt.cpu().numpy(), label.cpu().numpy()

    def one(i):
        c_oracle.iou_argmax(lib, a, g[i])
        O.patch_batch(dict(image=None, bbox=g[i], label=l[i]), a, seed=0, image_index=i,
                      batch=cfg.rpn_batch, max_pos=cfg.rpn_max_pos)
        _, idx = c_oracle.topk(lib, s[i].reshape(-1), cfg.pre_k)
        cand = c_oracle.decode_gather(lib, a, c[i], idx)
        prop, _, _ = c_oracle.nms(lib, cand, cfg.nms_thr, cfg.post_n)
        c_oracle.roi_crop_pool(lib, f[i], prop, cfg.crop, cfg.pool, want_winner=False)
        lab = O.label_proposals(prop[None], g[i:i + 1], l[i:i + 1], cfg.fg_thr, cfg.iou_floor,
                                cfg.bg_label, cfg.num_classes)
        O.sample_proposals(lab["p_iou"], seed=0, image_base=i, batch=cfg.head_batch,
                           max_pos=cfg.head_max_pos, bg_label=cfg.bg_label)

    t0 = time.perf_counter()
    for i in range(min(4, n)):
        one(i)
    t1 = (time.perf_counter() - t0) / min(4, n)
    threads = max(1, min(threads, os.cpu_count() or 1, n))
    passes = max(1, int(round(target_s / max(t1 * n / threads, 1e-3))))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(one, [i % n for i in range(n * passes)]))
    dt = time.perf_counter() - t0
    return dict(value=n * passes / dt, unit="images/s", cores=threads, kind="port",
                sample=f"{n * passes} images ({passes} passes over the step's {n}-image batch, one "
                       f"image per task) on {threads} host threads in {dt:.2f} s; single thread "
                       f"{1.0 / t1:.2f} images/s; C -O3 restatement of the oracle; host reports "
                       f"{os.cpu_count()} logical CPUs")


def cpu_baseline_numpy(cfg, fmap, score, coord, gt, label, target_s=12.0):
    """The op-for-op numpy restatement (oracle/frcnn_oracle.py: the reference's broadcast
    formulation, sequential greedy NMS, crop then pool) on whole images, one host thread,
    until about `target_s` seconds have been spent (at least one image)."""
    from oracle import frcnn_oracle as O
    a = O.make_anchor(cfg.H, cfg.W, cfg.scales, cfg.ratios, cfg.base_w, cfg.base_h)
    f, s, c = fmap.cpu().numpy(), score.cpu().numpy(), coord.cpu().numpy()
    g, l = gt.cpu().numpy(), label.cpu().numpy()
    n, done = fmap.shape[0], 0
    t0 = time.perf_counter()
    while done < n and (done == 0 or time.perf_counter() - t0 < target_s):
        i = done
        O.patch_batch(dict(image=None, bbox=g[i], label=l[i]), a, seed=0, image_index=i,
                      batch=cfg.rpn_batch, max_pos=cfg.rpn_max_pos)
        pc = O.proposal_creator(s[i:i + 1], c[i:i + 1], a, cfg.pre_k, cfg.post_n, cfg.nms_thr)
        O.roi_crop_pool(f[i:i + 1], pc["proposed"], cfg.crop, cfg.pool)
        lab = O.label_proposals(pc["proposed"], g[i:i + 1], l[i:i + 1], cfg.fg_thr, cfg.iou_floor,
                                cfg.bg_label, cfg.num_classes)
        O.sample_proposals(lab["p_iou"], seed=0, image_base=i, batch=cfg.head_batch,
                           max_pos=cfg.head_max_pos, bg_label=cfg.bg_label)
        done += 1
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="images/s", cores=1, kind="port",
                sample=f"{done} images in {dt:.2f} s on one host thread; numpy restatement of the "
                       f"reference's formulation (oracle/frcnn_oracle.py)")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(n, argv):
    """The launch the driver uses for N > 1, for a bare `python bench.py --gpus N ...`."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
            os.path.abspath(__file__)] + list(argv)


def timed_alone(fn, device, reps=5, skip=1):
    """Mean HIP-event time of fn() alone on the chip, the first `skip` calls left out (a long
    kernel needs a few launches before the clocks settle after a quiet phase)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize(device)
        ts.append(e0.elapsed_time(e1))
    return float(np.mean(ts[skip:]))


def kernel_table(path, cfg, B, C, fmap, score, coord, gt, label, device):
    """Every stage alone on an idle chip (outside the timed region), with its algorithmic
    bytes (per image: anchors N*16, GT G*16, assignment out N*9, deltas N*16, scores N*4,
    top-k out K*8 + K*16, feature map H*W*C*4, pooled R*49*C*4) and the fraction of the HBM
    peak those bytes over that time amount to."""
    from objectdetection_amd import ops
    N, K, R, G = cfg.N, cfg.pre_k, cfg.post_n, gt.shape[1]
    a = path.anchor.reshape(-1, 4)
    rows = []

    def add(name, fn, nbytes, **kw):
        ms = timed_alone(fn, device, **kw)
        rows.append(dict(name=name, ms_alone=ms, bytes=int(nbytes),
                         frac=nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS))

    add("iou_assign_kernel", lambda: ops.iou_assign(a, gt, cfg.rpn_pos_thr, cfg.rpn_neg_thr),
        B * (N * 16 + G * 16 + N * 9))
    add("rpn_sample_kernel", lambda: ops.rpn_sample(path.anchor, gt, path.best_gt, path.flags,
                                                    seed=1, batch=cfg.rpn_batch,
                                                    max_pos=cfg.rpn_max_pos),
        B * (N * 5 + cfg.rpn_batch * 100))
    sort_ws = path.topk_ws if path.topk == "sort" else torch.empty(
        ops.topk_sort_workspace_bytes(B, N), dtype=torch.uint8, device=device)
    add("topk_sort (ts_* kernels)", lambda: ops.topk(score.reshape(B, -1), K, a, coord, "sort",
                                                     workspace=sort_ws),
        B * (N * 4 + K * 16 + K * 8 + K * 16))
    if K <= 16384:
        add("topk_kernel (LDS radix select)",
            lambda: ops.topk(score.reshape(B, -1), K, a, coord, "lds"),
            B * (N * 4 + K * 16 + K * 8 + K * 16))
    add("nms (mask + scan kernels)", lambda: ops.nms(path.cand, cfg.nms_thr, R, workspace=path.nms_ws),
        B * (K * 16 + R * 20))
    add(ops.roi_fwd_kernel_name(B, cfg.H, cfg.W, C, R, cfg.crop, cfg.pool),
        lambda: ops.roi_crop_pool_fwd(fmap, path.proposed, cfg.crop, cfg.pool, out=path.pooled,
                                      workspace=path.roi_ws),
        algorithmic_bytes_roi(cfg, C, B), reps=24, skip=8)
    if ops.roi_fwd_kernel_name(B, cfg.H, cfg.W, C, R, cfg.crop, cfg.pool).startswith("roi_fwd_band"):
        add("roi_fwd_stage_kernel<2,false> (per-RoI kernel, not in the step)",
            lambda: ops.roi_crop_pool_fwd(fmap, path.proposed, cfg.crop, cfg.pool, out=path.pooled),
            algorithmic_bytes_roi(cfg, C, B), reps=24, skip=8)
    add("label_proposals_kernel", lambda: ops.label_proposals(path.proposed, gt, label, cfg.fg_thr,
                                                              cfg.iou_floor, cfg.bg_label,
                                                              cfg.num_classes),
        B * (R * 16 + G * 24 + R * (8 + 4 + 4 * cfg.num_classes + 16 + 8)))
    add("sample_proposals_kernel", lambda: ops.sample_proposals(path.p_label, cfg.bg_label, seed=1,
                                                                batch=cfg.head_batch,
                                                                max_pos=cfg.head_max_pos),
        B * (R * 8 + cfg.head_batch * 8))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="600x1000", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="path", choices=("path", "train"),
                    help="path: the hot path alone; train: RPN + head training step around it "
                         "(stand-in backbone / head, RoI backward, gradient all-reduce)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="total images per step, split over the ranks (strong scaling)")
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras of the default run (second workload, single-image "
                         "latency, the step through the reference's function names)")
    ap.add_argument("--inference", action="store_true", help="skip the training-only stages")
    ap.add_argument("--depth", type=int, default=2, choices=(1, 2),
                    help="batches in flight: 2 overlaps the proposals of batch i+1 with the RoI "
                         "kernel of batch i")
    ap.add_argument("--topk", default="sort", choices=("sort", "lds"))
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend (gloo lets several ranks rehearse on one GPU)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="initialise the process group, reduce one number, print n_gpus and exit "
                         "(checks the launch path without a GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one (nothing has touched the GPU yet) and relay the ranks' output
        proc = subprocess.run(launcher_command(args.gpus, sys.argv[1:]))
        raise SystemExit(proc.returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl" and not args.rendezvous_only:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo" if args.rendezvous_only else args.backend, rank=rank,
                                    world_size=world)
    if args.rendezvous_only:
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rendezvous": "ok", "max_rank_plus_1": t.item()}),
                  flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    wl = WORKLOADS[args.config]
    cfg = workload_cfg(args.config)
    C = args.channels
    if args.global_batch is not None:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be a multiple of --gpus")
        B, scaling = args.global_batch // world, "strong"
    else:
        B, scaling = args.batch or wl["batch"], "weak"

    if args.mode == "train":
        from objectdetection_amd.train import bench_train
        bench_train(args, cfg, wl, B, C, device, dist, rank, world, scaling)
        return

    from objectdetection_amd.pipeline import RegionProposalPath
    fmap, score, coord, gt, label = make_inputs(cfg, B, C, device, seed=1000 + rank,
                                                gt_side=wl["gt_side"])
    path = RegionProposalPath(cfg, B, C, device, train=not args.inference, depth=args.depth,
                              topk=args.topk)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    ev_nms = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]

    def sync():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    ready = torch.cuda.Event()           # the synthetic inputs are resident before the loop
    ready.record(torch.cuda.current_stream(device))
    for w in range(args.warmup):
        path.run(fmap, score, coord, gt, label, seed=w, image_base=rank * B, inputs_ready=ready)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        path.run(fmap, score, coord, gt, label, seed=100 + k, image_base=rank * B,
                 roi_events=ev[k], inputs_ready=ready, nms_events=ev_nms[k])
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        roi_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        nms_busy_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_nms]))
        from objectdetection_amd import ops
        kept = path.keep_count.float().mean().item()
        oracle_ok = verify_against_oracle(path, cfg, fmap, B // 2)
        verified = verify_last_step(path, cfg, B, C, device, fmap, score, coord, gt, label,
                                    seed=100 + args.steps - 1, image_base=rank * B,
                                    train=not args.inference, topk=args.topk)
        # stages alone on an idle chip (outside the timed region)
        nms_ms = timed_alone(lambda: ops.nms(path.cand, cfg.nms_thr, cfg.post_n,
                                             workspace=path.nms_ws), device, reps=24, skip=8)
        cl = clustered_boxes(B, cfg.pre_k, device, seed=77)
        cl_out = ops.nms(cl, cfg.nms_thr, cfg.post_n, workspace=path.nms_ws)
        nms_cl_ms = timed_alone(lambda: ops.nms(cl, cfg.nms_thr, cfg.post_n,
                                                workspace=path.nms_ws), device, reps=24, skip=8)
        nms_cl_kept = cl_out[2].float().mean().item()
        roi_alone_ms = timed_alone(lambda: ops.roi_crop_pool_fwd(fmap, path.proposed, cfg.crop,
                                                                 cfg.pool, out=path.pooled,
                                                                 workspace=path.roi_ws),
                                   device, reps=24, skip=8)
        roi_kernel = ops.roi_fwd_kernel_name(B, cfg.H, cfg.W, C, cfg.post_n, cfg.crop, cfg.pool)
        kernels = kernel_table(path, cfg, B, C, fmap, score, coord, gt, label, device) \
            if not args.inference else None
        alg = algorithmic_bytes_roi(cfg, C, B)
        achieved = alg / (roi_ms * 1e-3) / 1e9
        # counter-derived numbers are NOT measured in this run: they come from records the
        # builder collected with rocprofv3 --pmc, quoted only if collected on these kernel sources
        traffic = bounds = lds_rates = None
        rec, traffic_src = profile_record("roi_fwd_traffic.json")
        if rec is not None:
            if rec.get("batch") == B and rec.get("channels") == C and args.config == rec.get("config"):
                krec = rec.get("kernels", {}).get(roi_kernel, {})
                traffic = krec.get("hbm_bytes_per_launch")
                bounds = krec.get("co_bounds")
            else:
                traffic_src = "profiles/roi_fwd_traffic.json is for another workload: not quoted"
        lrec, lds_src = profile_record("lds_bank_conflicts.json")
        if lrec is not None:
            lds_rates = lrec.get("rates")
        out = {
            "metric": "images/sec (RPN anchor-match+proposal NMS+RoIPool) "
                      + ("600x1000" if args.config == "600x1000" else "768x768 ship"),
            "value": world * B * args.steps / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "nms_us_per_image": nms_ms * 1e3 / B,
            "nms_us_per_image_note": f"one NMS call over the {B}-image batch, alone on the chip, "
                                     f"divided by {B} (amortised); nms_us_single_image is one "
                                     "image's own call",
            "nms_us_per_image_clustered": nms_cl_ms * 1e3 / B,
            "nms_us_per_image_in_step": nms_busy_ms * 1e3 / B,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl["text"] + "; "
                            + ("inference stages only" if args.inference else
                               "with RPN + head target sampling (training forward)"),
                "images_per_gpu_per_step": B,
                "global_batch": B * world,
                "batches_in_flight": args.depth,
                "mean_kept_proposals": kept,
                "clustered_nms_mean_kept": nms_cl_kept,
            },
            "pipelining_verified": verified,
            "pipelining_verified_note": "last timed step (two batches in flight) == plain "
                                        "one-batch rerun on the same kernels, bit for bit",
            "oracle_verified": oracle_ok,
            "oracle_verified_note": f"image {B // 2} of the last timed step: keep list, proposals "
                                    "and the whole pooled tensor == CPU oracle (C restatement), "
                                    "bit for bit",
            "kernel_sources_sha": sources_sha(),
            "roofline": {
                "kernel": roi_kernel,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_launch_ms": roi_ms,
                "avg_launch_ms_source": "HIP events around the kernel inside the timed loop (this run)",
                "avg_launch_ms_alone": roi_alone_ms,
                "frac_alone": alg / (roi_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": alg,
                "co_bounds": bounds,
            },
        }
        if kernels is not None:
            out["kernels"] = kernels
        out["lds_bank_conflict_rate"] = lds_rates
        out["lds_bank_conflict_rate_source"] = lds_src
        extras = not args.no_extras and world == 1 and args.config == "600x1000" \
            and not args.inference
        if extras:
            api_ms = reference_api_step(cfg, B, C, device, fmap, score, coord, gt, label,
                                        path.anchor, args.steps, args.warmup)
            out["reference_api"] = {
                "ms_per_step": api_ms, "images_per_s": B / (api_ms * 1e-3),
                "vs_path_run": api_ms / (elapsed / args.steps * 1e3),
                "note": "the same step through the reference's function names: patch_batch (batched) "
                        "+ making_frcnn_input(fuse_pool=True) on their kept buffer sets"}
        if not args.no_cpu_baseline and world == 1:       # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(cfg, fmap, score, coord, gt, label)
            out["cpu_baseline_numpy"] = cpu_baseline_numpy(cfg, fmap, score, coord, gt, label)
        if extras:
            del path
            torch.cuda.empty_cache()
            out["single_image"] = single_image(cfg, C, device)
            out["training_side"] = training_side(cfg, C, device)
            out["secondary"] = secondary_workload("ship768", C, device, args.steps, args.warmup, rank)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
