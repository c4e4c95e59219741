#!/usr/bin/env python3
"""Times frcnn_fc1_fwd on the head's first dense layer of a batch-8 step ([16,000, 25,088] x
[25,088, 4,096], float32) beside torch.addmm + relu."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, K, N = int(os.environ.get("FC_M", "16000")), 25088, 4096
g = torch.Generator(device=dev)
g.manual_seed(1)
x = torch.relu(torch.randn((M, K), generator=g, device=dev))
w = torch.randn((K, N), generator=g, device=dev) * 0.01
b = torch.randn((N,), generator=g, device=dev)
out = torch.empty((M, N), device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


t = timed(lambda: ops.fc1_fwd(x, w, b, True, out=out))
print(f"frcnn_fc1_fwd {M}x{K}x{N}: {t:.2f} ms  {2.0 * M * K * N / t / 1e9:.1f} TFLOP/s  "
      f"{2.0 * M * K * N / t / 1e9 / 157.3:.3f} of the f32 MFMA peak")
if os.environ.get("FC_TORCH", "1") == "1":
    t = timed(lambda: torch.relu_(torch.addmm(b, x, w, out=out)))
    print(f"torch addmm+relu: {t:.2f} ms  {2.0 * M * K * N / t / 1e9:.1f} TFLOP/s")
