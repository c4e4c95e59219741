import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import frcnn_oracle as O
from objectdetection_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
H, W, C = 31, 31, 64
fm = rng.random((1, H, W, C), dtype=np.float32)
boxes = []
for i in range(40):
    y1, x1 = rng.random(2) * 0.6
    h, w = rng.random(2) * 0.4 + 0.01
    boxes.append([y1, x1, min(y1 + h, 1), min(x1 + w, 1)])
boxes[0] = [0, 0, 1, 1]
prop = np.array(boxes, np.float32)[None]
for pool in (1, 2):
    out, _ = ops.roi_crop_pool_fwd(torch.tensor(fm, device=dev), torch.tensor(prop, device=dev), (14, 14), pool)
    if pool == 1:
        ref = O.crop_and_resize(fm, prop.reshape(-1, 4), np.zeros(40, np.int64)).reshape(1, 40, 14, 14, C)
    else:
        ref, _ = O.roi_crop_pool(fm, prop, (14, 14), 2)
    got = out.cpu().numpy().reshape(ref.shape)
    bad = np.argwhere(got != ref)
    print("pool", pool, "mismatches", len(bad))
    if len(bad):
        for ax, nm in enumerate(["b", "r", "y", "x", "c"]):
            u, cnt = np.unique(bad[:, ax], return_counts=True)
            print("  ", nm, dict(zip(u.tolist(), cnt.tolist())))
        print(bad[:5], got[tuple(bad[0])], ref[tuple(bad[0])])
    if len(bad):
        b0 = tuple(bad[0][:4])
        print("got", got[b0][40:64])
        print("ref", ref[b0][40:64])
