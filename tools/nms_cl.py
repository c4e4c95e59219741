import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from objectdetection_amd import ops
dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
B = int(os.environ.get("NMS_B", "64"))
cl = bench.clustered_boxes(B, cfg.pre_k, dev, seed=77)
ws = torch.empty(ops.nms_workspace_bytes(B, cfg.pre_k, cfg.post_n), dtype=torch.uint8, device=dev)
for _ in range(12):
    ops.nms(cl, cfg.nms_thr, cfg.post_n, workspace=ws)
    torch.cuda.synchronize()
