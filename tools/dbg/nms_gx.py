import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from objectdetection_amd import ops
from objectdetection_amd.pipeline import RegionProposalPath
dev = torch.device("cuda:0")
cfg = bench.workload_cfg()
for B in (1, 8, 16, 64):
    fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 64, dev, 1000)
    path = RegionProposalPath(cfg, B, 64, dev, train=False)
    path.run(fmap, score, coord)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for gx in ("0",):
        if gx == "0":
            os.environ.pop("FRCNN_NMS_GX", None)
        else:
            os.environ["FRCNN_NMS_GX"] = gx
        ts = []
        for _ in range(12):
            e0.record(); ops.nms(path.cand, cfg.nms_thr, cfg.post_n, workspace=path.nms_ws); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print(f"B={B} gx={gx if gx != '0' else 'default'}: nms min {min(ts)*1e3:.1f} us  mean {np.mean(ts[2:])*1e3:.1f} us", flush=True)
