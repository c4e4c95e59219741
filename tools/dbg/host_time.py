import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from objectdetection_amd.pipeline import RegionProposalPath
dev = torch.device("cuda:0")
for name, B in (("ship768", 16), ("600x1000", 64)):
    cfg = bench.workload_cfg(name)
    fmap, score, coord, gt, label = bench.make_inputs(cfg, B, 512, dev, 1000)
    path = RegionProposalPath(cfg, B, 512, dev, train=True, depth=2)
    for i in range(4):
        path.run(fmap, score, coord, gt, label, seed=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for i in range(n):
        path.run(fmap, score, coord, gt, label, seed=i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e6*(t1-t0)/n:.0f} us per step; with sync {1e6*(t2-t0)/n:.0f} us per step", flush=True)
