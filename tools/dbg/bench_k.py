import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
bench.WORKLOADS["600x1000"]["cfg"]["pre_k"] = int(os.environ["PRE_K"])
sys.argv = ["bench.py", "--no-cpu-baseline"]
bench.main()
