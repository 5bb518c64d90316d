import json, sys, torch
sys.path.insert(0, "/root/repo")
import bench
torch.cuda.set_device(0)
print(json.dumps(bench.twowl_scaled_bench("cuda:0", iters=10, shape=(576_289, 30_326_273, 3, "ppa")), indent=1))
