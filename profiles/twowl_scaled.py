"""TwoWL over the structured tuple graph: the fb-pages-food forward (explicit vs implicit) and the collab-shaped one
(bench.py twowl_scaled).  python profiles/twowl_scaled.py [--fb-only]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
r = bench.twowl_forward_bench(dev, iters=50)
print(json.dumps({k: r[k] for k in ("us_per_forward", "us_per_forward_eager", "implicit")}, indent=1))
if "--fb-only" not in sys.argv:
    print(json.dumps(bench.twowl_scaled_bench(dev), indent=1))
