"""bench.py's seal_sharded object alone (collab-shaped SEAL scoring, uniform candidates and the reference's mix)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.cuda.set_device(0)
print(json.dumps(bench.seal_sharded_bench("cuda:0", 1, 0, None), indent=1))
