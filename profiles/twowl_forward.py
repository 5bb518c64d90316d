#!/usr/bin/env python3
"""fb-pages-food TwoWL scoring forward: eager vs hipGraph replay (run under rocprofv3 --kernel-trace
--stats for the per-kernel picture)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.twowl_forward_bench(torch.device("cuda:0"), iters=50)))
