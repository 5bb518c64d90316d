#!/usr/bin/env python3
"""degree() on one instance (for rocprofv3 --kernel-trace --stats): RMAT |V| = 5 M, |E| = 100 M."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from lpgnn_amd import ops
dev = torch.device("cuda:0")
N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5_000_000, 100_000_000)
idx = bench.rmat_edges(N, E, 5, dev)[1].contiguous()
for _ in range(6):
    d = ops.degree(idx, N)
torch.cuda.synchronize()
assert torch.equal(d, torch.bincount(idx, minlength=N))
print("ok")
