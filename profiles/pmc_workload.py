#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs).

1. calibration: a PERMUTATION graph (|E| = |V| = 5 M, every source row read exactly once, d = 128
   fp32).  Compulsory HBM traffic is known: X 2.56 GB read + 20 MB col + 40 MB rowid, Y 2.56 GB
   written -- X (2.56 GB) is 10x the 256 MiB Infinity Cache, so nothing can be re-served on die.
   This calibrates what FETCH_SIZE reports for THIS access pattern (16 B/lane row gathers).
2. the bench.py workload itself (RMAT |V| = 5 M, |E| = 100 M, d = 128), 3 steps.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench        # noqa: E402
import lpgnn_amd    # noqa: E402
from lpgnn_amd import _lib    # noqa: E402

_lib.set_option("spmm_fill_overlap", 0)   # the counters are per dispatch: keep main, fix-up, empty rows in that order

dev = torch.device("cuda:0")
N, d = 5_000_000, 128
gen = torch.Generator(device=dev)
gen.manual_seed(1)
perm = torch.randperm(N, generator=gen, device=dev)
ei = torch.stack([perm, torch.arange(N, device=dev)])
g = lpgnn_amd.CSRGraph.from_edge_index(ei, N, N)
x = torch.randn((N, d), device=dev)
for _ in range(3):
    y = g.spmm(x)
torch.cuda.synchronize()
print("calibration done: 3 permutation SpMM launches", flush=True)
del g, y, ei

g, x = bench.make_rmat_problem(N, 100_000_000, d, seed=5, device=dev)
for _ in range(3):
    y = g.spmm(x)
torch.cuda.synchronize()
print("workload done: 3 RMAT SpMM launches, nnz =", g.nnz(), flush=True)
