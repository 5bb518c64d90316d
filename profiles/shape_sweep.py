#!/usr/bin/env python3
"""CSR-SpMM across feature widths / dtypes / weighted on the bench graph (RMAT |V|=5M |E|=100M):
ms per call and fraction of the 8 TB/s HBM roofline on each shape's own algorithmic bytes."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
N, E = 5_000_000, 100_000_000
g, _ = bench.make_rmat_problem(N, E, 4, seed=5, device=dev)
nnz = g.nnz()
w = torch.rand(g.e_cap, device=dev)
for dtype, elt in ((torch.float32, 4), (torch.bfloat16, 2)):
    for d in (32, 64, 128, 256):
        x = torch.randn((N, d), device=dev).to(dtype)
        for weighted in (False, True):
            g.val = w if weighted else None
            for _ in range(3):
                g.spmm(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.spmm(x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            alg = bench.spmm_algorithmic_bytes(N, nnz, d, elt, weighted)
            print(json.dumps({"dtype": str(dtype).split(".")[-1], "d": d, "weighted": weighted, "ms": round(ms, 3),
                              "gedges_s": round(nnz / ms / 1e6, 2), "alg_GBs": round(alg / ms / 1e6, 1),
                              "frac_of_8TBs": round(alg / (ms * 1e-3) / 8e12, 4)}), flush=True)
        del x
