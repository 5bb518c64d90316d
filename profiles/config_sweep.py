#!/usr/bin/env python3
"""The other BASELINE.json configs' SpMM shapes on one GPU (synthetic stand-ins):
config 4 "ogbl-ppa": |V| = 576 289, 60.65 M nnz, d = 64;  config 5: RMAT |V| = 10 M, |E| = 200 M,
d = 256 (fp32 and bf16), unweighted and weighted."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
for name, N, E, d, seed in (("config4 ppa-shaped", 576_289, 60_652_546, 64, 3), ("config5 rmat", 10_000_000, 200_000_000, 256, 4)):
    g, _ = bench.make_rmat_problem(N, E, 4, seed=seed, device=dev)
    nnz = g.nnz()
    w = torch.rand(g.e_cap, device=dev)
    for dtype, elt in ((torch.float32, 4), (torch.bfloat16, 2)):
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
            print(json.dumps({"config": name, "V": N, "E": nnz, "d": d, "dtype": str(dtype)[6:], "weighted": weighted,
                              "ms": round(ms, 3), "gedges_s": round(nnz / ms / 1e6, 2),
                              "frac_of_8TBs": round(alg / (ms * 1e-3) / 8e12, 4)}), flush=True)
        del x
    del g, w
    torch.cuda.empty_cache()
