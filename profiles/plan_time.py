#!/usr/bin/env python3
"""time of lpg_csr_plan_build on the bench graph (|V| = 5 M, |E| = 100 M), repeated."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
g, x = bench.make_rmat_problem(5_000_000, 100_000_000, 32, seed=5, device=dev)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g._build_plan(); torch.cuda.synchronize()
    print(f"plan build {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
deg = (g.rowptr[1:] - g.rowptr[:-1])
print("max row", int(deg.max()), "empty rows", int((deg == 0).sum()), "rows", g.n_rows)
