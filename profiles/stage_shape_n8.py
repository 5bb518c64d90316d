#!/usr/bin/env python3
"""One pipeline stage of the source-partitioned SpMM as rank 0 of 8 sees it: X shard 625 K rows (320 MB),
25 M edges into 1.25 M destination rows (a quarter of V), d = 128 fp32 -- occupancy limiter on / off."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _opts                # LPG_* variables -> the library's option table (profiles/_opts.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, lpgnn_amd
dev = torch.device("cuda:0")
V, E, d, G, S = 5_000_000, 100_000_000, 128, int(os.environ.get("WORLD", "8")), 4
ei = bench.rmat_edges(V, E, 5, dev, src_range=(0, V // G))
keep = ei[1] % S == 0                      # a quarter of the destination rows, like one of 4 stages
ei = torch.stack([ei[0][keep], ei[1][keep] // S])
g = lpgnn_amd.CSRGraph.from_edge_index(ei, V // S, V // G)
x = torch.randn((V // G, d), device=dev)
for lds in ("-1", "0", "65536", "90000"):
    os.environ["LPG_SPMM_LDS"] = lds
    _opts.sync()
    ts = []
    for _ in range(4):
        g.spmm(x); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(8): g.spmm(x)
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 8)
    print(f"world={G} stage: nnz={g.nnz()} rows={g.n_rows} x_rows={V // G}  LDS={lds}: {statistics.median(ts):.3f} ms", flush=True)
