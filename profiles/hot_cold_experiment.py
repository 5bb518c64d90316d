#!/usr/bin/env python3
"""Experiment: does separating hub-source edges from the rest in TIME pay?  Split the bench graph's
edges by source out-degree rank into hot (top H sources) and cold, run the existing SpMM on each
part, and compare t_hot + t_cold with the single pass."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lpgnn_amd  # noqa: E402

dev = torch.device("cuda:0")
N, E, d = 5_000_000, 100_000_000, 128
ei = bench.rmat_edges(N, E, 5, dev)
x = torch.randn((N, d), device=dev)


def t(g):
    for _ in range(3):
        g.spmm(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.spmm(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


full = t(lpgnn_amd.CSRGraph.from_edge_index(ei, N, N))
outdeg = torch.bincount(ei[0], minlength=N)
order = torch.argsort(outdeg, descending=True)
for H in (8192, 32768, 131072, 524288):
    hot = torch.zeros(N, dtype=torch.bool, device=dev)
    hot[order[:H]] = True
    m = hot[ei[0]]
    gh = lpgnn_amd.CSRGraph.from_edge_index(ei[:, m], N, N)
    gc = lpgnn_amd.CSRGraph.from_edge_index(ei[:, ~m], N, N)
    th, tc = t(gh), t(gc)
    print(json.dumps({"H": H, "hot_edge_frac": round(float(m.float().mean()), 4), "t_full_ms": round(full, 3),
                      "t_hot_ms": round(th, 3), "t_cold_ms": round(tc, 3), "sum_ms": round(th + tc, 3),
                      "hot_GBs": round(int(m.sum()) * 516 / th / 1e6), "cold_GBs": round(int((~m).sum()) * 516 / tc / 1e6)}),
          flush=True)
    del gh, gc
