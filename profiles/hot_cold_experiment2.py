#!/usr/bin/env python3
"""Experiment 2: the hot-source part of the bench graph (top H sources) under different occupancy
settings of the existing kernel -- is a cache-resident gather faster with more waves?"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _opts                # LPG_* variables -> the library's option table (profiles/_opts.py)
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


outdeg = torch.bincount(ei[0], minlength=N)
order = torch.argsort(outdeg, descending=True)
for H in (8192, 32768):
    hot = torch.zeros(N, dtype=torch.bool, device=dev)
    hot[order[:H]] = True
    m = hot[ei[0]]
    for name, sub in (("hot", ei[:, m]), ("cold", ei[:, ~m])):
        g = lpgnn_amd.CSRGraph.from_edge_index(sub, N, N)
        res = {"H": H, "part": name, "edges": sub.shape[1]}
        for lds in ("0", "40000", "65536", "90000"):
            os.environ["LPG_SPMM_LDS"] = lds
            _opts.sync()
            res["lds" + lds] = round(t(g), 3)
        os.environ.pop("LPG_SPMM_LDS"); _opts.sync()
        print(json.dumps(res), flush=True)
        del g
