#!/usr/bin/env python3
"""Adversarial shape: a star -- most edges point at ONE destination row (a row spanning tens of
thousands of chunks): is the fix-up chain a problem?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lpgnn_amd  # noqa: E402

dev = torch.device("cuda:0")
N, E, d = 1_000_000, 20_000_000, 128
gen = torch.Generator(device=dev)
gen.manual_seed(0)
src = torch.randint(0, N, (E,), generator=gen, device=dev)
for frac in (0.0, 0.5, 1.0):
    dst = torch.randint(0, N, (E,), generator=gen, device=dev)
    k = int(frac * E)
    dst[:k] = 12345
    g = lpgnn_amd.CSRGraph.from_edge_index(torch.stack([src, dst]), N, N)
    x = torch.randn((N, d), device=dev)
    for _ in range(2):
        y = g.spmm(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = g.spmm(x)
    e1.record()
    torch.cuda.synchronize()
    ref = x[src[:k]].double().sum(0) if k else None
    err = None
    if k:
        rest = dst[k:] == 12345
        ref = ref + x[src[k:][rest]].double().sum(0)
        err = float((y[12345].double() - ref).abs().max() / ref.abs().max())
    print(json.dumps({"star_fraction": frac, "ms": round(e0.elapsed_time(e1) / 5, 3), "hub_row_rel_err": err}), flush=True)
