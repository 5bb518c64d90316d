#!/usr/bin/env python3
"""one shape of the fp32 linear on the bf16-plane kernel, for rocprofv3 --pmc passes: n k m [mode]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpgnn_amd import ops, _lib
n, k, m = (int(a) for a in sys.argv[1:4])
_lib.set_option("linear_x3", int(sys.argv[4]) if len(sys.argv) > 4 else 1)
dev = torch.device("cuda:0")
torch.manual_seed(0)
x, w = torch.randn((n, k), device=dev), torch.randn((m, k), device=dev) * 0.1
for _ in range(6):
    y = ops.linear(x, w)
torch.cuda.synchronize()
