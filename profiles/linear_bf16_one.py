#!/usr/bin/env python3
"""one shape of the bf16-storage linear, for rocprofv3 --pmc passes: n k m"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpgnn_amd import ops
n, k, m = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
torch.manual_seed(0)
x, w = torch.randn((n, k), device=dev).to(torch.bfloat16), (torch.randn((m, k), device=dev) * 0.1).to(torch.bfloat16)
for _ in range(6):
    y = ops.linear(x, w)
torch.cuda.synchronize()
