#!/usr/bin/env python3
"""bf16-storage linear (W-resident kernels) on the wide shapes, and fp32 outputs of two column strips"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpgnn_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for dt, n, k, m in [(torch.bfloat16, 2_000_000, 256, 256), (torch.bfloat16, 10_000_000, 256, 256), (torch.bfloat16, 5_000_000, 128, 128),
                    (torch.bfloat16, 5_000_000, 256, 128), (torch.bfloat16, 400_000, 256, 256), (torch.bfloat16, 50_000, 256, 256),
                    (torch.bfloat16, 2_000_000, 128, 256), (torch.float32, 2_000_000, 128, 256), (torch.float32, 2_000_000, 64, 256)]:
    x = torch.randn((n, k), device=dev).to(dt)
    w = (torch.randn((m, k), device=dev) * 0.1).to(dt)
    with _lib.options(linear_x3=0):
        ms = timed(lambda: ops.linear(x, w))
    gb = n * (k + m) * x.element_size() / 1e9
    print(f"{str(dt)[6:]} n={n} k={k} m={m}: {ms:.3f} ms ({gb / ms:.2f} TB/s = {gb / ms / 8:.2f} of 8 TB/s, {2 * n * k * m / ms / 1e9:.0f} TF)", flush=True)
    del x
