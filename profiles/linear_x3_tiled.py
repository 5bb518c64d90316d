#!/usr/bin/env python3
"""the strip form (linear_x3 = 1) next to the tiled form (= 2) of the bf16-plane fp32 linear on the big shapes"""
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


for n, k, m in [(5_000_000, 128, 128), (2_000_000, 256, 256), (400_000, 128, 128), (100_000, 256, 256), (50_000, 128, 128)]:
    x = torch.randn((n, k), device=dev)
    w = torch.randn((m, k), device=dev) * 0.1
    out = {}
    for mode in (1, 2):
        with _lib.options(linear_x3=mode):
            out[mode] = timed(lambda: ops.linear(x, w))
    print(f"n={n} k={k} m={m}: strip {out[1]:.3f} ms  tiled {out[2]:.3f} ms ({n * (k + m) * 4 / out[2] / 1e9:.2f} TB/s, "
          f"{12 * n * k * m / out[2] / 1e9:.0f} TF of bf16 MFMA)", flush=True)
    del x
