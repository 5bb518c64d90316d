#!/usr/bin/env python3
"""fp32 linear: the bf16-plane kernel (csrc/linear_x3.hip, option linear_x3 = 1) next to the fp32-MFMA kernels
(linear_x3 = 0) on the shapes the hot path has; error of both against an fp64 product on sampled rows."""
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


shapes = [(5_000_000, 128, 128, False, None), (2_000_000, 256, 256, False, None), (100_000, 288, 128, True, "relu"),
          (100_000, 448, 128, True, "relu"), (5_000_000, 128, 64, False, None), (5_000_000, 64, 64, False, None),
          (5_000_000, 128, 32, False, None), (1_000_000, 64, 128, False, None), (50_000, 128, 128, False, None),
          (16_384, 128, 128, False, None), (10_000_000, 256, 256, False, None)]
shapes += [(n, 128, 128, False, None) for n in (32_768, 65_536, 100_000, 200_000, 400_000)]
shapes += [(n, 288, 128, True, "relu") for n in (16_384, 32_768, 50_000)]
shapes += [(n, 256, 256, False, None) for n in (16_384, 50_000, 100_000)]
for n, k, m, km, act in shapes:
    x = torch.randn((n, k), device=dev)
    w = torch.randn((k, m) if km else (m, k), device=dev) * 0.1
    bias = torch.randn(m, device=dev) if act else None
    rows = torch.randint(0, n, (512,), device=dev)
    wd = (w if km else w.t()).double()
    ref = x[rows].double() @ wd + (bias.double() if act else 0)
    if act: ref = ref.relu()
    mag = x[rows].double().abs() @ wd.abs()
    out = {}
    tiled = (not km) and k % 32 == 0 and m % 128 == 0
    for mode in ((0, 1, 2) if tiled else (0, 1)):
        with _lib.options(linear_x3=mode):
            y = ops.linear(x, w, bias, w_is_km=km, act=act)
            err = float(((y[rows].double() - ref).abs() / mag).max())
            ms = timed(lambda: ops.linear(x, w, bias, w_is_km=km, act=act))
        out[mode] = (ms, err)
        del y
    flops = 2 * n * k * m
    print(f"n={n} k={k} m={m} km={km} act={act}: fp32 MFMA {out[0][0]:.3f} ms ({flops / out[0][0] / 1e9:.0f} TF) err/mag {out[0][1]:.1e} | "
          f"bf16 planes {out[1][0]:.3f} ms ({flops / out[1][0] / 1e9:.0f} TF-equiv, {(n * (k + m) * 4) / out[1][0] / 1e9:.2f} TB/s) err/mag {out[1][1]:.1e}"
          + (f" | tiled {out[2][0]:.3f} ms ({(n * (k + m) * 4) / out[2][0] / 1e9:.2f} TB/s) err/mag {out[2][1]:.1e}" if tiled else ""),
          flush=True)
    del x
