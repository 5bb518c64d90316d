#!/usr/bin/env python3
"""A/B of the dense linear (K5): tiled kernels (linear.hip / linear_bf16.hip) vs the persistent
W-resident kernels (linear_wres.hip), same process, interleaved; each checked against an fp64 matmul."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _opts                # LPG_* variables -> the library's option table (profiles/_opts.py)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpgnn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, it=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


shapes = [  # n, k, m, ldx, w_is_km, dtype, act, bias
    (5_000_000, 128, 128, 128, False, torch.float32, 0, False),
    (5_000_000, 128, 128, 128, False, torch.float32, 1, True),
    (5_000_000, 128, 32, 128, False, torch.float32, 0, False),
    (5_000_000, 32, 32, 32, True, torch.float32, 2, False),
    (5_000_000, 64, 64, 64, False, torch.float32, 0, True),
    (1_464_374, 134, 32, 136, True, torch.float32, 0, False),
    (1_464_374, 134, 32, 134, True, torch.float32, 0, False),
    (2_000_000, 256, 64, 256, False, torch.float32, 0, False),
    (100_001, 100, 20, 100, False, torch.float32, 1, True),
    (5_000_000, 128, 128, 128, False, torch.bfloat16, 0, False),
    (5_000_000, 128, 128, 128, False, torch.bfloat16, 1, True),
    (5_000_000, 256, 128, 256, False, torch.bfloat16, 0, False),
    (5_000_000, 128, 32, 128, False, torch.bfloat16, 0, False),
    (1_464_374, 134, 32, 136, True, torch.bfloat16, 2, False),
    (5_000_000, 32, 32, 32, True, torch.bfloat16, 0, True),
    (100_001, 100, 20, 104, False, torch.bfloat16, 1, True),
]
ACTS = [None, 'relu', 'tanh']
for n, k, m, ldx, km, dt, acti, has_b in shapes:
    act = ACTS[acti]
    xb = torch.randn((n, ldx), device=dev).to(dt)
    if ldx > k:
        xb[:, k:] = float("nan")                         # the pad must never be used
    x = xb[:, :k]
    w = (torch.randn((k, m) if km else (m, k), device=dev) * 0.1).to(dt)
    b = torch.randn(m, device=dev) if has_b else None
    out = {}
    res = {}
    for mode in ("0", "1", "2", "3"):
        os.environ["LPG_LINEAR_WRES"] = mode
        _opts.sync()
        y = ops.linear(x, w, b, w_is_km=km, act=act)
        res[mode] = y
        out[mode] = timed(lambda: ops.linear(x, w, b, w_is_km=km, act=act))
    rows = torch.randint(0, n, (4096,), device=dev)
    rows[0], rows[1] = n - 1, 0
    ref = x[rows].double() @ (w.double() if km else w.double().t())
    if b is not None:
        ref = ref + b.double()
    ref = torch.relu(ref) if acti == 1 else (torch.tanh(ref) if acti == 2 else ref)
    tol = 2e-5 if dt == torch.float32 else 1.5e-2
    errs = {md: float(((res[md][rows].double() - ref).abs() / (1 + ref.abs())).max()) for md in res}
    same_shape = res["1"].shape == (n, m) and bool(torch.isfinite(res["1"].float()).all())
    elt = 4 if dt == torch.float32 else 2
    gb = n * (k + m) * elt / 1e9
    best = min(out["1"], out["2"], out["3"])
    ok = all(errs[md] < tol for md in errs) and same_shape and bool(torch.isfinite(res["2"].float()).all())
    print(f"n={n} k={k} m={m} ldx={ldx} km={int(km)} {str(dt)[6:]} act={act} bias={int(has_b)}: tiled {out['0']:.3f} "
          f"wres1 {out['1']:.3f} wres2 {out['2']:.3f} wres3 {out['3']:.3f} ms (best {gb / best * 1e3:.0f} GB/s, {2 * n * k * m / best / 1e9:.1f} TF)  "
          f"err {errs['0']:.1e} {errs['1']:.1e} {errs['2']:.1e} {errs['3']:.1e} {'OK' if ok else 'FAIL'}", flush=True)
    del xb, x, w, res
