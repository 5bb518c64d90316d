#!/usr/bin/env python3
"""quick timing of the dense linear at the headline shapes (fp32 / bf16, n = 5 M)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _opts                # LPG_* variables -> the library's option table (profiles/_opts.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpgnn_amd import ops
dev = torch.device("cuda:0")
def timed(fn, it=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
n = 5_000_000
for dt in (torch.float32, torch.bfloat16):
    for k, m in ((32, 32), (64, 64), (128, 32), (128, 128)):
        if dt == torch.float32 and k > 128: continue
        x = torch.randn((n, k), device=dev).to(dt); w = (torch.randn((m, k), device=dev) * 0.1).to(dt)
        r = []
        for mode in sys.argv[1:] or ["auto"]:
            if mode == "auto": os.environ.pop("LPG_LINEAR_WRES", None)
            else: os.environ["LPG_LINEAR_WRES"] = mode
            _opts.sync()
            r.append(f"{mode}={timed(lambda: ops.linear(x, w)):.3f}")
        ref = x[:4096].double() @ w.double().t()
        os.environ["LPG_LINEAR_WRES"] = "3"; _opts.sync()
        err = float(((ops.linear(x, w)[:4096].double() - ref).abs() / (1 + ref.abs())).max())
        print(str(dt)[6:], k, m, " ".join(r), "ms  v3 err", f"{err:.1e}", flush=True)
        del x, w
