#!/usr/bin/env python3
"""A/B of the CSR-SpMM between the current library and an older build (profiles/_oldlib/liblpg_old.so,
old ABI without n_cols; built by hand from `git show <commit>:.../spmm.hip`) in ONE process on the
same graph -- separates real regressions from box-to-box variance."""
import ctypes
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lpgnn_amd  # noqa: E402

old = ctypes.CDLL(os.path.join(ROOT, "profiles", "_oldlib", "liblpg_old.so"))
for fn in (old.lpg_csr_spmm_f32, old.lpg_csr_spmm_bf16):
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_void_p] + [ctypes.c_int64] * 4 + \
                  [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
old.lpg_csr_spmm_bytes.restype = ctypes.c_int64
old.lpg_csr_spmm_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]

dev = torch.device("cuda:0")
N, E = 5_000_000, 100_000_000
g, _ = bench.make_rmat_problem(N, E, 4, seed=5, device=dev)
for dtype, d in ((torch.float32, 128), (torch.bfloat16, 128), (torch.float32, 64), (torch.bfloat16, 256), (torch.float32, 32)):
    x = torch.randn((N, d), device=dev).to(dtype)
    y = torch.empty((N, d), device=dev, dtype=dtype)
    nb = old.lpg_csr_spmm_bytes(g.e_cap, d)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    fn = old.lpg_csr_spmm_f32 if dtype == torch.float32 else old.lpg_csr_spmm_bf16

    def run_old():
        rc = fn(g._plan.data_ptr(), g.rowptr.data_ptr(), g.col.data_ptr(), None, x.data_ptr(), d, y.data_ptr(), d, N,
                g.e_cap, d, None, None, 0, g.n_empty, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
        assert rc == 0

    def run_new():
        g.spmm(x, out=y)

    res = {"old": [], "new": []}
    for rnd in range(4):
        for name, f in (("old", run_old), ("new", run_new)):
            for _ in range(2):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 8)
    print(json.dumps({"dtype": str(dtype)[6:], "d": d, "old_ms": round(statistics.median(res["old"]), 3),
                      "new_ms": round(statistics.median(res["new"]), 3)}), flush=True)
    del x, y, ws
