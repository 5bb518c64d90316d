#!/usr/bin/env python3
"""Tuning sweep of the CSR-SpMM kernel on the bench.py workload, ONE process, variants interleaved
over several rounds on the same graph / same allocation (device-to-device and run-to-run spread is
+-8 % on this pool, so only in-process interleaved numbers are comparable).
    python profiles/sweep_spmm.py [K=V,K=V ...]      (each argument is one variant)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _opts                # LPG_* variables -> the library's option table (profiles/_opts.py)
_opts.use_tuning_build()    # MODE 2 (gather-only) exists in the -DLPG_TUNING build only
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KEYS = ["LPG_SPMM_CHUNK", "LPG_SPMM_U", "LPG_SPMM_MODE", "LPG_SPMM_LDS", "LPG_SPMM_PIPE", "LPG_SPMM_BLOCK", "LPG_SPMM_OFF32", "LPG_SPMM_NO_EDGE"]
variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[1:]] or [{}]
dev = torch.device("cuda:0")
N, E, d = 5_000_000, 100_000_000, int(os.environ.get("SWEEP_D", "128"))
DT = torch.bfloat16 if os.environ.get("SWEEP_DTYPE") == "bf16" else torch.float32
g, x = bench.make_rmat_problem(N, E, d, seed=5, device=dev, dtype=DT)
nnz = g.nnz()
alg = bench.spmm_algorithmic_bytes(N, nnz, d, 2 if DT == torch.bfloat16 else 4)
times = {i: [] for i in range(len(variants))}
ROUNDS, STEPS = 4, 8
for rnd in range(ROUNDS):
    for i, v in enumerate(variants):
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update({("LPG_SPMM_" + k if not k.startswith("LPG_") else k): val for k, val in v.items() if val != ""})
        _opts.sync()
        g._build_plan()                      # the plan depends on the chunk size
        for _ in range(2):
            g.spmm(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(STEPS):
            g.spmm(x)
        e1.record()
        torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / STEPS)
for i, v in enumerate(variants):
    t = times[i]
    med = statistics.median(t)
    print(json.dumps({"d": d, "dtype": str(DT)[6:], "variant": v, "ms_median": round(med, 3), "ms_min": round(min(t), 3), "ms_max": round(max(t), 3),
                      "frac_of_8TBs": round(alg / (med * 1e-3) / 8e12, 4)}), flush=True)
