"""Every op of the generic (kernel-per-op) path across problem SIZES, to catch dispatch cliffs between the small-graph and
the big-graph regimes: time should grow monotonically and smoothly with n.  python profiles/size_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lpgnn_amd
from lpgnn_amd import ops
from bench import rmat_edges

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("# csr_spmm_f32 (unit weights + gcn self loops), us")
for d in (32, 128):
    for E in (10_000, 30_000, 100_000, 300_000, 1_000_000, 3_000_000, 10_000_000, 30_000_000):
        N = max(E // 20, 64)
        ei = rmat_edges(N, E, 7, dev)
        g = lpgnn_amd.CSRGraph.from_edge_index(ei, N, N, gcn_self_loops=True).gcn_normalize_()
        x = torch.randn(N, d, device=dev)
        us = timed(lambda: g.spmm(x))
        alg = (E + N) * (8 + d * 4) + N * (8 + d * 4)
        print(f"spmm d={d} E={E} N={N}: {us:.1f} us  {alg / us / 1e3:.0f} GB/s  max_row={g.max_row}", flush=True)
        del g, x, ei
print("# graph_norm f32 (relu), us")
for d in (32, 128):
    w, b, m = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev), torch.rand(d, device=dev) + 0.5
    for n in (500, 2_000, 8_000, 32_000, 128_000, 512_000, 2_000_000):
        x = torch.randn(n, d, device=dev)
        us = timed(lambda: ops.graph_norm(x, w, b, m, relu=True))
        print(f"graph_norm d={d} n={n}: {us:.1f} us  {n * d * 12 / us / 1e3:.0f} GB/s", flush=True)
print("# pair_gather_mul / pair_readout, us")
for P in (4_000, 16_000, 64_000, 256_000, 1_000_000, 4_000_000):
    n, d = max(P // 8, 64), 32
    x = torch.randn(n, d, device=dev)
    pos = torch.randint(0, n, (P, 2), device=dev)
    us = timed(lambda: ops.pair_gather_mul(x, pos))
    print(f"pair_gather_mul d={d} P={P}: {us:.1f} us  {P * (16 + d * 12) / us / 1e3:.0f} GB/s", flush=True)
print("# degree (int64 histogram), us")
for E in (3_000, 30_000, 300_000, 3_000_000, 4_194_304, 8_000_000, 30_000_000):
    for N in (620, 20_000, 1_000_000):
        idx = torch.randint(0, N, (E,), device=dev)
        us = timed(lambda: ops.degree(idx, N), reps=10)
        print(f"degree E={E} N={N}: {us:.1f} us  {(E * 8 + N * 8) / us / 1e3:.0f} GB/s", flush=True)
