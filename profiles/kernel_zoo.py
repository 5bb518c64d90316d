#!/usr/bin/env python3
"""Secondary kernels at scale: achieved algorithmic GB/s (or TFLOP/s) against their roofline."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lpgnn_amd  # noqa: E402
from lpgnn_amd import ops, twowl  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def report(name, ms, nbytes=None, flops=None, note=""):
    r = {"kernel": name, "ms": round(ms, 3)}
    if nbytes is not None:
        r["alg_GBs"] = round(nbytes / ms / 1e6, 1)
        r["frac_of_8TBs"] = round(nbytes / (ms * 1e-3) / 8e12, 3)
    if flops is not None:
        r["TFLOPs"] = round(flops / ms / 1e9, 2)
    if note:
        r["note"] = note
    print(json.dumps(r), flush=True)


N, E, d = 5_000_000, 100_000_000, 128
ei = bench.rmat_edges(N, E, 5, dev)
report("degree int64 (scatter_add ones), E=100M N=5M", timeit(lambda: twowl.degree(ei, N), 5), E * 8 + N * 8)
ei_s = ei[:, :10_000_000].contiguous()
src = torch.randn((10_000_000, d), device=dev)
out = torch.zeros((N, d), device=dev)
_min = ops.SCATTER_SORT_MIN
ops.SCATTER_SORT_MIN = 1 << 62
report("scatter_add f32 atomics, E=10M d=128", timeit(lambda: ops.scatter_add_rows(src, ei_s[1], out), 5),
       10_000_000 * (8 + d * 4) + 0, note="bound: memory-side float atomics ~1.3 TB/s of added bytes")
ops.SCATTER_SORT_MIN = _min
idx_s = ei_s[1].contiguous()
report("scatter_add f32 grouped by destination (the route lists >= 2^20 take), E=10M d=128",
       timeit(lambda: ops.scatter_add_rows(src, idx_s, out, True), 5), 10_000_000 * (8 + d * 4) + N * d * 4,
       note="plan memoised per index tensor; first call adds the grouping (profiles/r01/scatter_add_time.log)")
ops._scatter_plans.clear()
del idx_s
del src, out
report("coo_to_csr + plan (one-time), E=100M", timeit(lambda: lpgnn_amd.CSRGraph.from_edge_index(ei, N, N), 3, 1), E * 16)
g = lpgnn_amd.CSRGraph.from_edge_index(ei, N, N, gcn_self_loops=True)
def _gcn():
    g.val = None                      # unit weights (edge_weight=None), the reference's only use
    g.gcn_normalize_()


report("gcn_norm on CSR (unit weights), E=105M", timeit(_gcn, 5), (E + N) * 12 + N * 12)
del g
x = torch.randn((N, d), device=dev)
wt, bs, ms_ = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev), torch.rand(d, device=dev) + 0.5
report("graph_norm f32 n=5M d=128", timeit(lambda: ops.graph_norm(x, wt, bs, ms_, relu=True)), N * d * 4 * 3)
W = torch.randn((d, d), device=dev)
report("linear f32 MFMA n=5M k=128 m=128", timeit(lambda: ops.linear(x, W)), N * d * 4 * 2, flops=2 * N * d * d)
xb, Wb = x.bfloat16(), W.bfloat16()
report("linear bf16 MFMA n=5M k=128 m=128", timeit(lambda: ops.linear(xb, Wb)), N * d * 2 * 2, flops=2 * N * d * d)
W2 = torch.randn((256, 256), device=dev)
x2 = torch.randn((2_000_000, 256), device=dev)
report("linear f32 MFMA n=2M k=256 m=256", timeit(lambda: ops.linear(x2, W2)), 2_000_000 * 256 * 4 * 2, flops=2 * 2_000_000 * 256 * 256)
x2b, W2b = x2.bfloat16(), W2.bfloat16()
report("linear bf16 MFMA n=2M k=256 m=256", timeit(lambda: ops.linear(x2b, W2b)), 2_000_000 * 256 * 2 * 2, flops=2 * 2_000_000 * 256 * 256)
del x2, x2b
pos = torch.randint(0, N, (10_000_000, 2), device=dev)
report("pair_gather_mul P=10M d=128", timeit(lambda: ops.pair_gather_mul(x, pos), 5), 10_000_000 * (16 + 3 * d * 4))
idx = torch.randint(0, 1000, (N,), device=dev)
tab = torch.randn((1000, d), device=dev)
report("embedding n=5M d=128 (1000-row table)", timeit(lambda: ops.embedding(tab, idx)), N * (8 + d * 4))
