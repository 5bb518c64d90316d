#!/usr/bin/env python3
"""does the empty-row fill of the headline SpMM hide behind the main kernel when it runs on a second stream?
(main kernel: 1 workgroup per CU by LDS reservation, gather-bound at 5.6 TB/s of 8)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, lpgnn_amd
dev = torch.device("cuda:0")
N, E, d = 5_000_000, 100_000_000, 128
ei = bench.rmat_edges(N, E, 5, dev)
g = lpgnn_amd.CSRGraph.from_edge_index(ei, N, N)
x = torch.randn((N, d), device=dev)
y = torch.empty((N, d), device=dev)
side = torch.cuda.Stream()
fill = torch.empty((g.n_empty, d), device=dev)            # as many bytes as the empty rows' fill


def timed(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def overlapped():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        fill.zero_()
    g.spmm(x, out=y, skip_empty_rows=True)
    cur.wait_stream(side)


print(f"n_empty {g.n_empty} of {N}")
print(f"spmm (main + fix-up + empty rows): {timed(lambda: g.spmm(x, out=y)):.3f} ms")
print(f"spmm without the empty-row fill:    {timed(lambda: g.spmm(x, out=y, skip_empty_rows=True)):.3f} ms")
print(f"fill-sized memset alone:            {timed(lambda: fill.zero_()):.3f} ms")
print(f"spmm without fill || memset on a second stream: {timed(overlapped):.3f} ms")
