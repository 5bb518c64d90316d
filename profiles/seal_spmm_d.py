#!/usr/bin/env python3
"""SpMM on the block-diagonal SEAL batch graph (1.46 M rows, 8.5 M nnz, weighted, tanh epilogue) vs d and chunk."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _opts                # LPG_* variables -> the library's option table (profiles/_opts.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from lpgnn_amd import seal
dev = torch.device("cuda:0")
N, E, B = 235_868, 1_285_465, 100_000
ei = bench.rmat_edges(N, E, 1, dev)
graph = seal.build_train_csr(ei.t().cpu().numpy(), N, test_ratio=0.0, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(2)
pairs = torch.randint(0, N, (B, 2), generator=gen, device=dev)
batch = seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0)
tag_map, Tn = seal.compact_tag_map(batch)
def T(fn, it=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3
for chunk in sys.argv[1:] or ["0"]:
    os.environ["LPG_SPMM_CHUNK"] = chunk
    _opts.sync()
    g, d_inv, x = seal.gnn_inputs(batch, tag_map, Tn, None)
    out = []
    for d in (1, 2, 4, 8, 16, 32, 64):
        h = torch.randn((g.n_cols, d), device=dev)
        out.append(f"d={d}: {T(lambda: g.spmm(h, act='tanh')):.0f}")
    print(f"chunk={chunk} nnz={g.nnz()} rows={g.n_rows}  us: " + "  ".join(out), flush=True)
