"""SpMM on the collab-shaped graph (|V| = 235 868, 2.57 M nnz + self loops, d = 32 / 64): chunk-length sweep.
The automatic chunk (16 at this size) was tuned on the 100 M-edge headline; the fix-up kernel costs as much as the main one here."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import lpgnn_amd  # noqa: E402
from lpgnn_amd import _lib, twowl  # noqa: E402

dev = "cuda:0"
N, E = 235_868, 1_285_465
ei = twowl.double(bench.rmat_edges(N, E, 1, dev))
for gcn in (True, False):
    g = lpgnn_amd.CSRGraph.from_edge_index(ei, N, N, gcn_self_loops=gcn)
    if gcn:
        g.gcn_normalize_()
    for d in (32, 64):
        x = torch.randn(N, d, device=dev)
        for chunk in (0, 16, 32, 64, 128, 256, 512):
            with _lib.options(spmm_chunk=chunk):
                ms = bench._time_ms(lambda: g.spmm(x, use_val=gcn), 30, warm=5)
            print(f"gcn={gcn} d={d} chunk={chunk:3d}: {ms * 1e3:7.1f} us", flush=True)
