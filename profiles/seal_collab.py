#!/usr/bin/env python3
"""configs[2] stand-in: "ogbl-collab"-shaped synthetic SEAL scoring forward.
RMAT |V| = 235 868, 1 285 465 undirected edges (seed 1), 100 K candidate pairs (seed 2), 1-hop
enclosing subgraphs capped at 100 nodes, X = [onehot(DRNL) || 128-d N(0,1) features], 4 graph convs
(random-init weights of the reference's architecture, T+128 -> 32 -> 32 -> 32 -> 1) + DGCNN head."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lpgnn_amd import seal  # noqa: E402

dev = torch.device("cuda:0")
N, E, B, D = 235_868, 1_285_465, 100_000, 128
ei = bench.rmat_edges(N, E, 1, dev)
graph = seal.build_train_csr(ei.t().cpu().numpy(), N, test_ratio=0.0, device=dev)
gen = torch.Generator(device=dev)
gen.manual_seed(2)
pairs = torch.randint(0, N, (B, 2), generator=gen, device=dev)
emb = torch.randn((N, D), generator=gen, device=dev)
k = 18


BF16 = "--bf16" in sys.argv          # configs[2] says bf16: features, weights and activations stored in bf16


def run():
    batch = seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0)
    tag_map, T = seal.compact_tag_map(batch)
    g, d_inv, x = seal.gnn_inputs(batch, tag_map, T, emb, dtype=torch.bfloat16 if BF16 else torch.float32, build_csr=BF16)
    return batch, T, g, d_inv, x


batch, T, g, d_inv, x = run()
torch.manual_seed(0)
gw = [torch.randn(a, b, device=dev) * 0.1 for a, b in ((T + D, 32), (32, 32), (32, 32), (32, 1))]
hw = {"conv1d_kernel_1": torch.randn(96, 1, 16, device=dev) * 0.1, "conv1d_kernel_2": torch.randn(5, 16, 32, device=dev) * 0.1,
      "weight_1": torch.randn((k - 4) * 32, 128, device=dev) * 0.1, "bias_1": torch.zeros(128, device=dev),
      "weight_2": torch.randn(128, 2, device=dev), "bias_2": torch.zeros(2, device=dev)}


gw16 = [w.bfloat16() for w in gw]


def score():
    batch, T, g, d_inv, x = run()
    if BF16:
        z, key = seal.graph_conv_stack(g, d_inv, x, gw16)
        z, key = z.float(), key.float()
    else:
        z, key = seal.graph_conv_stack(g, d_inv, x, gw, batch=batch)      # dense-block MFMA stack, no CSR
    return seal.head(z, key, batch.node_off(), hw)[1], batch, g


s, batch, g = score()
torch.cuda.synchronize()
t0 = time.time()
it = 5
for _ in range(it):
    s, batch, g = score()
torch.cuda.synchronize()
dt = (time.time() - t0) / it
t0 = time.time()
for _ in range(it):
    run()
torch.cuda.synchronize()
dt_struct = (time.time() - t0) / it
print(json.dumps({"workload": f"synthetic collab-shaped SEAL: |V|={N} undirected |E|={E}, {B} candidate pairs, d={D} " + ("bf16 storage" if BF16 else "fp32"),
                  "tags_size": T, "total_vertices": int(batch.n.sum()), "block_csr_nnz": (g.nnz() if g is not None else None),
                  "capped_pairs": int(batch.capped.sum()), "ms_total": dt * 1e3, "ms_structure_and_inputs": dt_struct * 1e3,
                  "scored_edges_per_s": B / dt}))
