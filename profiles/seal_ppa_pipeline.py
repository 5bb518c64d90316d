#!/usr/bin/env python3
"""The SEAL scoring forward on the ppa-shaped graph of BASELINE configs[3] (|V| = 576 289, 30.3 M undirected edges, d = 58 padded to 64, 50 K candidates) as bench.py runs the collab-shaped one (extraction, the score stack, the dense layers:
no host read inside a forward), wall time per forward next to the sum of the stages timed alone (profiles/seal_stages.py).  Under
rocprofv3 --kernel-trace the trace shows the gaps between the kernels of one forward."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from lpgnn_amd import seal
dev = torch.device("cuda:0")
N, E, B, D, k = 576_289, 30_326_273, 50_000, 64, 18
ei = bench.rmat_edges(N, E, 3, dev)
graph = seal.build_train_csr(ei.t().contiguous(), N, test_ratio=0.0, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(2)
pairs = torch.randint(0, N, (B, 2), generator=gen, device=dev)
emb = torch.randn((N, D), generator=gen, device=dev)
tag_map, T = seal.compact_tag_map(seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0))
torch.manual_seed(0)
gw32 = [torch.randn(a, b, device=dev) * 0.1 for a, b in ((T + D, 32), (32, 32), (32, 32), (32, 1))]
hw = {"conv1d_kernel_1": torch.randn(96, 1, 16, device=dev) * 0.1, "conv1d_kernel_2": torch.randn(5, 16, 32, device=dev) * 0.1,
      "weight_1": torch.randn((k - 4) * 32, 128, device=dev) * 0.1, "bias_1": torch.zeros(128, device=dev),
      "weight_2": torch.randn(128, 2, device=dev), "bias_2": torch.zeros(2, device=dev)}
k1 = hw["conv1d_kernel_1"].reshape(-1, 16)
for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
    gw, e = [w.to(dt) for w in gw32], emb.to(dt)
    def fwd():
        batch = seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0)
        pooled = seal.score_stack(batch, tag_map, T, e, gw, k1, pool_k=k)
        return seal.head(None, None, batch.node_off(), hw, pooled=pooled, check=False)[1]
    for _ in range(3): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): s = fwd()
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 10
    print(f"{name}: {dtm * 1e3:.3f} ms per forward = {B / dtm / 1e6:.1f} M scored edges/s")
