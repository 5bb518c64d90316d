#!/usr/bin/env python3
"""extraction on the ppa-shaped graph of BASELINE configs[3] (|V| = 576 289, 30.3 M undirected edges, mean degree 105:
nearly every 1-hop subgraph hits the 100-vertex cap): 50 K candidate pairs, and the same against hub_degree (first line: the library's own choice; last line: no hub matrix at all)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from lpgnn_amd import seal
dev = torch.device("cuda:0")
N, E, B = 576_289, 30_326_273, 50_000
ei = bench.rmat_edges(N, E, 3, dev)
gen = torch.Generator(device=dev); gen.manual_seed(4)
pairs = torch.randint(0, N, (B, 2), generator=gen, device=dev)
def T(fn, it=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
p = ei.t().contiguous()
for cand, cap in ((seal.TrainGraph.HUB_DEGREES, seal.TrainGraph.MAX_HUBS), ((96,), 1 << 18), ((128,), 1 << 18), ((192,), 1 << 18), ((256,), 1 << 18), ((512,), 1 << 18), ((1024,), 1 << 18), ((1 << 20,), 1)):
    seal.TrainGraph.HUB_DEGREES, seal.TrainGraph.MAX_HUBS = cand, cap
    t0 = time.perf_counter()
    graph = seal.build_train_csr(p, N, test_ratio=0.0, device=dev)
    torch.cuda.synchronize(); tb = time.perf_counter() - t0
    H = 0 if graph.hub_bits is None else graph.hub_bits.shape[0]
    t1 = T(lambda: seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0))
    b = seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0)
    print(f"candidates {cand}, at most {cap} hubs: hub_degree {graph.hub_degree}, {H} hubs ({H * H / 8 / 1e6:.1f} MB), graph build {tb * 1e3:.0f} ms; "
          f"{B} pairs {t1:.2f} ms = {B / t1 / 1e3:.1f} M pairs/s (mean n {b.n.float().mean():.1f}, capped {int(b.capped.sum())})", flush=True)
