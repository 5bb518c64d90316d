#!/usr/bin/env python3
"""extraction of 100 K collab-shaped candidate pairs against hub_degree (neighbour lists longer than it are not
streamed; their nodes get rows in the hub bit matrix) with MAX_HUBS raised so that every threshold has its matrix"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from lpgnn_amd import seal
dev = torch.device("cuda:0")
N, E, B = 235_868, 1_285_465, 100_000
ei = bench.rmat_edges(N, E, 1, dev)
gen = torch.Generator(device=dev); gen.manual_seed(2)
pairs = torch.randint(0, N, (B, 2), generator=gen, device=dev)
def T(fn, it=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
seal.TrainGraph.MAX_HUBS = 1 << 15
for th in (32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 224, 256, 384):
    seal.TrainGraph.HUB_DEGREES = (th,)
    graph = seal.build_train_csr(ei.t().contiguous(), N, test_ratio=0.0, device=dev)
    H = 0 if graph.hub_bits is None else graph.hub_bits.shape[0]
    t1 = T(lambda: seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0))
    print(f"hub_degree = {graph.hub_degree:3d}: {H:5d} hubs, matrix {H * H / 8 / 1e6:6.2f} MB, hop 1 {t1:.3f} ms", flush=True)
