#!/usr/bin/env python3
"""Where does seal_subgraph_kernel's time go?  hop 0 (fixed cost), hop 1 on low-degree pairs, hop 1 on all."""
import os, sys, time, torch
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
deg = (graph.rowptr[1:] - graph.rowptr[:-1])
def T(fn, it=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r
t0, _ = T(lambda: seal.extract_subgraphs(pairs, graph, 0, cap_policy="device", seed=0))
t1, b1 = T(lambda: seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0))
small = pairs[(deg[pairs[:, 0]] + deg[pairs[:, 1]]) <= 8]
small = small[torch.arange(B, device=dev) % small.shape[0]]
t2, b2 = T(lambda: seal.extract_subgraphs(small, graph, 1, cap_policy="device", seed=0))
n = b1.n.long()
print(f"hop0 {t0:.3f} ms | hop1 all {t1:.3f} ms (mean n {n.float().mean():.1f}, n>=64: {(n >= 64).sum().item()}, capped {int(b1.capped.sum())}) | "
      f"hop1 small-degree pairs {t2:.3f} ms (mean n {b2.n.float().mean():.1f})")
for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 64), (64, 129)):
    m = (n >= lo) & (n < hi)
    if m.sum() == 0: continue
    sub = pairs[m]; sub = sub[torch.arange(B, device=dev) % sub.shape[0]]
    tt, _ = T(lambda: seal.extract_subgraphs(sub, graph, 1, cap_policy="device", seed=0), it=5)
    print(f"  n in [{lo},{hi}): {int(m.sum())} pairs of the batch; a batch of {B} such pairs takes {tt:.3f} ms")
