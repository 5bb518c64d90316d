#!/usr/bin/env python3
"""fb-pages-food SEAL structure stage by piece (host-bound: 4202 pairs)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpgnn_amd import seal
dev = "cuda"
z = np.load(os.path.join(ROOT, "tests", "golden", "seal_int.npz"))
pos, neg = z["positive"].astype(np.int64), z["negative"].astype(np.int64)
def T(fn, it=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r
t_all, _ = T(lambda: seal.link2subgraph(pos, neg, 620, 0.1, 1, device=dev))
t_csr, graph = T(lambda: seal.build_train_csr(pos, 620, 0.1, dev))
pairs = np.concatenate([neg, pos]).astype(np.int64)
t_h2d, pairs_t = T(lambda: torch.as_tensor(pairs, dtype=torch.int64, device=dev))
t_ext, batch = T(lambda: seal.extract_subgraphs(pairs_t, graph, 1))
t_tag, _ = T(lambda: seal.compact_tag_map(batch))
t_lab, _ = T(lambda: torch.cat([torch.zeros(len(neg), dtype=torch.uint8), torch.ones(len(pos), dtype=torch.uint8)]))
print(f"link2subgraph {t_all:.3f} ms = build_train_csr {t_csr:.3f} + pairs H2D {t_h2d:.3f} + extract {t_ext:.3f} + tag map {t_tag:.3f} + labels {t_lab:.3f}")
