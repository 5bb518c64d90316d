#!/usr/bin/env python3
"""Stage-by-stage time of the collab-shaped SEAL scoring forward (profiles/seal_collab.py's workload)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from lpgnn_amd import seal
dev = torch.device("cuda:0")
N, E, B, D = 235_868, 1_285_465, 100_000, 128
ei = bench.rmat_edges(N, E, 1, dev)
graph = seal.build_train_csr(ei.t().cpu().numpy(), N, test_ratio=0.0, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(2)
pairs = torch.randint(0, N, (B, 2), generator=gen, device=dev)
emb = torch.randn((N, D), generator=gen, device=dev)
k = 18
def T(fn, it=10):
    keep = [fn(), fn()]                    # two live results: the allocator's second block exists before the clock starts
    del keep                               # (a 562 MB hipMalloc inside the loop read as +1.5 ms per call)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r
t_ext, batch = T(lambda: seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0))
t_map, (tag_map, Tn) = T(lambda: seal.compact_tag_map(batch))
BF16 = "--bf16" in sys.argv
DT = torch.bfloat16 if BF16 else torch.float32
GENERIC = "--generic" in sys.argv or BF16        # bf16 storage has no dense-block kernel
t_in, (g, d_inv, x) = T(lambda: seal.gnn_inputs(batch, tag_map, Tn, emb, dtype=DT, build_csr=GENERIC))
torch.manual_seed(0)
gw = [torch.randn(a, b, device=dev) * 0.1 for a, b in ((Tn + D, 32), (32, 32), (32, 32), (32, 1))]
hw = {"conv1d_kernel_1": torch.randn(96, 1, 16, device=dev) * 0.1, "conv1d_kernel_2": torch.randn(5, 16, 32, device=dev) * 0.1,
      "weight_1": torch.randn((k - 4) * 32, 128, device=dev) * 0.1, "bias_1": torch.zeros(128, device=dev),
      "weight_2": torch.randn(128, 2, device=dev), "bias_2": torch.zeros(2, device=dev)}
gw = [w.to(DT) for w in gw]
t_conv, (z, key) = T(lambda: seal.graph_conv_stack(g, d_inv, x, gw, batch=None if GENERIC else batch))
if not GENERIC:
    t_h1, h1 = T(lambda: seal.first_layer_h1(batch, tag_map, Tn, emb, gw[0]))
    t_conv2, _ = T(lambda: seal.graph_conv_stack(None, None, None, gw, batch=batch, h1=h1))
    print(f"table path: first_layer_h1 {t_h1:.3f} + dense stack {t_conv2:.3f} ms (replaces gnn_inputs + conv_stack)")
    if os.environ.get("LPG_SEAL_CONV_FP32", "0") == "0":
        k1 = hw["conv1d_kernel_1"].reshape(-1, 16)
        t_conv3, (a16, key3) = T(lambda: seal.graph_conv_stack(None, None, None, gw, batch=batch, h1=h1, head_k1=k1))
        t_head3, _ = T(lambda: seal.head(None, key3, batch.node_off(), hw, a_all=a16))
        print(f"fused path: dense stack + conv1d #1 {t_conv3:.3f} ms, head without it {t_head3:.3f} ms")
z, key = z.float(), key.float()
off = batch.node_off()
t_head, _ = T(lambda: seal.head(z, key, off, hw))
t_off, _ = T(lambda: batch.node_off())
print(("bf16 " if BF16 else "fp32 ") + f"extract {t_ext:.3f}  tag_map {t_map:.3f}  gnn_inputs {t_in:.3f}  conv_stack {t_conv:.3f}  head {t_head:.3f}  (node_off {t_off:.3f}) ms; "
      f"total vertices {int(batch.n.sum())}" + (f", nnz {g.nnz()}" if g is not None else ", dense conv stack"))
