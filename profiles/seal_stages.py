#!/usr/bin/env python3
"""Stage-by-stage time of the collab-shaped SEAL scoring forward (bench.py's seal_sharded workload: RMAT |V| = 235 868,
100 K candidate pairs, d = 128), fp32 and bf16 storage, on the fast path (tables + dense-block stack + fused conv1d #1)
and, with --generic, layer by layer over the block CSR.  --variant N sets option seal_conv_variant (1 = the r01 kernel)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from lpgnn_amd import _lib, seal
dev = torch.device("cuda:0")
N, E, B, D = 235_868, 1_285_465, 100_000, 128
ei = bench.rmat_edges(N, E, 1, dev)
graph = seal.build_train_csr(ei.t().contiguous(), N, test_ratio=0.0, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(2)
pairs = torch.randint(0, N, (B, 2), generator=gen, device=dev)
emb = torch.randn((N, D), generator=gen, device=dev)
k = 18
if "--window" in sys.argv:
    seal.CONV_WINDOW = int(sys.argv[sys.argv.index("--window") + 1])
if "--variant" in sys.argv:
    _lib.set_option("seal_conv_variant", int(sys.argv[sys.argv.index("--variant") + 1]))
def T(fn, it=10):
    keep = [fn(), fn()]                    # two live results: the allocator's second block exists before the clock starts
    del keep
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r
t_ext, batch = T(lambda: seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0))
t_map, (tag_map, Tn) = T(lambda: seal.compact_tag_map(batch))
torch.manual_seed(0)
gw32 = [torch.randn(a, b, device=dev) * 0.1 for a, b in ((Tn + D, 32), (32, 32), (32, 32), (32, 1))]
hw = {"conv1d_kernel_1": torch.randn(96, 1, 16, device=dev) * 0.1, "conv1d_kernel_2": torch.randn(5, 16, 32, device=dev) * 0.1,
      "weight_1": torch.randn((k - 4) * 32, 128, device=dev) * 0.1, "bias_1": torch.zeros(128, device=dev),
      "weight_2": torch.randn(128, 2, device=dev), "bias_2": torch.zeros(2, device=dev)}
k1 = hw["conv1d_kernel_1"].reshape(-1, 16)
print(f"extract {t_ext:.3f}  tag_map {t_map:.3f} ms; total vertices {int(batch.n.sum())}")
for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    gw, e = [w.to(dt) for w in gw32], emb.to(dt)
    t_h1, h1 = T(lambda: seal.first_layer_h1(batch, tag_map, Tn, e, gw[0], check=False))
    t_cat, _ = T(lambda: seal.graph_conv_stack(None, None, None, gw, batch=batch, h1=h1))
    t_conv, (a16, key) = T(lambda: seal.graph_conv_stack(None, None, None, gw, batch=batch, h1=h1, head_k1=k1))
    t_head, _ = T(lambda: seal.head(None, key, batch.node_off(), hw, a_all=a16))
    tot = t_ext + t_h1 + t_conv + t_head
    print(f"{name}: first_layer_h1 {t_h1:.3f}  dense stack + conv1d #1 {t_conv:.3f} (writing the concat instead: {t_cat:.3f})  head {t_head:.3f} ms"
          f"  => extract + h1 + stack + head = {tot:.3f} ms = {B / tot / 1e3:.1f} M scored edges/s")
    t_fused, pooled = T(lambda: seal.score_stack(batch, tag_map, Tn, e, gw, k1, pool_k=k))
    t_head2, _ = T(lambda: seal.head(None, None, batch.node_off(), hw, pooled=pooled))
    tot2 = t_ext + t_fused + t_head2
    print(f"{name} fused ends: tables + stack + conv1d #1 + SortPooling {t_fused:.3f}  dense layers {t_head2:.3f} ms"
          f"  => extract + stack + head = {tot2:.3f} ms = {B / tot2 / 1e3:.1f} M scored edges/s")
    if "--generic" in sys.argv:
        t_in, (g, d_inv, x) = T(lambda: seal.gnn_inputs(batch, tag_map, Tn, e.float(), dtype=dt))
        t_gen, _ = T(lambda: seal.graph_conv_stack(g, d_inv, x, gw))
        print(f"{name} layer by layer: gnn_inputs {t_in:.3f}  4 x (linear + SpMM) {t_gen:.3f} ms, nnz {g.nnz()}")
