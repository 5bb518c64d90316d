"""SEAL scoring forward of bench.py's refmix / uniform sets, one stream against chunks alternating over two streams
(extraction of chunk i + 1 beside the conv stack of chunk i)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lpgnn_amd import seal  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
N, E, D, k = 235_868, 1_285_465, 128, 18
ei = bench.rmat_edges(N, E, 1, dev)
graph = seal.build_train_csr(ei.t().cpu().numpy(), N, test_ratio=0.0, device=dev)
gen = torch.Generator(device=dev)
gen.manual_seed(2)
B = 100_000
uniform = torch.randint(0, N, (B, 2), generator=gen, device=dev)
emb = torch.randn((N, D), generator=gen, device=dev)
pick = torch.randint(0, E, (B // 2,), generator=gen, device=dev)
refmix = torch.cat([torch.randint(0, N, (B - B // 2, 2), generator=gen, device=dev), ei[:, pick].t()], 0).contiguous()
refmix = refmix[torch.randperm(B, generator=gen, device=dev)]      # mixed order: every chunk holds the same mix
torch.manual_seed(0)
hw = {"conv1d_kernel_1": torch.randn(96, 1, 16, device=dev) * 0.1, "conv1d_kernel_2": torch.randn(5, 16, 32, device=dev) * 0.1,
      "weight_1": torch.randn((k - 4) * 32, 128, device=dev) * 0.1, "bias_1": torch.zeros(128, device=dev),
      "weight_2": torch.randn(128, 2, device=dev), "bias_2": torch.zeros(2, device=dev)}
k1 = hw["conv1d_kernel_1"].reshape(-1, 16)
w_rest = [torch.randn(a, b, device=dev) * 0.1 for a, b in ((32, 32), (32, 32), (32, 1))]
out = {}
for label, pairs in (("uniform", uniform), ("refmix", refmix)):
    probe = seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0)
    tag_map, T = seal.compact_tag_map(probe)
    del probe
    gw32 = [torch.randn(T + D, 32, device=dev) * 0.1] + w_rest
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        gw, e = [w.to(dt) for w in gw32], emb.to(dt)

        def score_fn(p):
            batch = seal.extract_subgraphs(p, graph, 1, cap_policy="device", seed=0)
            pooled = seal.score_stack(batch, tag_map, T, e, gw, k1, pool_k=k)
            return seal.head(None, None, batch.node_off(), hw, pooled=pooled, check=False)[1]

        streams = [torch.cuda.Stream(), torch.cuda.Stream()]

        def chunked(chunks):
            main = torch.cuda.current_stream()
            parts = []
            bounds = [B * i // chunks for i in range(chunks + 1)]
            for s in streams:
                s.wait_stream(main)
            for i in range(chunks):
                s = streams[i % 2]
                with torch.cuda.stream(s):
                    parts.append(score_fn(pairs[bounds[i]:bounds[i + 1]]))
            for s in streams:
                main.wait_stream(s)
            for p in parts:
                p.record_stream(main)
            return torch.cat(parts)

        def timed(fn, iters=5):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / iters * 1e3, r

        base, r0 = timed(lambda: score_fn(pairs))
        row = {"one_stream_ms": base}
        for c in (2, 4, 8):
            ms, r = timed(lambda: chunked(c))
            row[f"two_streams_{c}_chunks_ms"] = ms
            row[f"max_diff_{c}"] = float((r - r0).abs().max())
        out[f"{label}_{name}"] = row
print(json.dumps(out, indent=1))
