"""Where a workgroup of the SEAL conv stack spends a 128-row window (a -DLPG_TUNING build): prologue (bounds, per-row
tables, the two table gathers, plane split) / the four layers / the SortPooling epilogue; collab-shaped graph, uniform
candidates and the reference's mix, bf16 and fp32 storage."""
import ctypes
import os
import sys
import time

import torch

os.environ.setdefault("LPG_LIB_VARIANT", "tuning")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lpgnn_amd import _lib, seal  # noqa: E402

dev = torch.device("cuda:0")
N, E, B, D, k = 235_868, 1_285_465, 100_000, 128, 18
ei = bench.rmat_edges(N, E, 1, dev)
graph = seal.build_train_csr(ei.t().contiguous(), N, test_ratio=0.0, device=dev)
gen = torch.Generator(device=dev)
gen.manual_seed(2)
uniform = torch.randint(0, N, (B, 2), generator=gen, device=dev)
emb = torch.randn((N, D), generator=gen, device=dev)
pick = torch.randint(0, E, (B // 2,), generator=gen, device=dev)
refmix = torch.cat([torch.randint(0, N, (B - B // 2, 2), generator=gen, device=dev), ei[:, pick].t()], 0).contiguous()
torch.manual_seed(0)
k1 = (torch.randn(96, 1, 16, device=dev) * 0.1).reshape(-1, 16)
w_rest = [torch.randn(a, b, device=dev) * 0.1 for a, b in ((32, 32), (32, 32), (32, 1))]
lib = _lib.load()
names = ["prologue", "layers", "sortpool epilogue"]
for label, pairs in (("uniform", uniform), ("refmix", refmix)):
    batch = seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0)
    tag_map, T = seal.compact_tag_map(batch)
    gw32 = [torch.randn(T + D, 32, device=dev) * 0.1] + w_rest
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        gw, e = [w.to(dt) for w in gw32], emb.to(dt)
        out = (ctypes.c_uint64 * 12)()
        seal.score_stack(batch, tag_map, T, e, gw, k1, pool_k=k)
        _lib.check(lib.lpg_seal_conv_phase_ticks(out, 1), "conv_phase_ticks")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            seal.score_stack(batch, tag_map, T, e, gw, k1, pool_k=k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        _lib.check(lib.lpg_seal_conv_phase_ticks(out, 1), "conv_phase_ticks")
        wins = max(out[3], 1)
        pro = out[4] + out[5] + out[7]
        lay = out[0] + out[6] + out[1]
        epi = out[8] + out[9] + out[2]
        sub = list(out)
        out[0], out[1], out[2] = pro, lay, epi
        tot = pro + lay + epi
        print(f"{label} {name}: {ms:.3f} ms per launch, {out[3] / 5:.0f} windows per launch, {tot / wins / 100:.2f} us per window "
              f"in its workgroup")
        for j in range(3):
            print(f"    {names[j]:18s} {out[j] / wins / 100:7.2f} us per window  {100.0 * out[j] / max(tot, 1):5.1f} %")
        print("    layer steps: " + ", ".join(f"{n} {sub[j] / wins / 100:.2f}" for n, j in (("aggregate + dense (to the barrier)", 0), ("plane writes", 6), ("layer 4", 1))))
        print("    epilogue steps: " + ", ".join(f"{n} {sub[j] / wins / 100:.2f}" for n, j in (("rows + keys to LDS", 8), ("ranks", 9), ("stores", 2))))
        print("    prologue steps: " + ", ".join(f"{n} {sub[j] / wins / 100:.2f}" for n, j in (("bounds + offsets", 4), ("row table + tags / nodes", 5), ("table gathers + planes", 7))))
