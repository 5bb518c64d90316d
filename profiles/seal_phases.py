"""Where the 128-thread teams of seal_subgraph_kernel spend their time, per phase (a -DLPG_TUNING build:
LPG_LIB_VARIANT=tuning python profiles/seal_phases.py), on the collab-shaped graph with uniform candidates and with the
reference's mix (half true edges)."""
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
N, E, B = 235_868, 1_285_465, 100_000
ei = bench.rmat_edges(N, E, 1, dev)
graph = seal.build_train_csr(ei.t().contiguous(), N, test_ratio=0.0, device=dev)
gen = torch.Generator(device=dev)
gen.manual_seed(2)
uniform = torch.randint(0, N, (B, 2), generator=gen, device=dev)
_ = torch.randn((N, 128), generator=gen, device=dev)
pick = torch.randint(0, E, (B // 2,), generator=gen, device=dev)
refmix = torch.cat([torch.randint(0, N, (B - B // 2, 2), generator=gen, device=dev), ei[:, pick].t()], 0).contiguous()
lib = _lib.load()
names = ["node set", "rows + filter", "streamed lists", "hub bit matrix", "searches", "DRNL", "outputs"]
for label, pairs in (("uniform", uniform), ("refmix", refmix)):
    out = (ctypes.c_uint64 * 8)()
    seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0)
    _lib.check(lib.lpg_seal_phase_ticks(out, 1), "phase_ticks")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        b = seal.extract_subgraphs(pairs, graph, 1, cap_policy="device", seed=0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    _lib.check(lib.lpg_seal_phase_ticks(out, 1), "phase_ticks")
    teams = out[7] / 5
    tot = sum(out[:7])
    print(f"{label}: {ms:.3f} ms per launch (instrumented), {teams:.0f} team pairs per launch, mean n {b.n.float().mean():.1f}; "
          f"team time per pair {tot / max(out[7], 1) / 100:.1f} us")
    for k in range(7):
        print(f"    {names[k]:16s} {out[k] / max(out[7], 1) / 100:7.2f} us per team pair  {100.0 * out[k] / max(tot, 1):5.1f} %")
