"""float32 row scatter_add [E, d] -> [N, d] at scale: float atomics vs the destination-grouped route (first call with the
grouping, later calls with the memoised plan).  python profiles/scatter_add_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lpgnn_amd import ops
from lpgnn_amd.torch_scatter import scatter_add

dev = torch.device("cuda:0")


def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return min(out)


for E, N, d in [(10_000_000, 1_000_000, 128), (2_570_930, 235_868, 128), (50_000_000, 5_000_000, 32)]:
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    idx = torch.randint(0, N, (E,), generator=g, device=dev)
    src = torch.randn((E, d), generator=g, device=dev)
    alg = E * (8 + d * 4) + N * d * 4
    ops.SCATTER_SORT_MIN = 1 << 62
    t_atomic = timed(lambda: scatter_add(src, idx, dim=0, dim_size=N))
    ya = scatter_add(src, idx, dim=0, dim_size=N)
    ops.SCATTER_SORT_MIN = 1 << 20

    def first():
        ops._scatter_plans.clear()
        scatter_add(src, idx, dim=0, dim_size=N)
    t_first = timed(first, reps=3)
    t_again = timed(lambda: scatter_add(src, idx, dim=0, dim_size=N))
    yb = scatter_add(src, idx, dim=0, dim_size=N)
    err = float((ya - yb).abs().max() / ya.abs().max())
    print(f"scatter_add f32 E={E} N={N} d={d}: atomics {t_atomic:.3f} ms ({alg / t_atomic / 1e6:.0f} GB/s) | grouped, first call "
          f"{t_first:.3f} ms, later calls {t_again:.3f} ms ({alg / t_again / 1e6:.0f} GB/s, {alg / t_again / 8e9:.3f} of 8 TB/s); "
          f"max rel diff between the two {err:.2e}", flush=True)
    ops._scatter_plans.clear()
    del idx, src, ya, yb
