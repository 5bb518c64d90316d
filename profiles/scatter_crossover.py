"""Where the destination-grouped float32 scatter_add overtakes float atomics: E = 2^13 .. 2^21 rows, N = E / 10, d = 128 /
32 / 8; atomics vs grouped (first call, with the grouping; later calls, plan memoised).  Sets ops.SCATTER_SORT_MIN."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lpgnn_amd import ops
from lpgnn_amd.torch_scatter import scatter_add

dev = torch.device("cuda:0")


def timed(fn, reps=7):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return sorted(out)[len(out) // 2]


for d in (128, 32, 8):
    for lg in range(13, 22):
        E = 1 << lg
        N = max(E // 10, 16)
        g = torch.Generator(device=dev)
        g.manual_seed(lg)
        idx = torch.randint(0, N, (E,), generator=g, device=dev)
        src = torch.randn((E, d), generator=g, device=dev)
        ops.SCATTER_SORT_MIN = 1 << 62
        t_atomic = timed(lambda: scatter_add(src, idx, dim=0, dim_size=N))
        ops.SCATTER_SORT_MIN = 1

        def first():
            ops._scatter_plans.clear()
            scatter_add(src, idx, dim=0, dim_size=N)
        t_first = timed(first, reps=5)
        t_again = timed(lambda: scatter_add(src, idx, dim=0, dim_size=N))
        print(f"d={d:4d} E=2^{lg}: atomics {t_atomic * 1e3:8.1f} us | grouped first call {t_first * 1e3:8.1f} us, later calls "
              f"{t_again * 1e3:8.1f} us", flush=True)
        ops._scatter_plans.clear()
