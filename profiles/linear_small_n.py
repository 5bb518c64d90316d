import os, sys, torch
sys.path.insert(0, os.getcwd())
from lpgnn_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (k, m) in [(288, 128), (128, 2), (256, 32), (106, 32), (448, 128), (128, 128)]:
    w = torch.randn(k, m, device=dev)
    for n in [386, 1024, 2048, 4096, 8192, 16384, 32768, 50000, 65536]:
        x = torch.randn(n, k, device=dev)
        print(f"k={k} m={m} n={n}: {timed(lambda: ops.linear(x, w, w_is_km=True)):.1f} us", flush=True)
