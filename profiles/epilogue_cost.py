import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
N, E, d = 5_000_000, 100_000_000, 128
g, x = bench.make_rmat_problem(N, E, d, seed=5, device=dev)
rs = torch.rand(N, device=dev) + 0.5
b = torch.randn(d, device=dev)
w = torch.rand(g.e_cap, device=dev)
def t(**kw):
    for _ in range(3): g.spmm(x, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.spmm(x, **kw)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / 10, 3)
print("plain", t())
print("bias", t(bias=b))
print("bias+relu", t(bias=b, act="relu"))
print("scale+tanh", t(row_scale=rs, act="tanh"))
g.val = w
print("weighted plain", t())
print("weighted bias+relu", t(bias=b, act="relu"))
print("weighted scale+tanh", t(row_scale=rs, act="tanh"))
g.val = None
print("relu only (no bias, no scale)", t(act="relu"))
print("scale only", t(row_scale=rs))
