import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
N, E, d = 5_000_000, 100_000_000, 128
g, x = bench.make_rmat_problem(N, E, d, seed=5, device=dev)
b = torch.randn(d, device=dev)
for _ in range(2):
    g.spmm(x)
    g.spmm(x, act="relu")
    g.spmm(x, bias=b)
torch.cuda.synchronize()
