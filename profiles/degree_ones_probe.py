import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from lpgnn_amd import ops
from lpgnn_amd.torch_scatter import scatter_add
dev = "cuda:0"
N, E = 5_000_000, 100_000_000
ei = bench.rmat_edges(N, E, 5, dev)
idx = ei[1].contiguous()
ones = torch.ones_like(idx)
a = bench._time_ms(lambda: ops.degree(idx, N), 10)
b = bench._time_ms(lambda: scatter_add(ones, idx, dim_size=N), 10)
assert torch.equal(ops.degree(idx, N), scatter_add(ones, idx, dim_size=N))
print(f"degree() {a:.3f} ms; scatter_add(ones_like(idx), idx) {b:.3f} ms  (|E| = 100 M, |V| = 5 M)")
