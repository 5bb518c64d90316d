"""One-time graph build at the headline size: COO -> CSR (+ GCN self loops), plan, gcn_norm, timed separately.
Run on the GPU box: python profiles/csr_build_time.py [edges] [nodes]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lpgnn_amd
from lpgnn_amd import _lib, ops
from bench import rmat_edges

E = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
dev = torch.device("cuda:0")
ei = rmat_edges(N, E, 5, dev)
lib = _lib.load()


def timed(fn, reps=3):
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


for loops in (0, 1):
    for weighted in (False, True):
        M = E + (N if loops else 0)
        rowptr = torch.empty(N + 1, dtype=torch.int64, device=dev)
        col = torch.empty(M, dtype=torch.int32, device=dev)
        w = torch.rand(E, device=dev) if weighted else None
        val = torch.empty(M, dtype=torch.float32, device=dev) if weighted else None
        nbytes = lib.lpg_coo_to_csr_bytes(E, N, loops)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        flag = ops.err_flag(dev)
        st = torch.cuda.current_stream().cuda_stream
        p = lambda t: 0 if t is None else t.data_ptr()
        ms = timed(lambda: lib.lpg_coo_to_csr(p(ei[0]), p(ei[1]), p(w), E, N, N, loops, 0, p(rowptr), p(col), p(val), 0,
                                              p(ws), nbytes, p(flag), st))
        print(f"coo_to_csr E={E} N={N} self_loops={loops} weighted={weighted}: {ms:.2f} ms "
              f"(workspace {nbytes / 1e9:.2f} GB)", flush=True)
        del ws, col, val, w
t = time.time()
g = lpgnn_amd.CSRGraph.from_edge_index(ei, N, N, gcn_self_loops=True)
g.gcn_normalize_()
torch.cuda.synchronize()
print(f"from_edge_index + plan + gcn_norm, wall incl. allocation and the one host sync: {(time.time() - t) * 1e3:.1f} ms")
