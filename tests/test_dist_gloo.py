"""CPU, world_size 2, gloo: the N > 1 path -- source-edge partition + all-reduce of the partial
node aggregates, row-block pipelined -- must reproduce the single-process result.  The per-block
SpMM is supplied by the fp64 oracle here (there is no GPU in this container); on the GPU the same
class runs the HIP CSR kernels and RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lpgnn_amd import dist as ldist
    from oracle import aggregation as agg
    rng = np.random.default_rng(0)                      # identical on every rank
    n, e, d = 203, int(os.environ.get("LPG_TEST_EDGES", "5000")), 12
    ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
    x = rng.standard_normal((n, d))
    lo, hi = ldist.source_range(n, world, rank)
    mine = ldist.partition_edges_by_source(torch.from_numpy(ei), n, world, rank).numpy()
    assert mine.shape[1] > 0 and mine[0].min() >= lo and mine[0].max() < hi

    def block_spmm(stage, rows, x_local):
        """partial aggregate of this rank's edges for the given (padded) destination rows."""
        rows = rows.numpy()
        pos = np.full(rows.max() + 2, -1, dtype=np.int64)
        pos[rows] = np.arange(rows.size)
        keep = pos[np.minimum(mine[1], rows.max() + 1)] >= 0
        keep &= mine[1] <= rows.max()
        rowptr, col, _, _ = agg.coo_to_csr(pos[mine[1][keep]], mine[0][keep] - lo, rows.size)
        return torch.from_numpy(agg.csr_spmm(rowptr, col, None, x_local.numpy()))

    rowptr, col, _, _ = agg.coo_to_csr(ei[1], ei[0], n)
    want = agg.csr_spmm(rowptr, col, None, x)
    errs = []
    for collective in ("reduce_scatter", "all_reduce", "sparse", "auto"):
        op = ldist.SourcePartitionedSpMM(torch.from_numpy(mine), n, (lo, hi), n_blocks=5, block_spmm=block_spmm,
                                         collective=collective)
        y = op.forward(torch.from_numpy(x[lo:hi])).numpy()
        if collective in ("sparse", "auto"):
            assert 0.0 < op.nonempty_fraction <= 1.0
            assert op.collective == ("sparse" if collective == "sparse" or op.nonempty_fraction <= 0.75 else "reduce_scatter")
            if e <= 150:
                assert op.collective == "sparse" and op.nonempty_fraction < 0.5
            if op.collective == "sparse":        # on the wire: exactly the destination rows this rank's edges reach
                assert sum(sum(sp) for sp in op.in_splits) == np.unique(mine[1]).size
            y2 = op.forward(torch.from_numpy(x[lo:hi]), collective="reduce_scatter").numpy()
            assert np.abs(y - y2).max() < 1e-12
        if collective != "all_reduce":           # this rank's (padded) shard of the destination rows
            r0 = rank * op.shard
            ref = np.zeros((op.shard, d))
            ref[: max(0, min(n, r0 + op.shard) - r0)] = want[r0:r0 + op.shard]
            errs.append(float(np.abs(y - ref).max()))
        else:
            errs.append(float(np.abs(y - want).max()))
    # bf16 partials on the wire (SURVEY 2b C1 "bf16 option"): the rows travel rounded to bf16 and are summed into an fp32
    # shard; bytes on the links halve, the result is within one bf16 rounding of every exchanged row
    op16 = ldist.SourcePartitionedSpMM(torch.from_numpy(mine), n, (lo, hi), n_blocks=5, block_spmm=block_spmm,
                                       collective="sparse", wire_dtype=torch.bfloat16)
    x32 = torch.from_numpy(x[lo:hi]).float()
    y16 = op16.forward(x32).double().numpy()
    y32 = op16.forward(x32, collective="reduce_scatter").double().numpy()
    mag = np.zeros((op16.shard, d))
    r0 = rank * op16.shard
    rp_, cl_, _, _ = agg.coo_to_csr(ei[1], ei[0], n)
    mag[: max(0, min(n, r0 + op16.shard) - r0)] = agg.csr_spmm(rp_, cl_, None, np.abs(x))[r0:r0 + op16.shard]
    assert y16.dtype == np.float64 and np.all(np.abs(y16 - y32) <= 2.0 ** -8 * mag + 1e-6)
    assert op16.bytes_on_links_per_step(d, torch.float32) * 2 == \
        ldist.SourcePartitionedSpMM(torch.from_numpy(mine), n, (lo, hi), n_blocks=5, block_spmm=block_spmm,
                                    collective="sparse").bytes_on_links_per_step(d, torch.float32)
    dense = op16.bytes_on_links_per_step(d, torch.float32, collective="reduce_scatter")
    assert dense == op16.n_stages * op16.sub * world * d * 4 * (world - 1) // world
    q.put((rank, max(errs), mine.shape[1]))
    dist.destroy_process_group()


def test_source_partition_sparse_graph_world2(monkeypatch):
    """few edges: most rows of a partial are empty, `auto` must pick the sparse exchange and agree with the dense one."""
    monkeypatch.setenv("LPG_TEST_EDGES", "150")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[2] for r in res) == 150
    for _, err, _ in res:
        assert err < 1e-12


def test_source_partition_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[2] for r in res) == 5000                 # every edge owned by exactly one rank
    for _, err, _ in res:
        assert err < 1e-12


def _score_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lpgnn_amd import dist as ldist
    seen = []

    def score_fn(p):                                     # stands in for the single-GPU SEAL scorer
        seen.append(p.clone())
        return (p[:, 0] * 1000 + p[:, 1]).to(torch.float32)

    out = {}
    for B in (11, 8, 1, 0):
        pairs = torch.arange(2 * B, dtype=torch.int64).view(B, 2)
        sc = ldist.ShardedLinkScorer(score_fn)
        got = sc(pairs)
        out[B] = (got.tolist(), sc.my_slice(B), sc(pairs, gather=False).tolist())
    q.put((rank, out))
    dist.destroy_process_group()


def test_scored_edge_sharding_world2():
    """8e (i): candidate pairs split across ranks, graph replicated, only the [B] scores gathered."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_score_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for B in (11, 8, 1, 0):
        want = [float(2 * i * 1000 + 2 * i + 1) for i in range(B)]
        assert res[0][B][0] == want and res[1][B][0] == want          # every rank ends with all scores, in order
        (a0, b0), (a1, b1) = res[0][B][1], res[1][B][1]
        assert a0 == 0 and b0 == a1 and b1 == B and abs((b0 - a0) - (b1 - a1)) <= 1
        assert res[0][B][2] == want[a0:b0] and res[1][B][2] == want[a1:b1]


def _combine_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    mine = {"scored_edges": 200, "ms_per_forward": 2.0 + rank, "scored_edges_per_s": 0.0}
    ok = bench.combine_over_ranks(mine, world, rank, timeout_s=30)
    mine2 = {"error": "boom"} if rank == 1 else {"scored_edges": 200, "ms_per_forward": 1.0, "scored_edges_per_s": 0.0}
    dist.barrier()
    from torch.distributed import distributed_c10d as c10d
    store = c10d._get_default_store()
    store.set(f"lpg_seal_sharded_{rank}", __import__("json").dumps(mine2))       # second round: rank 1 reports a failure
    dist.barrier()
    bad = bench.combine_over_ranks(mine2, world, rank, timeout_s=30)
    q.put((rank, ok, bad))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_combines_secondary_figure_through_the_store_world2():
    """bench.py's sharded-SEAL figure is combined without NCCL collectives: slowest rank wins, a failed rank
    turns into an error field instead of a hang."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_combine_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (ok, bad) for r, ok, bad in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ok0, bad0 = res[0]
    assert ok0["ms_per_forward"] == 3.0 and abs(ok0["scored_edges_per_s"] - 200 / 3e-3) < 1e-6
    assert "error" in bad0 and bad0["ranks_failed"] == 1


def _gpu_worker(rank, world, port, q):
    """two real ranks on ONE GPU (gloo as the transport, HIP kernels for everything else): the packed-row SpMM, the
    sparse exchange's plan and lpg_rows_add_f32 under a real source partition, against the unpartitioned SpMM."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lpgnn_amd
    from lpgnn_amd import dist as ldist
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7)               # identical on every rank
    n, e, d = 30_011, 90_000, 32
    ei = torch.randint(0, n, (2, e), generator=gen)
    ei[1, : e // 2] = torch.randint(0, n // 50, (e // 2,), generator=gen)      # many rows receive nothing
    x = torch.randn((n, d), generator=gen)
    lo, hi = ldist.source_range(n, world, rank)
    mine = ldist.partition_edges_by_source(ei, n, world, rank).to(dev)
    full = lpgnn_amd.CSRGraph.from_edge_index(ei.to(dev), n, n).spmm(x.to(dev))
    out = {}
    for collective in ("sparse", "reduce_scatter"):
        op = ldist.SourcePartitionedSpMM(mine, n, (lo, hi), n_blocks=3, collective=collective)
        y = op.forward(x[lo:hi].to(dev))
        r0 = rank * op.shard
        ref = torch.zeros((op.shard, d), device=dev)
        ref[: max(0, min(n, r0 + op.shard) - r0)] = full[r0:r0 + op.shard]
        out[collective] = (float((y - ref).abs().max()), float(ref.abs().max()), op.nonempty_fraction, bool(op.graphs_packed))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_exchange_two_ranks_on_the_gpu():
    import pytest
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):
        for collective in ("sparse", "reduce_scatter"):
            err, scale, frac, packed = res[rank][collective]
            assert err <= 1e-5 * (scale + 1.0), (rank, collective, err)
            assert frac < 0.75 and packed == (collective == "sparse")


test_sparse_exchange_two_ranks_on_the_gpu = __import__("pytest").mark.gpu(test_sparse_exchange_two_ranks_on_the_gpu)


def test_ranges_cover_and_balance():
    from lpgnn_amd import dist as ldist
    for n, w in [(10, 3), (5_000_000, 8), (7, 8), (1, 1)]:
        r = [ldist.source_range(n, w, k) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(w - 1))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1
    assert ldist.row_blocks(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert ldist.shard_layout(5_000_000, 8, 4) == (625_000, 156_250)
    assert ldist.shard_layout(203, 2, 5) == (105, 21)


def _twowl_shard_worker(rank, world, port, q):
    """two processes on GPU 0 over gloo: dist.ShardedTwoWLScorer against the one-GPU forward over all candidates"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lpgnn_amd import dist as ldist, twowl
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(9)                    # identical on every rank
    n = 2500
    pe = twowl.double(torch.randint(0, n, (2, 15_000), generator=g).to(dev))
    pr = twowl.double(torch.randint(0, n, (2, 777), generator=g).to(dev))
    xdeg = twowl.degree(pe, n)
    torch.manual_seed(1)
    mod = twowl.LocalWLNet(int(xdeg.max()), channels_1wl=24, channels_2wl=16, depth2=3).to(dev).eval()
    with torch.no_grad():
        full = mod.score_tuple_graph(xdeg, pe, twowl.tuple_graph(n, pe, pr), pe.shape[1] + torch.arange(pr.shape[1], device=dev))
    sc = ldist.ShardedTwoWLScorer(mod, n, pe, pr)
    mine = sc(xdeg)
    # without the exchange the GraphNorm statistics would cover this rank's candidates only: a different function
    sc.collective = False
    alone = sc(xdeg)
    q.put((rank, float((mine - full[sc.lo:sc.hi]).abs().max()), float((alone - full[sc.lo:sc.hi]).abs().max()),
           float(full.abs().max()), sc.hi - sc.lo))
    dist.barrier()
    dist.destroy_process_group()


def test_twowl_candidate_sharding_world2_on_the_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_twowl_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sum(r[4] for r in res) == 777
    for rank, err, err_alone, scale, _ in res:
        assert err <= 2e-5 * scale, (rank, err, scale)
        assert err_alone > 10 * err            # the all-reduce of the statistics is what makes the shards agree


test_twowl_candidate_sharding_world2_on_the_gpu = __import__("pytest").mark.gpu(test_twowl_candidate_sharding_world2_on_the_gpu)
