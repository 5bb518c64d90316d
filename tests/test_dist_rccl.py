"""GPU: the RCCL (backend "nccl") branches of lpgnn_amd/dist.py -- reduce_scatter_tensor, all_reduce,
all_to_all_single with variable splits (async work + the staging ring), all_gather_into_tensor -- which the gloo tests
cannot reach (gloo takes the emulation branches).  On the one-GPU box: a process group of world size 1 over RCCL in a
child process, the HIP kernels for everything else, against the unpartitioned SpMM.  With two or more visible GPUs the
same worker also runs as two real ranks over xGMI.  Also: `python bench.py --gpus 2` starts its own ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rccl_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import lpgnn_amd
    from lpgnn_amd import dist as ldist
    gen = torch.Generator().manual_seed(11)              # identical on every rank
    n, e, d = 40_003, 150_000, 64
    ei = torch.randint(0, n, (2, e), generator=gen)
    ei[1, : e // 2] = torch.randint(0, n // 40, (e // 2,), generator=gen)      # many destination rows receive nothing
    x = torch.randn((n, d), generator=gen)
    lo, hi = ldist.source_range(n, world, rank)
    mine = ldist.partition_edges_by_source(ei, n, world, rank).to(dev)
    full = lpgnn_amd.CSRGraph.from_edge_index(ei.to(dev), n, n).spmm(x.to(dev))
    out = {"backend": dist.get_backend()}
    for collective in ("reduce_scatter", "all_reduce", "sparse"):
        op = ldist.SourcePartitionedSpMM(mine, n, (lo, hi), n_blocks=5, collective=collective)   # 5 stages > the 3-deep ring
        for rep in range(2):                              # twice: the staging ring and its pending work are reused
            y = op.forward(x[lo:hi].to(dev))
        if collective == "all_reduce":
            ref = full
        else:
            r0 = rank * op.shard
            ref = torch.zeros((op.shard, d), device=dev)
            ref[: max(0, min(n, r0 + op.shard) - r0)] = full[r0:r0 + op.shard]
        out[collective] = (float((y - ref).abs().max()), float(ref.abs().max()), op.collective,
                           bool(op.graphs_packed), tuple(y.shape))

    # bf16 on the links: (a) fp32 features with wire_dtype = bfloat16 (lpg_rows_add_bf16 into an fp32 shard), (b) bf16
    # storage end to end (the bf16 SpMM writes the packed rows); both against the fp32 result within one bf16 rounding
    # of what each row sums, and the profiled step reports per-stage SpMM / exchange-wait times
    mag = lpgnn_amd.CSRGraph.from_edge_index(ei.to(dev), n, n).spmm(x.abs().to(dev))
    r0 = rank * op.shard
    magr = torch.zeros((op.shard, d), device=dev)
    magr[: max(0, min(n, r0 + op.shard) - r0)] = mag[r0:r0 + op.shard]
    ref = torch.zeros((op.shard, d), device=dev)
    ref[: max(0, min(n, r0 + op.shard) - r0)] = full[r0:r0 + op.shard]
    opw = ldist.SourcePartitionedSpMM(mine, n, (lo, hi), n_blocks=5, collective="sparse", wire_dtype=torch.bfloat16)
    opw.profile = True
    yw = opw.forward(x[lo:hi].to(dev))
    ok_w = bool(torch.all((yw - ref).abs() <= 2.0 ** -8 * magr + 1e-6)) and yw.dtype == torch.float32
    prof_ok = (opw.last_profile is not None and len(opw.last_profile["spmm_ms"]) == opw.n_stages
               and len(opw.last_profile["exchange_wait_ms"]) == opw.n_stages and min(opw.last_profile["spmm_ms"]) > 0)
    opb = ldist.SourcePartitionedSpMM(mine, n, (lo, hi), n_blocks=5, collective="sparse")
    yb = opb.forward(x[lo:hi].to(dev).bfloat16())
    ok_b = bool(torch.all((yb.float() - ref).abs() <= 3 * 2.0 ** -8 * magr + 1e-6)) and yb.dtype == torch.bfloat16
    out["bf16_wire"] = (ok_w, ok_b, prof_ok, opw.bytes_on_links_per_step(d, torch.float32),
                        opb.bytes_on_links_per_step(d, torch.float32))

    # TwoWL with the candidate tuples sharded over the ranks: GraphNorm's statistics over the non-member rows are
    # all-reduced per 2-WL layer (dist.ShardedTwoWLScorer); this rank's logits == its slice of the one-GPU forward
    from lpgnn_amd import twowl
    g2 = torch.Generator().manual_seed(5)
    n2 = 3000
    und = torch.randint(0, n2, (2, 20_000), generator=g2)
    pe = twowl.double(und.to(dev))
    pr = twowl.double(torch.randint(0, n2, (2, 1001), generator=g2).to(dev))
    xdeg = twowl.degree(pe, n2)
    torch.manual_seed(3)
    mod = twowl.LocalWLNet(int(xdeg.max()), channels_1wl=32, channels_2wl=16, depth2=2).to(dev).eval()
    tg_all = twowl.tuple_graph(n2, pe, pr)
    with torch.no_grad():
        full_logits = mod.score_tuple_graph(xdeg, pe, tg_all, pe.shape[1] + torch.arange(pr.shape[1], device=dev))
    sc2 = ldist.ShardedTwoWLScorer(mod, n2, pe, pr)
    mine_logits = sc2(xdeg)
    out["twowl_sharded"] = (float((mine_logits - full_logits[sc2.lo:sc2.hi]).abs().max()), float(full_logits.abs().max()),
                            tuple(mine_logits.shape), sc2.hi - sc2.lo)

    def score_fn(p):                                      # stands in for the single-GPU SEAL scorer
        return (p[:, 0] * 1000 + p[:, 1]).to(torch.float32)

    # the library's own RCCL entry points (lpg_comm_init / lpg_allreduce_sum / lpg_reduce_scatter_sum), id over the store
    comm = ldist.RcclComm(world, rank)
    v = torch.arange(4096, dtype=torch.float32, device=dev) * (rank + 1)
    want = torch.arange(4096, dtype=torch.float32, device=dev) * (world * (world + 1) / 2)
    out["c_allreduce"] = bool(torch.equal(comm.all_reduce_(v.clone()), want))
    out["c_allreduce_bf16"] = bool(torch.equal(comm.all_reduce_(torch.ones(257, device=dev).bfloat16()),
                                               torch.full((257,), float(world), device=dev).bfloat16()))
    blk = comm.reduce_scatter(v.view(world, -1).contiguous().view(world * (4096 // world), 1))
    out["c_reduce_scatter"] = bool(torch.equal(blk.view(-1), want.view(world, -1)[rank]))
    torch.cuda.synchronize()
    comm.close()
    pairs = torch.arange(2 * 1001, dtype=torch.int64, device=dev).view(1001, 2)
    sc = ldist.ShardedLinkScorer(score_fn, always_collective=True)             # the all_gather_into_tensor branch
    out["scores"] = bool(torch.equal(sc(pairs, gather=True), score_fn(pairs)))
    torch.cuda.synchronize()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _run(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in range(world):
        out = res[rank]
        assert out["backend"] == "nccl" and out["scores"]
        assert out["c_allreduce"] and out["c_allreduce_bf16"] and out["c_reduce_scatter"]
        for collective in ("reduce_scatter", "all_reduce", "sparse"):
            err, scale, used, packed, _ = out[collective]
            assert used == collective and packed == (collective == "sparse")
            assert err <= 1e-5 * (scale + 1.0), (rank, collective, err)
        err_t, scale_t, shape_t, npairs = out["twowl_sharded"]
        assert shape_t == (npairs, 1) and err_t <= 2e-5 * scale_t, (rank, err_t, scale_t)
        ok_w, ok_b, prof_ok, bytes_w, bytes_f = out["bf16_wire"]
        assert ok_w and ok_b and prof_ok and bytes_w * 2 == bytes_f and (bytes_f > 0) == (world > 1)


def test_rccl_collectives_world1():
    """world size 1 over RCCL: every collective call of dist.py executes on the GPU box (dist.py: reduce_scatter_tensor,
    all_reduce, all_to_all_single, all_gather_into_tensor) and reproduces the unpartitioned SpMM."""
    _run(1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node)")
def test_rccl_collectives_world2():
    """two real ranks over xGMI: source-edge partition + each exchange form against the single-GPU SpMM."""
    _run(2)


def test_bench_launches_its_own_ranks():
    """plain `python bench.py --gpus 2` (no launcher around it) starts two ranks as child processes and prints ONE
    JSON line; rehearsed over gloo with both ranks on this box's one GPU (LPG_BENCH_BACKEND=gloo), small graph."""
    env = dict(os.environ, LPG_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--nodes", "200000", "--edges", "2000000", "--no-twowl", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["collective"] in ("sparse", "reduce_scatter")
