"""Multi-GPU aggregation: the north star's source-edge partition with an RCCL reduction.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).  Rank g owns the
source-node range [g*V/G, (g+1)*V/G): the rows of X in that range and every edge that leaves
them.  It computes a partial Y_g[V, d] over its edges; Y = sum_g Y_g.

Which collective.  all-reduce = reduce-scatter + all-gather.  The consumer of Y in a source-
partitioned multi-layer GNN is the next layer on the SAME partition, which reads only the rows of
its own range -- exactly the shard a reduce-scatter leaves on each rank.  So the default is
``collective="reduce_scatter"`` (rank g ends up with Y[lo_g:hi_g], fully reduced; (G-1)/G * V*d*4
bytes per rank instead of twice that); ``collective="all_reduce"`` is the literal north-star form
and returns the full Y on every rank.

Overlap.  xGMI is point-to-point, and even the reduce-scatter of a [V, d] fp32 partial moves more
bytes than the local SpMM takes time for on few GPUs.  The destination rows
are therefore cut into S stages; stage s holds sub-block s of EVERY rank's shard, laid out
[G, rows, d] so that one reduce_scatter_tensor serves it, and the collective of stage s is issued
asynchronously while the SpMM of stage s+1 runs.

Sparse exchange.  A rank's partial has a row for EVERY destination, but only the rows that receive an edge from
its sources are non-zero (the bench's RMAT graph: 58 % of them).  Which rows those are is a property of the graph,
known when it is built: ``collective="sparse"`` packs a stage's non-empty rows (one gather), moves them with ONE
variable-split all-to-all, and the owner adds each peer's rows into its shard in rank order (unique row ids per
peer: no atomics race, fixed order).  Same result as the reduce-scatter, 0.58 x the bytes on the links -- which is
what bounds the step at 2 and 4 GPUs.  ``collective="auto"`` takes it when fewer than 3/4 of the rows are non-empty.

The reference has no distributed code at all; this file has no reference counterpart.
"""
from typing import Callable, List, Optional, Tuple

import torch


def source_range(n_nodes: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous, balanced source-node ranges; the first ranks get the remainder."""
    base, rem = divmod(n_nodes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def row_blocks(n_rows: int, n_blocks: int) -> List[Tuple[int, int]]:
    n_blocks = max(1, min(n_blocks, n_rows)) if n_rows > 0 else 1
    return [source_range(n_rows, n_blocks, k) for k in range(n_blocks)]


def partition_edges_by_source(edge_index: torch.Tensor, n_nodes: int, world: int, rank: int) -> torch.Tensor:
    """the columns of edge_index [2,E] whose SOURCE (row 0) lies in this rank's range, order kept."""
    lo, hi = source_range(n_nodes, world, rank)
    m = (edge_index[0] >= lo) & (edge_index[0] < hi)
    return edge_index[:, m]


def shard_layout(n_nodes: int, world: int, n_stages: int) -> Tuple[int, int]:
    """(rows per rank shard, rows per stage sub-block): equal, padded shards so that one
    reduce_scatter_tensor per stage works.  Shard r covers destination rows
    [r*shard, (r+1)*shard) of the padded row space; rows >= n_nodes are padding (no edges)."""
    shard = -(-n_nodes // world)
    n_stages = max(1, min(n_stages, shard)) if shard > 0 else 1
    sub = -(-shard // n_stages)
    return sub * n_stages, sub


class SourcePartitionedSpMM:
    """Y = A @ X with A's edges partitioned by source across ranks.

    edge_index_local: [2, E_local] int64 with GLOBAL ids; every source must lie in `src_range`.
    forward(x_local [hi-lo, d]):
        collective="reduce_scatter" -> this rank's shard  Y[rank*shard : (rank+1)*shard]  (padded)
        collective="sparse"         -> the same shard, only the non-empty rows of the partial on the wire
        collective="auto"           -> "sparse" when fewer than 3/4 of the partial's rows are non-empty
        collective="all_reduce"     -> the full Y [V, d] on every rank
    `block_spmm` (CPU tests): callable (stage, dst_rows LongTensor, x_local) -> partial rows for
    those destination rows, replacing the HIP path.
    """

    def __init__(self, edge_index_local, n_nodes, src_range, n_blocks=4, edge_weight=None, group=None,
                 block_spmm: Optional[Callable] = None, collective="reduce_scatter", world=None, rank=None,
                 wire_dtype=None):
        """wire_dtype (sparse exchange only): torch.bfloat16 sends the partial rows in bf16 whatever x's dtype is (half
        the bytes on the links -- what bounds the step on few GPUs); the owner adds the peers' rows into an fp32 shard
        (lpg_rows_add_bf16), so the sum is rounded once at the end.  None: the rows travel in x's dtype."""
        import torch.distributed as dist
        self.n_nodes, self.src_range, self.group = int(n_nodes), src_range, group
        self.collective = collective
        self.wire_dtype = wire_dtype
        self.profile = False               # True: forward() records events around the stage SpMMs and the exchange waits
        self.last_profile = None
        ready = dist.is_available() and dist.is_initialized()
        self.world = world if world is not None else (dist.get_world_size(group) if ready else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if ready else 0)
        self.shard, self.sub = shard_layout(self.n_nodes, self.world, n_blocks)
        self.n_stages = self.shard // self.sub
        self.block_spmm = block_spmm
        self.graphs = []
        lo, hi = src_range
        dev = edge_index_local.device
        # stage s covers, for every rank r, padded rows [r*shard + s*sub, r*shard + (s+1)*sub)
        self.stage_rows = []
        for s in range(self.n_stages):
            base = torch.arange(self.world, device=dev).view(-1, 1) * self.shard + s * self.sub
            self.stage_rows.append((base + torch.arange(self.sub, device=dev).view(1, -1)).reshape(-1))
        src, dst = edge_index_local[0] - lo, edge_index_local[1]
        if edge_index_local.numel() and (int(src.min()) < 0 or int(src.max()) >= hi - lo):
            raise ValueError("SourcePartitionedSpMM: an edge's source is outside this rank's range")
        owner = torch.div(dst, self.shard, rounding_mode="floor")
        within = dst - owner * self.shard
        stage = torch.div(within, self.sub, rounding_mode="floor")
        local = owner * self.sub + (within - stage * self.sub)          # row inside the stage buffer
        if block_spmm is None:
            from .ops import CSRGraph
            for s in range(self.n_stages):
                m = stage == s
                ei = torch.stack([src[m], local[m]])
                w = None if edge_weight is None else edge_weight[m]
                self.graphs.append(CSRGraph.from_edge_index(ei, self.world * self.sub, hi - lo, edge_weight=w))
        self._plan_sparse(stage, local, dev, ready)
        self.graphs_packed = []
        if block_spmm is None and self.collective == "sparse":
            # the sparse exchange's SpMM writes the PACKED send buffer directly: destination rows renumbered to their
            # position among the stage's non-empty rows (owner-major), so nothing is gathered and no empty row exists
            from .ops import CSRGraph
            for s in range(self.n_stages):
                m = stage == s
                remap = torch.cumsum(self.row_mask[s].to(torch.int64), 0) - 1
                ei = torch.stack([src[m], remap[local[m]]])
                w = None if edge_weight is None else edge_weight[m]
                self.graphs_packed.append(CSRGraph.from_edge_index(ei, int(self.send_idx[s].numel()), hi - lo, edge_weight=w))
        self._stage_buf = {}

    def _plan_sparse(self, stage, local, dev, ready):
        """which rows of every stage buffer are non-empty on every rank (one all_gather of a byte mask at build time),
        and from that the gather list, the split sizes and the receive-side row ids of the sparse exchange."""
        import torch.distributed as dist
        rows = self.world * self.sub
        mask = torch.zeros((self.n_stages, rows), dtype=torch.uint8, device=dev)
        if stage.numel():
            mask.view(-1)[stage * rows + local] = 1
        if ready and self.world > 1:
            every = [torch.empty_like(mask) for _ in range(self.world)]
            dist.all_gather(every, mask, group=self.group)
            every = torch.stack(every)                                    # [world, S, world * sub]
        elif self.world > 1:
            # ranks emulated without a process group (layout tests): the peers' masks are unknown, no sparse plan
            self.nonempty_fraction = float(mask.float().mean())
            if self.collective == "sparse":
                raise ValueError("SourcePartitionedSpMM: collective='sparse' needs an initialised process group")
            if self.collective == "auto":
                self.collective = "reduce_scatter"
            return
        else:
            every = mask[None]
        every = every.view(self.world, self.n_stages, self.world, self.sub)
        self.nonempty_fraction = float(every.float().mean())
        use = self.collective == "sparse" or (self.collective == "auto" and self.nonempty_fraction <= 0.75)
        if self.collective == "auto":
            self.collective = "sparse" if use else "reduce_scatter"
        if not use:
            return
        mine = every[self.rank]                                           # [S, world(owner), sub]
        self.row_mask = mine.reshape(self.n_stages, -1)                   # own non-empty rows per stage buffer
        self.send_idx = [torch.nonzero(mine[s].reshape(-1)).flatten() for s in range(self.n_stages)]
        self.in_splits = [mine[s].sum(1, dtype=torch.int64).tolist() for s in range(self.n_stages)]
        to_me = every[:, :, self.rank]                                    # [world(peer), S, sub]
        self.recv_idx = [[torch.nonzero(to_me[p, s]).flatten() for p in range(self.world)] for s in range(self.n_stages)]
        self.out_splits = [[int(ix.numel()) for ix in self.recv_idx[s]] for s in range(self.n_stages)]
        self.max_split = every.sum(-1, dtype=torch.int64).amax(dim=(0, 2)).tolist()    # per stage, the same on every rank

    def nnz(self):
        return sum(g.nnz() for g in self.graphs)

    def _buffers(self, d, dtype, device):
        key = (d, dtype)
        if key not in self._stage_buf:
            self._stage_buf[key] = [torch.empty((self.world * self.sub, d), dtype=dtype, device=device)
                                    for _ in range(min(self.n_stages, 3))]      # ring of staging buffers
        return self._stage_buf[key]

    def forward(self, x_local, out=None, collective=None):
        """collective: override of the constructor's choice for this call (bench.py checks the sparse exchange against
        the reduce-scatter once, on the same runner)."""
        import torch.distributed as dist
        d = x_local.shape[1]
        dev, dt = x_local.device, x_local.dtype
        use_dist = dist.is_available() and dist.is_initialized()
        backend = dist.get_backend(self.group) if use_dist else None
        asyn = use_dist and backend != "gloo"
        bufs = self._buffers(d, dt, dev)
        mode = collective or self.collective
        if mode == "sparse" and not hasattr(self, "send_idx"):
            mode = "reduce_scatter"           # no plan (dense graph): the dense collective
        if mode == "sparse":
            return self._forward_sparse(x_local, out, bufs, use_dist, asyn)
        rs = mode == "reduce_scatter"
        if rs:
            y = out if out is not None else torch.empty((self.shard, d), dtype=dt, device=dev)
        else:       # padded row space [world, n_stages, sub, d]; the caller gets the first n_nodes rows
            full = torch.empty((self.world * self.shard, d), dtype=dt, device=dev)
            full4 = full.view(self.world, self.n_stages, self.sub, d)
        pending = [None] * len(bufs)          # (work, stage) whose staging buffer is still in use

        def retire(slot):
            if pending[slot] is None:
                return
            work, st = pending[slot]
            if work is not None:
                work.wait()
            if not rs:                        # all-reduced stage rows go back to their global positions
                full4[:, st].copy_(bufs[slot].view(self.world, self.sub, d))
            pending[slot] = None

        for s in range(self.n_stages):
            slot = s % len(bufs)
            retire(slot)
            buf = bufs[slot]
            if self.block_spmm is not None:
                buf.copy_(self.block_spmm(s, self.stage_rows[s], x_local))
            else:
                self.graphs[s].spmm(x_local, out=buf)
            work = None
            if rs:
                dst = y[s * self.sub:(s + 1) * self.sub]
                if asyn:                      # overlaps the SpMM of stage s+1
                    work = dist.reduce_scatter_tensor(dst, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                else:                         # no process group, or gloo (no reduce_scatter): CPU test path
                    if use_dist:
                        dist.all_reduce(buf, group=self.group)
                    dst.copy_(buf[self.rank * self.sub:(self.rank + 1) * self.sub])
            elif use_dist:
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=asyn)
                work = work if asyn else None
            pending[slot] = (work, s)
        for slot in range(len(bufs)):
            retire(slot)
        if rs:
            return y
        if out is not None:
            out.copy_(full[: self.n_nodes])
            return out
        return full[: self.n_nodes]


    def preallocate(self, x_local):
        """every buffer a forward of either exchange form allocates, up front: a rank that cannot get them fails HERE,
        before any peer has entered a collective (bench.py agrees on the outcome before the first exchange)."""
        d, dt, dev = x_local.shape[1], x_local.dtype, x_local.device
        self._buffers(d, dt, dev)
        held = [torch.empty((self.shard, d), dtype=dt, device=dev)]
        if hasattr(self, "out_splits"):
            for s in range(min(self.n_stages, 3)):
                held.append(torch.empty((sum(self.out_splits[s]), d), dtype=dt, device=dev))
                held.append(torch.empty((int(self.send_idx[s].numel()), d), dtype=dt, device=dev))
        del held                                          # back to torch's caching allocator: the forwards reuse them

    def compute_only(self, x_local):
        """the stage SpMMs of one step without any exchange (bench.py reports it next to the whole step)."""
        bufs = self._buffers(x_local.shape[1], x_local.dtype, x_local.device)
        for s in range(self.n_stages):
            if self.collective == "sparse" and self.graphs_packed:
                self.graphs_packed[s].spmm(x_local)
            elif self.block_spmm is None:
                self.graphs[s].spmm(x_local, out=bufs[s % len(bufs)])

    # ---- sparse exchange ------------------------------------------------------------------------------------
    def _gather_rows(self, buf, idx):
        if buf.is_cuda:
            from . import ops
            return ops.embedding(buf, idx)                            # lpg_embedding_f32: out[i, :] = buf[idx[i], :]
        return buf[idx]

    def _add_rows(self, y_stage, idx, rows):
        if not idx.numel():
            return
        if y_stage.is_cuda:
            from . import ops
            ops.rows_add_distinct(rows, idx, y_stage)                 # unique ids per peer: plain read-modify-writes (fp32 or bf16 rows)
        else:
            y_stage.index_add_(0, idx, rows.to(y_stage.dtype))

    def _exchange(self, send, s, use_dist, asyn):
        """all-to-all of the packed rows of stage s; returns (recv [sum(out_splits), d], work or None)."""
        import torch.distributed as dist
        d = send.shape[1]
        recv = torch.empty((sum(self.out_splits[s]), d), dtype=send.dtype, device=send.device)
        if not use_dist:
            recv.copy_(send)
            return recv, None
        if asyn:
            work = dist.all_to_all_single(recv, send, self.out_splits[s], self.in_splits[s], group=self.group, async_op=True)
            return recv, work
        # gloo (CPU tests) has no all_to_all: every rank publishes its padded per-owner pieces, takes its own
        width = max(int(self.max_split[s]), 1)                  # a width every rank agrees on
        pad = send.new_zeros((self.world, width, d))
        o = 0
        for r, c in enumerate(self.in_splits[s]):
            pad[r, :c] = send[o:o + c]
            o += c
        every = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(every, pad, group=self.group)
        o = 0
        for p, c in enumerate(self.out_splits[s]):
            recv[o:o + c] = every[p][self.rank, :c]
            o += c
        return recv, None

    def bytes_on_links_per_step(self, d, dtype, collective=None):
        """bytes this rank SENDS to other ranks in one forward (the quantity xGMI bounds): the packed rows owned by
        peers for the sparse exchange; (G - 1) / G of every stage buffer for a reduce-scatter, twice that for an
        all-reduce (reduce-scatter + all-gather), whatever algorithm RCCL picks."""
        mode = collective or self.collective
        elt = torch.empty((), dtype=dtype).element_size()
        if mode == "sparse" and hasattr(self, "in_splits"):
            wire = torch.empty((), dtype=self.wire_dtype or dtype).element_size()
            rows = sum(c for s in range(self.n_stages) for p, c in enumerate(self.in_splits[s]) if p != self.rank)
            return int(rows * d * wire)
        per = self.n_stages * self.world * self.sub * d * elt * (self.world - 1) // self.world
        return int(per * (2 if mode == "all_reduce" else 1))

    def _forward_sparse(self, x_local, out, bufs, use_dist, asyn):
        d = x_local.shape[1]
        dev, dt = x_local.device, x_local.dtype
        wire = self.wire_dtype or dt
        # the shard the peers' rows are added into is fp32 whenever rows travel or are stored in bf16
        low = (torch.bfloat16, torch.float16)
        acc_dt = torch.float32 if (wire in low or dt in low) else dt
        y_out = out if out is not None else torch.empty((self.shard, d), dtype=dt, device=dev)
        y = y_out if acc_dt == dt else torch.empty((self.shard, d), dtype=acc_dt, device=dev)
        y.zero_()
        pending = [None] * len(bufs)          # (work, stage, recv, send): recv lands in y once the work is done
        prof = [] if (self.profile and x_local.is_cuda) else None

        def retire(slot):
            if pending[slot] is None:
                return
            work, st, recv, _send = pending[slot]
            if prof is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                prof.append(("wait0", st, e))
            if work is not None:
                work.wait()
            if prof is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                prof.append(("wait1", st, e))
            ys = y[st * self.sub:(st + 1) * self.sub]
            o = 0
            for p in range(self.world):                               # fixed order: rank 0's rows first
                c = self.out_splits[st][p]
                self._add_rows(ys, self.recv_idx[st][p], recv[o:o + c])
                o += c
            pending[slot] = None

        for s in range(self.n_stages):
            slot = s % len(bufs)
            retire(slot)
            buf = bufs[slot]
            if prof is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                prof.append(("spmm0", s, e))
            if self.graphs_packed:
                send = self.graphs_packed[s].spmm(x_local)            # straight into the packed rows (owner-major)
            elif self.block_spmm is not None:
                buf.copy_(self.block_spmm(s, self.stage_rows[s], x_local))
                send = self._gather_rows(buf, self.send_idx[s])       # the non-empty rows, grouped by owner
            else:
                self.graphs[s].spmm(x_local, out=buf, skip_empty_rows=True)    # empty rows are never gathered
                send = self._gather_rows(buf, self.send_idx[s])
            if send.dtype != wire:
                send = send.to(wire)                                  # fp32 partials rounded once for the wire
            if prof is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                prof.append(("spmm1", s, e))
            recv, work = self._exchange(send, s, use_dist, asyn)      # overlaps the SpMM of stage s+1
            pending[slot] = (work, s, recv, send)
        for slot in range(len(bufs)):
            retire(slot)
        if y is not y_out:
            y_out.copy_(y)                                            # one rounding of the finished sums
        if prof is not None:
            torch.cuda.synchronize()
            ev = {(k, st): e for k, st, e in prof}
            self.last_profile = {
                "spmm_ms": [ev["spmm0", s].elapsed_time(ev["spmm1", s]) for s in range(self.n_stages)],
                "exchange_wait_ms": [ev["wait0", s].elapsed_time(ev["wait1", s]) for s in range(self.n_stages)]}
        return y_out


class ShardedLinkScorer:
    """Scored-edge parallelism: candidate pairs are independent units, so
    rank g scores the contiguous slice ``source_range(B, G, g)`` of the candidate list against a
    REPLICA of the (small) observed graph; there is no exchange on the data path.  The only
    communication is the optional gather of the [B] scores at the end (``gather=True``: one
    all_gather of equal, padded slices -- 4 B per scored edge).

    ``score_fn(pairs_slice [b, 2]) -> scores [b]`` is the single-GPU scorer, e.g. a closure over
    ``lpgnn_amd.seal.extract_subgraphs / gnn_inputs / graph_conv_stack / head``.
    """

    def __init__(self, score_fn: Callable, group=None, world=None, rank=None, always_collective=False):
        import torch.distributed as dist
        ready = dist.is_available() and dist.is_initialized()
        self.score_fn, self.group = score_fn, group
        self.always_collective = bool(always_collective) and ready    # world-size-1 rehearsal of the gather (tests)
        self.world = world if world is not None else (dist.get_world_size(group) if ready else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if ready else 0)

    def my_slice(self, n_pairs: int) -> Tuple[int, int]:
        return source_range(n_pairs, self.world, self.rank)

    def __call__(self, pairs: torch.Tensor, gather: bool = True) -> torch.Tensor:
        import torch.distributed as dist
        B = pairs.shape[0]
        lo, hi = self.my_slice(B)
        mine = self.score_fn(pairs[lo:hi])
        if mine.shape[0] != hi - lo:
            raise ValueError("ShardedLinkScorer: score_fn must return one score per pair")
        if not gather or (self.world == 1 and not self.always_collective):
            return mine
        width = -(-B // self.world)                       # slices differ by at most one pair: pad to the widest
        send = mine.new_zeros((width,) + tuple(mine.shape[1:]))
        send[: hi - lo] = mine
        recv = mine.new_empty((self.world * width,) + tuple(mine.shape[1:]))
        if dist.get_backend(self.group) == "gloo":        # CPU tests: gloo has no all_gather_into_tensor
            dist.all_gather(list(recv.view(self.world, width, *mine.shape[1:]).unbind(0)), send, group=self.group)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        parts = []
        for r in range(self.world):
            a, b = source_range(B, self.world, r)
            parts.append(recv[r * width: r * width + (b - a)])
        return torch.cat(parts)


class RcclComm:
    """RCCL through the library's own C entry points (include/lpg_hip.h: lpg_comm_*, lpg_allreduce_sum,
    lpg_reduce_scatter_sum) -- the C boundary a non-PyTorch host would bind.  The 128-byte unique id travels through
    `exchange(id_bytes_or_None) -> id_bytes` (rank 0 passes its id, every rank gets rank 0's back): by default a
    torch.distributed broadcast_object_list on the already-initialised process group (any backend, e.g. gloo)."""

    def __init__(self, world, rank, exchange=None):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        lib = _lib.load()
        if not lib.lpg_comm_available():
            raise _lib.LpgError("RCCL is not available in this process (dlopen(librccl.so.1) failed)")
        idb = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(lib.lpg_comm_unique_id(ctypes.addressof(idb)), "comm_unique_id")
        if exchange is None:
            import torch.distributed as dist

            def exchange(b):
                box = [b]
                if world > 1:
                    dist.broadcast_object_list(box, src=0)
                return box[0]
        got = exchange(idb.raw if rank == 0 else None)
        idb = ctypes.create_string_buffer(got, 128)
        self._comm = ctypes.c_void_p()
        _lib.check(lib.lpg_comm_init(ctypes.addressof(self._comm), int(world), int(rank), ctypes.addressof(idb)), "comm_init")
        self.world, self.rank = int(world), int(rank)

    def _args(self, t):
        if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_cuda or not t.is_contiguous():
            raise TypeError("RcclComm: contiguous float32 / bfloat16 device tensors")
        return int(t.dtype == torch.bfloat16), torch.cuda.current_stream(t.device).cuda_stream

    def all_reduce_(self, t):
        """t <- sum over ranks (in place), enqueued on the current stream"""
        bf, st = self._args(t)
        self._lib.check(self._lib.load().lpg_allreduce_sum(self._comm, t.data_ptr(), t.data_ptr(), t.numel(), bf, st), "allreduce_sum")
        return t

    def reduce_scatter(self, send, out=None):
        """send [world * k, ...] -> this rank's block of the sum, [k, ...]"""
        bf, st = self._args(send)
        if send.shape[0] % self.world:
            raise ValueError(f"reduce_scatter: {send.shape[0]} rows do not split over {self.world} ranks")
        k = send.shape[0] // self.world
        if out is None:
            out = torch.empty((k,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        elif (out.numel() * self.world != send.numel() or out.dtype != send.dtype or out.device != send.device
              or not out.is_contiguous()):
            raise ValueError("reduce_scatter: `out` must be a contiguous tensor of this rank's block "
                             f"({k} x {tuple(send.shape[1:])}, {send.dtype}) on the same device")
        self._lib.check(self._lib.load().lpg_reduce_scatter_sum(self._comm, send.data_ptr(), out.data_ptr(), out.numel(), bf, st),
                        "reduce_scatter_sum")
        return out

    def close(self):
        if self._comm:
            self._lib.check(self._lib.load().lpg_comm_destroy(self._comm), "comm_destroy")
            self._comm = self._ct.c_void_p()


class ShardedTwoWLScorer:
    """TwoWL scoring with the CANDIDATE tuples sharded across ranks (SURVEY.md section 8e (i) for the 2-WL model): rank g
    holds a replica of the graph, the model and the node-level tables, and the non-member rows of its own contiguous slice
    of the candidate pairs.  Unlike SEAL's subgraphs the candidates are not independent: GraphNorm's column statistics
    run over ALL tuples (TwoWL/model/model.py:38,77 -- Seq passes the whole [P, C] tensor), so every 2-WL layer has one
    real exchange step: the sums of y and y^2 over the non-member rows, a [2, 64] float64 block, are all-reduced
    (RCCL over xGMI; 1 KB per layer), the member rows' sums are identical on every rank.  Scores equal the single-GPU
    forward over the whole candidate list up to the order of those sums.

    pos_edge [2, E1], pred_edge [2, Ep] (both in double()'s layout: a candidate PAIR is two consecutive tuples)."""

    def __init__(self, mod, n_node, pos_edge, pred_edge, group=None, world=None, rank=None):
        import torch.distributed as dist
        from . import twowl
        ready = dist.is_available() and dist.is_initialized()
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if ready else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if ready else 0)
        self.collective = ready
        self.mod, self.pos_edge = mod, pos_edge
        E1, Ep = pos_edge.shape[1], pred_edge.shape[1]
        if Ep % 2:
            raise ValueError("ShardedTwoWLScorer: candidate tuples come in pairs (double()'s layout)")
        self.lo, self.hi = source_range(Ep // 2, self.world, self.rank)          # candidate PAIRS of this rank
        local = pred_edge[:, 2 * self.lo: 2 * self.hi].contiguous()
        self.tg = twowl.TupleGraph(n_node, pos_edge, local)
        self.n_rows = E1 + Ep                                                     # tuples of the whole candidate set
        self.idx = E1 + torch.arange(local.shape[1], device=pos_edge.device)
        self.bytes_per_layer = 2 * 64 * 8

    def _reduce(self, block):
        import torch.distributed as dist
        if self.collective:
            dist.all_reduce(block, op=dist.ReduceOp.SUM, group=self.group)

    @torch.no_grad()
    def __call__(self, x):
        """x: the degree vector (or node index input) of the model; returns this rank's [hi - lo, 1] logits"""
        return self.mod.score_tuple_graph(x, self.pos_edge, self.tg, self.idx, reduce_q=self._reduce, n_rows=self.n_rows)
