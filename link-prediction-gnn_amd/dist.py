"""Multi-GPU aggregation: the north star's source-edge partition with an RCCL all-reduce.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).  Rank g owns the
source-node range [g*V/G, (g+1)*V/G): the rows of X in that range and every edge that leaves
them.  It computes a partial Y_g[V, d] over its edges; Y = sum_g Y_g is an all-reduce.

xGMI is point-to-point and the all-reduce moves ~2x the [V, d] partial per rank, which is more
time than the local SpMM.  So the destination rows are cut into row
blocks and the all-reduce of block k is issued asynchronously while the SpMM of block k+1 runs:
the collective of every block but the last hides behind compute (or vice versa).

The reference has no distributed code at all; this file has no reference counterpart.
"""
from typing import Callable, List, Optional, Tuple

import torch


def source_range(n_nodes: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous, balanced source-node ranges; the last ranks get the remainder."""
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


class SourcePartitionedSpMM:
    """Y = A @ X with A's edges partitioned by source across ranks.

    edge_index_local: [2, E_local] int64 with GLOBAL ids; every source must lie in `src_range`.
    forward(x_local [hi-lo, d]) returns the full, all-reduced Y [V, d] on every rank.
    `block_spmm` (tests): callable (k, row0, row1, x_local) -> partial block, replacing the HIP path.
    """

    def __init__(self, edge_index_local, n_nodes, src_range, n_blocks=8, edge_weight=None, group=None,
                 block_spmm: Optional[Callable] = None):
        self.n_nodes, self.src_range, self.group = int(n_nodes), src_range, group
        lo, hi = src_range
        self.blocks = row_blocks(self.n_nodes, n_blocks)
        self.block_spmm = block_spmm
        self.graphs = []
        if block_spmm is None:
            from .ops import CSRGraph
            src, dst = edge_index_local[0] - lo, edge_index_local[1]
            if edge_index_local.numel() and (int(src.min()) < 0 or int(src.max()) >= hi - lo):
                raise ValueError("SourcePartitionedSpMM: an edge's source is outside this rank's range")
            for (r0, r1) in self.blocks:
                m = (dst >= r0) & (dst < r1)
                ei = torch.stack([src[m], dst[m] - r0])
                w = None if edge_weight is None else edge_weight[m]
                self.graphs.append(CSRGraph.from_edge_index(ei, r1 - r0, hi - lo, edge_weight=w))

    def nnz(self):
        return sum(g.nnz() for g in self.graphs)

    def forward(self, x_local, out=None):
        import torch.distributed as dist
        d = x_local.shape[1]
        y = out if out is not None else torch.empty((self.n_nodes, d), dtype=x_local.dtype, device=x_local.device)
        works = []
        use_dist = dist.is_available() and dist.is_initialized()
        for k, (r0, r1) in enumerate(self.blocks):
            blk = y[r0:r1]
            if self.block_spmm is not None:
                blk.copy_(self.block_spmm(k, r0, r1, x_local))
            else:
                self.graphs[k].spmm(x_local, out=blk)
            if use_dist:       # async: the collective of block k overlaps the SpMM of block k+1
                works.append(dist.all_reduce(blk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        return y
