"""Drop-in for the ``torch_sparse`` call surface named by the north star.

The reference pins torch-sparse 0.6.17 (requirements.txt:19) but never imports it; the
signatures are kept [3P] so that a PyG-style caller can switch:
    spmm(index[2,nnz], value[nnz], m, n, matrix[n,k]) -> [m,k]     out[row] += value * matrix[col]
    SparseTensor(row=, rowptr=, col=, value=, sparse_sizes=, is_sorted=);  .csr();  matmul(A, x)
Both run the chunked CSR SpMM kernel (lpg_csr_spmm_*).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops as _ops


def spmm(index: Tensor, value: Optional[Tensor], m: int, n: int, matrix: Tensor) -> Tensor:
    """torch_sparse.spmm: sparse [m,n] (COO: index[0]=row, index[1]=col) @ dense [n,k]."""
    squeeze = matrix.dim() == 1
    if squeeze:
        matrix = matrix.unsqueeze(-1)
    # CSR rows are the OUTPUT rows: "destination" = index[0], "source" = index[1]
    g = _ops.CSRGraph.from_edge_index(torch.stack([index[1], index[0]]), m, n, edge_weight=value)
    out = g.spmm(matrix)
    return out.squeeze(-1) if squeeze else out


class SparseTensor:
    """Minimal CSR holder with the torch_sparse.SparseTensor constructor keywords."""

    def __init__(self, row: Optional[Tensor] = None, rowptr: Optional[Tensor] = None, col: Optional[Tensor] = None,
                 value: Optional[Tensor] = None, sparse_sizes: Optional[Tuple[int, int]] = None,
                 is_sorted: bool = False):
        if col is None:
            raise ValueError("SparseTensor: `col` is required")
        if sparse_sizes is None:
            m = int(rowptr.numel() - 1) if rowptr is not None else int(row.max()) + 1
            sparse_sizes = (m, int(col.max()) + 1)
        self._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        if rowptr is not None:
            self._g = _ops.CSRGraph.from_csr(rowptr, col, value, self._sizes[1])
        else:
            self._g = _ops.CSRGraph.from_edge_index(torch.stack([col, row]), self._sizes[0], self._sizes[1],
                                                    edge_weight=value)

    @classmethod
    def from_edge_index(cls, edge_index: Tensor, edge_attr: Optional[Tensor] = None,
                        sparse_sizes: Optional[Tuple[int, int]] = None):
        return cls(row=edge_index[0], col=edge_index[1], value=edge_attr, sparse_sizes=sparse_sizes)

    def sparse_sizes(self):
        return self._sizes

    def size(self, dim: int):
        return self._sizes[dim]

    def nnz(self):
        return self._g.nnz()

    def csr(self):
        n = self._g.nnz()
        val = None if self._g.val is None else self._g.val[:n]
        return self._g.rowptr, self._g.col[:n].to(torch.int64), val

    def matmul(self, other: Tensor, reduce: str = "sum") -> Tensor:
        return matmul(self, other, reduce)

    def __matmul__(self, other):
        return matmul(self, other)


def matmul(src: SparseTensor, other: Tensor, reduce: str = "sum") -> Tensor:
    if reduce not in ("sum", "add"):
        raise ValueError("lpgnn_amd.torch_sparse.matmul: only reduce='sum' is on the reference's path")
    return src._g.spmm(other)
