"""Drop-in for the ``torch_scatter`` call surface the reference uses.

Reference call site: TwoWL/utils.py:5,10 -- ``from torch_scatter import scatter_add`` inside the
``@torch.jit.script`` function ``degree``.  Signature kept from torch-scatter 2.1.1 [3P]:
    scatter_add(src, index, dim=-1, out=None, dim_size=None) -> Tensor   (alias of scatter_sum)
``index`` is broadcast to ``src`` along every dim but ``dim``; with ``out=None`` the result is
zeros with size[dim] = dim_size or index.max()+1.

The function body is TorchScript-compatible and dispatches to registered torch ops
(``torch.ops.lpgnn.*``), so scripted callers such as ``degree`` keep working.  The ops run the
HIP kernels lpg_scatter_add_{i64,f32}; CPU tensors raise (no fallback).
"""
from typing import Optional

import torch
from torch import Tensor

from . import ops as _ops

_lib = torch.library.Library("lpgnn", "DEF")
_lib.define("scatter_add(Tensor src, Tensor index, int dim, int? dim_size) -> Tensor")
_lib.define("scatter_add_out(Tensor src, Tensor index, int dim, Tensor(a!) out) -> ()")


def _rows_view(src, index, dim):
    """Move `dim` first and flatten the rest: the kernels scatter whole rows."""
    dim = dim if dim >= 0 else src.dim() + dim
    if index.dim() == 1 and src.dim() >= 1 and (dim == 0 or src.dim() == 1):
        if index.numel() != src.shape[dim]:
            raise ValueError("scatter_add: index length does not match src along dim")
        cols = 1
        for v in src.shape[1:]:
            cols *= int(v)
        return dim, src.reshape(src.shape[0], cols) if src.dim() > 1 else src, index, None
    # general case: index broadcast to src (torch_scatter.utils.broadcast), element-wise scatter
    idx = index
    if idx.dim() == 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        idx = idx.view(shape)
    idx = idx.expand_as(src)
    return dim, src, idx, True


def _impl_into(src, index, dim, out, fresh=False):
    dim, s, idx, general = _rows_view(src, index, dim)
    if general is None:
        cols = 1
        for v in out.shape[1:]:
            cols *= int(v)
        o = out.view(out.shape[0], cols) if out.dim() > 1 else out
        if not o.is_contiguous():
            raise ValueError("scatter_add: `out` must be contiguous")
        if (s.dtype == torch.int64 and s.dim() == 1 and o.dim() == 1 and idx.numel() >= _ops.DEGREE_SORT_MIN and s.is_cuda
                and 16384 < o.shape[0] < (1 << 31) and idx.numel() < (1 << 31)):
            # the reference's degree() is exactly this call -- scatter_add(ones_like(ei[1]), ei[1], dim_size=N)
            # (TwoWL/utils.py:8-10): a big list of ones is a histogram and takes the atomic-free degree path.  Whether
            # src really is all ones is checked INSIDE that path (the kernels that read the index read src beside it);
            # round 2 spent a separate comparison pass + reduction + host sync on it (0.75 against degree()'s 0.46 ms
            # per 100 M entries)
            hist = o if fresh else torch.zeros_like(o)
            if _ops.degree_if_ones(idx, s, hist):
                if not fresh:
                    o += hist
                _ops.maybe_check(out.device, "scatter_add index")
                return
            if fresh:
                o.zero_()
        _ops.scatter_add_rows(s, idx, o, fresh)
    else:
        # element-wise: flatten to a 1-D scatter with linearised destination offsets
        src_m = s.movedim(dim, -1)
        idx_m = idx.movedim(dim, -1)
        out_m = out.movedim(dim, -1)
        if not out_m.is_contiguous():
            raise ValueError("scatter_add: unsupported `out` layout for dim != 0")
        lead = out_m.numel() // out_m.shape[-1] if out_m.shape[-1] else 0
        base = torch.arange(lead, device=src.device, dtype=torch.int64).view(out_m.shape[:-1] + (1,)) * out_m.shape[-1]
        # bounds: an index >= out.shape[dim] must be flagged, not alias a neighbour row
        bad = (idx_m < 0) | (idx_m >= out_m.shape[-1])
        lin = torch.where(bad, torch.full_like(idx_m, -1), idx_m + base)
        _ops.scatter_add_rows(src_m.reshape(-1), lin.reshape(-1), out_m.view(-1))
    _ops.maybe_check(out.device, "scatter_add index")


def _scatter_add(src, index, dim, dim_size):
    d = dim if dim >= 0 else src.dim() + dim
    size = list(src.shape)
    if dim_size is not None:
        size[d] = int(dim_size)
    elif index.numel() == 0:
        size[d] = 0
    else:
        size[d] = int(index.max()) + 1
    # big float32 row scatters write every output row themselves (ops.scatter_add_rows, grouped route): no zero fill
    grouped = d == 0 and index.dim() == 1 and src.dim() == 2 and src.is_cuda and \
        _ops.scatter_add_sorts(index.numel(), src.shape[1], src.dtype)
    out = (torch.empty if grouped else torch.zeros)(size, dtype=src.dtype, device=src.device)
    _impl_into(src, index, dim, out, fresh=True)
    return out


def _scatter_add_out(src, index, dim, out):
    _impl_into(src, index, dim, out)


_lib.impl("scatter_add", _scatter_add, "CompositeExplicitAutograd")
_lib.impl("scatter_add_out", _scatter_add_out, "CompositeExplicitAutograd")


def scatter_sum(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                dim_size: Optional[int] = None) -> Tensor:
    if out is None:
        return torch.ops.lpgnn.scatter_add(src, index, dim, dim_size)
    else:
        torch.ops.lpgnn.scatter_add_out(src, index, dim, out)
        return out


def scatter_add(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                dim_size: Optional[int] = None) -> Tensor:
    return scatter_sum(src, index, dim, out, dim_size)


def scatter(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
            dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    if reduce == "sum" or reduce == "add":
        return scatter_sum(src, index, dim, out, dim_size)
    raise ValueError("lpgnn_amd.torch_scatter: only reduce='sum'/'add' is on the reference's path")
