"""PyG-signature modules the reference's TwoWL model is built from, running on the HIP kernels.

    GCNConv(in_channels, out_channels, improved=False, cached=False, add_self_loops=True,
            normalize=True, bias=True).forward(x, edge_index, edge_weight=None)
    GraphNorm(in_channels, eps=1e-5).forward(x, batch=None)

[3P] torch-geometric 2.3.1 semantics, restated from the published definitions; reference call
sites TwoWL/model/model.py:36-41 (relu_conv), :54 (emb GraphNorm).  Parameter names match PyG
(`lin.weight`, `bias`; `weight`, `bias`, `mean_scale`) so reference state_dicts load unchanged.

Forward only (the north star is the scoring forward; the training loop is out of scope,
SURVEY.md section 2 row 3): calling with autograd enabled in training mode raises.
"""
import math
from collections import OrderedDict

import torch
from torch import nn

from . import ops as _ops


class _GraphCache:
    """Per-process cache of CSR conversions keyed by the edge_index tensor identity.

    PyG recomputes gcn_norm on every call (cached=False); the result only depends on
    edge_index / edge_weight, so the conversion is memoised on (data_ptr, _version, shape).
    Entries hold a reference to the key tensors, so their storage cannot be recycled while
    cached; in-place edits bump _version and miss.
    """

    def __init__(self, capacity=16):
        self.capacity = capacity
        self._d = OrderedDict()

    def get(self, edge_index, edge_weight, n, gcn, flip=0):
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), edge_index.device, n, gcn, flip,
               None if edge_weight is None else (edge_weight.data_ptr(), edge_weight._version))
        hit = self._d.get(key)
        if hit is not None:
            self._d.move_to_end(key)
            return hit[0]
        g = _ops.CSRGraph.from_edge_index(edge_index, n, n, edge_weight=edge_weight, gcn_self_loops=gcn, flip=flip)
        if gcn:
            g.gcn_normalize_()
        self._d[key] = (g, edge_index, edge_weight)
        if len(self._d) > self.capacity:
            self._d.popitem(last=False)
        return g

    def clear(self):
        self._d.clear()


graph_cache = _GraphCache()
SMALL_GRAPH_KERNELS = True      # GCNConv on graphs of <= lpg_small_max_rows() rows takes the small-graph kernels


def _forward_only(mod):
    if mod.training and torch.is_grad_enabled():
        raise NotImplementedError(
            "lpgnn_amd modules implement the scoring forward only (eval mode / torch.no_grad()); "
            "the reference's training loop is outside the hot path (SURVEY.md section 8)")


class _LinNoBias(nn.Module):
    """stand-in for torch_geometric.nn.dense.linear.Linear(in, out, bias=False): `weight` [out,in]."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        a = math.sqrt(6.0 / (in_channels + out_channels))      # glorot
        nn.init.uniform_(self.weight, -a, a)

    def forward(self, x):
        return _ops.linear(x, self.weight)


class GCNConv(nn.Module):
    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=True,
                 normalize=True, bias=True, **kwargs):
        super().__init__()
        if improved or not add_self_loops or not normalize:
            raise NotImplementedError("only the configuration the reference uses is built: "
                                      "improved=False, add_self_loops=True, normalize=True")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.cached = cached
        self.lin = _LinNoBias(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x, edge_index, edge_weight=None):
        """edge_index: [2,E] int64 tensor (PyG), or a prebuilt gcn-normalised CSRGraph."""
        _forward_only(self)
        if hasattr(edge_index, "orient") and hasattr(edge_index, "graph"):
            # a twowl.TupleOrientation (reverse() of a TupleGraph): the operator form of the 2-WL aggregation -- group
            # sums of the member rows + one broadcast, no tuple graph; the explicit edges when the layout conditions fail
            tg = edge_index.graph
            tp = tg.plan(operator_form=True)
            if tp.ok and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
                return _ops.tuple_gcn_operator(tp, edge_index.orient, x, self.lin.weight, self.bias)
            g = graph_cache.get(tg.materialize(), None, x.shape[0], True, flip=1 + edge_index.orient)
        elif isinstance(edge_index, _ops.CSRGraph):
            g = edge_index
        else:
            g = graph_cache.get(edge_index, edge_weight, x.shape[0], True)
        n = x.shape[0]
        if (SMALL_GRAPH_KERNELS and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 0 < n <= _ops.small_max_rows()
                and g.unit_gcn and g.n_rows == n and g.n_cols == n and self.in_channels <= 256):
            # the reference's own graphs (620 nodes / 4252 tuples) are launch-bound: two launches of the small-graph
            # kernels (transposed linear; aggregation + bias out of LDS) instead of linear + SpMM + fix-up, also when
            # the module is called through the reference's own Seq (INTEGRATION.md option A)
            hT = _ops.small_linear_t(x, self.lin.weight)
            out, = _ops.small_gcn(n, [(hT, g, self.bias, None, False)])
            return out.y
        h = self.lin(x)                                   # K5: H = X W^T on the fp32 matrix cores
        return g.spmm(h, bias=self.bias)                  # K3+K4: A_hat @ H + b, one fused kernel


class GraphNorm(nn.Module):
    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = nn.Parameter(torch.ones(in_channels))

    def forward(self, x, batch=None, *, relu=False, add_to=None):
        _forward_only(self)
        if batch is not None:
            raise NotImplementedError("the reference only calls GraphNorm with batch=None (single graph)")
        return _ops.graph_norm(x, self.weight, self.bias, self.mean_scale, self.eps, relu=relu, add_to=add_to)
