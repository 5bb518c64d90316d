"""Thin torch <-> C-ABI glue: tensors in, raw device pointers + sizes + hipStream_t out.

PyTorch is used only for device memory, streams and dtype bookkeeping; every computation below
is a kernel of liblpg_hip.so (include/lpg_hip.h).  All tensors must live on a ROCm device.
"""
import os
from collections import OrderedDict
from typing import Optional

import torch

from . import _lib

ACT = {None: 0, "none": 0, "relu": 1, "tanh": 2}
_CHECK = os.environ.get("LPG_CHECK_INDICES", "1") != "0"
_err_flags = {}


def _need_gpu(*ts):
    """every tensor of a call lives on ONE ROCm device, and that device is the current one: the launches below go to
    torch's current stream of the current device, so a tensor of another GPU would be dereferenced on the wrong one."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.LpgError(
                "lpgnn_amd ops run only on a ROCm device (got a CPU tensor); there is no CPU "
                "fallback in the product path -- the CPU oracle lives under oracle/ for tests only")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise _lib.LpgError(f"lpgnn_amd: tensors of one call live on different devices ({dev} and {t.device})")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise _lib.LpgError(f"lpgnn_amd: tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}; "
                            f"wrap the call in `with torch.cuda.device({dev.index}):` (one process per GPU is the "
                            "supported layout)")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def err_flag(device):
    f = _err_flags.get(device)
    if f is None:
        f = torch.zeros(1, dtype=torch.int32, device=device)
        _err_flags[device] = f
    return f


def raise_if_flagged(device, what="index"):
    """Reads the device error flag (synchronises).  Out-of-range indices were skipped by the
    kernels, never dereferenced; surface them the way ATen does on CPU."""
    f = err_flag(device)
    v = int(f.item())
    if v:
        f.zero_()
        raise IndexError(f"lpgnn_amd: {what} out of range (flag={v})")


def maybe_check(device, what="index"):
    # no host sync while a hipGraph is being captured: the flag is read after the replay instead
    if _CHECK and not torch.cuda.is_current_stream_capturing():
        raise_if_flagged(device, what)


def _c(t, dtype=None):
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------ scatter_add
DEGREE_SORT_MIN = 1 << 21           # entries from which int64 histograms take the atomic-free path of csrc/degree.hip
# entries from which float32 row scatters group by destination instead of using float atomics.  Measured
# (profiles/scatter_crossover.py -> profiles/r02/scatter_crossover.log, N = E / 10): with the grouping memoised -- a GNN
# calls with the same index tensor every layer and every step -- the grouped route is never slower from 2^13 rows and
# 1.4x (2^16) to 3.4x (2^21) faster at d = 128; a FIRST call on a new index tensor also pays the grouping (0.11-0.17 ms
# up to 2^19 rows, one host sync) and only wins outright from E * d ~ 2^27 elements.  2^16: from there the steady state
# wins clearly, every sum is bitwise reproducible (entry order), and the one-time grouping is ~2 call times.
SCATTER_SORT_MIN = 1 << 16


def scatter_add_sorts(E, d, dtype):
    return dtype == torch.float32 and E >= SCATTER_SORT_MIN and d >= 4 and E < (1 << 31)


class _ScatterPlans:
    """destination-grouped form of an index tensor (CSRGraph with column = entry id), memoised on the tensor's identity
    the way nn.graph_cache memoises edge_index; entries keep the index alive so its storage cannot be recycled."""

    def __init__(self, capacity=2):
        self.capacity, self._d = capacity, OrderedDict()

    def get(self, index, n_rows):
        key = (index.data_ptr(), index._version, index.numel(), index.device, n_rows)
        hit = self._d.get(key)
        if hit is not None:
            self._d.move_to_end(key)
            return hit[0]
        g = CSRGraph.from_scatter_index(index, n_rows)
        self._d[key] = (g, index)
        if len(self._d) > self.capacity:
            self._d.popitem(last=False)
        return g

    def clear(self):
        self._d.clear()


_scatter_plans = _ScatterPlans()


def scatter_add_rows(src: Optional[torch.Tensor], index: torch.Tensor, out: torch.Tensor, fresh: bool = False):
    """out[index[e], :] += src[e, :]  (src None => ones, int64 only).  out is [N] or [N, d].
    fresh: the caller allocated `out` for this call and does not need its old contents (it may be uninitialised
    when scatter_add_sorts() holds)."""
    lib = _lib.load()
    _need_gpu(index, out, src)
    index = _c(index, torch.int64)
    E = index.numel()
    N = out.shape[0]
    d = out.numel() // N if N > 0 else 1
    assert out.is_contiguous()
    flag = err_flag(out.device)
    if out.dtype == torch.int64:
        src = _c(src, torch.int64)
        _lib.check(lib.lpg_scatter_add_i64(_ptr(src), _ptr(index), E, d, _ptr(out), N, _ptr(flag), _stream()),
                   "scatter_add_i64")
    elif out.dtype == torch.float32:
        src = _c(src, torch.float32)
        if scatter_add_sorts(E, d, out.dtype) and src is not None and out.dim() == 2:
            # big lists: no atomics.  The entries are grouped by destination once per index tensor (a CSR whose columns
            # are entry ids, memoised like the GCN graphs) and every call is one gather-reduce over it -- sums in entry
            # order, reproducible, and 4-5x the rate of float atomics on [E, d] rows
            g = _scatter_plans.get(index, N)
            if fresh:
                g.spmm(src, out=out, use_val=False)
            else:
                out += g.spmm(src, use_val=False)
            return out
        _lib.check(lib.lpg_scatter_add_f32(_ptr(src), _ptr(index), E, d, _ptr(out), N, _ptr(flag), _stream()),
                   "scatter_add_f32")
    else:
        raise TypeError(f"scatter_add: unsupported dtype {out.dtype} (int64 and float32 are built)")
    return out


def rows_add_distinct(src: torch.Tensor, index: torch.Tensor, out: torch.Tensor):
    """out[index[e], :] += src[e, :] where the entries of `index` are DISTINCT (no atomics); out float32 [N, d], src
    float32 or bfloat16 (bf16 rows are widened as they are added: the accumulator stays fp32)."""
    lib = _lib.load()
    _need_gpu(index, out, src)
    index = _c(index, torch.int64)
    assert out.is_contiguous() and out.dtype == torch.float32
    N = out.shape[0]
    d = out.numel() // N if N > 0 else 1
    if src.dtype == torch.bfloat16:
        src = _c(src)
        _lib.check(lib.lpg_rows_add_bf16(_ptr(src), _ptr(index), index.numel(), d, _ptr(out), N, _ptr(err_flag(out.device)),
                                         _stream()), "rows_add_bf16")
        return out
    src = _c(src, torch.float32)
    _lib.check(lib.lpg_rows_add_f32(_ptr(src), _ptr(index), index.numel(), d, _ptr(out), N, _ptr(err_flag(out.device)),
                                    _stream()), "rows_add_f32")
    return out


def degree_if_ones(index: torch.Tensor, ones: torch.Tensor, out: torch.Tensor) -> bool:
    """out[index[e]] += 1 through the atomic-free histogram while CHECKING that `ones` (the source of the reference's
    scatter_add(ones_like(index), index) spelling) really is all ones -- inside the kernels that read the index, one
    4-byte flag read at the end.  Returns False (and `out` is garbage) when it is not; big int64 lists only."""
    lib = _lib.load()
    _need_gpu(index, ones, out)
    index, ones = _c(index, torch.int64), _c(ones, torch.int64)
    E, n = index.numel(), out.shape[0]
    nbytes = _lib.check(lib.lpg_degree_i64_bytes(E, n), "degree_bytes")
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=index.device)
    flag = torch.zeros(1, dtype=torch.int32, device=index.device)
    _lib.check(lib.lpg_degree_i64_ones(_ptr(index), _ptr(ones), E, _ptr(out), n, _ptr(ws), nbytes, _ptr(flag),
                                       _ptr(err_flag(index.device)), _stream()), "degree_i64_ones")
    return int(flag.item()) == 0


def degree(index: torch.Tensor, n: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int64 histogram of `index` over [0, n): the reference's degree() (TwoWL/utils.py:8-10) without
    materialising the ones vector; lists of >= 2 M entries are counted without atomics (one radix pass over the high
    bits, then 32 K-bin LDS histograms; csrc/degree.hip)."""
    lib = _lib.load()
    _need_gpu(index)
    index = _c(index, torch.int64)
    if out is None:
        out = torch.zeros(n, dtype=torch.int64, device=index.device)
    elif out.dtype != torch.int64 or out.shape != (n,) or not out.is_contiguous():
        raise ValueError("degree: `out` must be a contiguous int64 [n] tensor (counts are added to it)")
    E = index.numel()
    nbytes = _lib.check(lib.lpg_degree_i64_bytes(E, n), "degree_bytes")
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=index.device) if E >= DEGREE_SORT_MIN else None
    _lib.check(lib.lpg_degree_i64(_ptr(index), E, _ptr(out), n, _ptr(ws), nbytes if ws is not None else 0,
                                  _ptr(err_flag(index.device)), _stream()), "degree_i64")
    maybe_check(index.device, "degree index")
    return out


# ------------------------------------------------------------------------------ CSR graph
class CSRGraph:
    """Destination-major CSR of a graph plus the SpMM plan; built once, reused by every layer.

    rowptr int64 [N+1], col int32 [e_cap], val float32 [e_cap] or None; the number of stored
    edges is rowptr[N] on the device (<= e_cap), so no host sync is needed to use it.
    """

    def __init__(self, rowptr, col, val, n_rows, n_cols, e_cap, perm=None, dinv=None):
        self.rowptr, self.col, self.val = rowptr, col, val
        self.n_rows, self.n_cols, self.e_cap = int(n_rows), int(n_cols), int(e_cap)
        self.perm, self.dinv = perm, dinv
        self.device = rowptr.device
        self._plan = None
        self._ws = {}
        self.unit_gcn = False      # gcn-normalised from unit edge weights: val[e] == dinv[row] * dinv[col]
        self.n_empty = -1          # rows without edges, once known on the host (0 skips a launch)
        self.max_row = -1          # longest row, once known on the host (short rows: one-launch row-parallel SpMM)

    # -- construction ----------------------------------------------------------------------
    @staticmethod
    def from_edge_index(edge_index, n_rows, n_cols=None, edge_weight=None, gcn_self_loops=False,
                        keep_perm=False, flip=0):
        """edge_index [2,E] int64 (row 0 = source, row 1 = destination), PyG convention.
        flip: bit 0 / bit 1 read source / destination ids as id^1 (TwoWL reverse())."""
        lib = _lib.load()
        _need_gpu(edge_index, edge_weight)
        n_cols = n_rows if n_cols is None else n_cols
        dev = edge_index.device
        E = edge_index.shape[1]
        src = _c(edge_index[0], torch.int64)
        dst = _c(edge_index[1], torch.int64)
        w = _c(edge_weight, torch.float32)
        e_cap = E + (n_rows if gcn_self_loops else 0)
        rowptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
        col = torch.empty(max(e_cap, 1), dtype=torch.int32, device=dev)
        need_val = w is not None
        val = torch.empty(max(e_cap, 1), dtype=torch.float32, device=dev) if need_val else None
        perm = torch.empty(max(e_cap, 1), dtype=torch.int32, device=dev) if keep_perm else None
        nbytes = _lib.check(lib.lpg_coo_to_csr_bytes(E, n_rows, int(gcn_self_loops)), "coo_to_csr_bytes")
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        flag = err_flag(dev)
        _lib.check(lib.lpg_coo_to_csr(_ptr(src), _ptr(dst), _ptr(w), E, n_rows, n_cols, int(gcn_self_loops), int(flip),
                                      _ptr(rowptr), _ptr(col), _ptr(val), _ptr(perm), _ptr(ws), nbytes,
                                      _ptr(flag), _stream()), "coo_to_csr")
        g = CSRGraph(rowptr, col, val, n_rows, n_cols, e_cap, perm=perm)
        g._build_plan()
        g._finish_build("edge_index")           # one sync per graph build
        return g

    @staticmethod
    def from_scatter_index(index, n_rows):
        """the grouping scatter_add needs: row r lists, in order, the entries e with index[e] == r (column = e)."""
        lib = _lib.load()
        _need_gpu(index)
        dev = index.device
        dst = _c(index, torch.int64)
        E = dst.numel()
        rowptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
        col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        nbytes = _lib.check(lib.lpg_coo_to_csr_bytes(E, n_rows, 0), "coo_to_csr_bytes")
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        _lib.check(lib.lpg_coo_to_csr(None, _ptr(dst), None, E, n_rows, max(E, 1), 0, 0, _ptr(rowptr), _ptr(col), None,
                                      None, _ptr(ws), nbytes, _ptr(err_flag(dev)), _stream()), "coo_to_csr")
        g = CSRGraph(rowptr, col, None, n_rows, E, E)
        g._build_plan()
        g._finish_build("scatter_add index")
        return g

    @staticmethod
    def from_csr(rowptr, col, val, n_cols):
        _need_gpu(rowptr, col, val)
        rowptr = _c(rowptr, torch.int64)
        col = _c(col, torch.int32)
        val = _c(val, torch.float32)
        g = CSRGraph(rowptr, col, val, rowptr.numel() - 1, n_cols, col.numel())
        g._build_plan()
        g._finish_build("csr col/rowptr")
        return g

    def _build_plan(self):
        lib = _lib.load()
        nbytes = _lib.check(lib.lpg_csr_plan_bytes(self.n_rows, self.e_cap), "csr_plan_bytes")
        self._plan = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        _lib.check(lib.lpg_csr_plan_build(_ptr(self.rowptr), _ptr(self.col), self.n_rows, self.n_cols, self.e_cap,
                                          _ptr(self._plan), nbytes, _ptr(err_flag(self.device)), _stream()),
                   "csr_plan_build")

    def _finish_build(self, what):
        """the single host sync of a graph build: error flag + number of empty rows."""
        raise_if_flagged(self.device, what)
        head = self._plan[:8].view(torch.int32).tolist()
        self.n_empty, self.max_row = int(head[0]), int(head[1])

    def gcn_normalize_(self):
        """val <- D^-1/2 val D^-1/2 with D = destination weight sums (PyG gcn_norm)."""
        lib = _lib.load()
        unit = self.val is None
        if unit:
            self.val = torch.empty(max(self.e_cap, 1), dtype=torch.float32, device=self.device)
        self.dinv = torch.empty(max(self.n_rows, 1), dtype=torch.float32, device=self.device)
        _lib.check(lib.lpg_gcn_norm_csr(_ptr(self._plan), _ptr(self.rowptr), _ptr(self.col), _ptr(self.val),
                                        _ptr(self.dinv), self.n_rows, self.e_cap, int(unit), _stream()), "gcn_norm_csr")
        self.unit_gcn = unit
        return self

    def nnz(self):
        return int(self.rowptr[-1].item())

    def small_pack(self, row_blocks=1):
        """packed (col | row << 16) edge stream of the small-graph kernels, built once per (graph, row_blocks)."""
        pk = self._ws.get(("small_pack", row_blocks))
        if pk is None:
            lib = _lib.load()
            nbytes = _lib.check(lib.lpg_small_pack_bytes(self.e_cap, row_blocks), "small_pack_bytes")
            pk = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(lib.lpg_small_pack_build(_ptr(self._plan), _ptr(self.rowptr), _ptr(self.col), self.n_rows,
                                                self.e_cap, row_blocks, _ptr(pk), nbytes, _stream()), "small_pack_build")
            self._ws[("small_pack", row_blocks)] = pk
        return pk

    # -- the hot op --------------------------------------------------------------------------
    def _workspace(self, d):
        lib = _lib.load()
        nbytes = _lib.check(lib.lpg_csr_spmm_bytes(self.e_cap, d), "csr_spmm_bytes")
        ws = self._ws.get("spmm")
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws["spmm"] = ws
        return ws, nbytes

    def spmm(self, x, row_scale=None, bias=None, act=None, out=None, use_val=True, skip_empty_rows=False):
        """Y = act(row_scale * (A @ x) + bias);  x [n_cols, d] float32 or bfloat16.
        skip_empty_rows: rows without edges are left UNWRITTEN (the caller never reads them: the sparse multi-GPU
        exchange gathers only the non-empty rows of a partial)."""
        lib = _lib.load()
        _need_gpu(self.rowptr, x, row_scale, bias, out)
        if x.dim() != 2 or x.shape[0] != self.n_cols:
            raise ValueError(f"spmm: x must be [{self.n_cols}, d], got {tuple(x.shape)}")
        if x.stride(1) != 1:
            x = x.contiguous()
        d = x.shape[1]
        if out is None:
            out = torch.empty((self.n_rows, d), dtype=x.dtype, device=x.device)
        elif out.shape != (self.n_rows, d) or out.dtype != x.dtype or out.stride(1) != 1:
            raise ValueError("spmm: bad `out`")
        row_scale = _c(row_scale, torch.float32)
        bias = _c(bias, torch.float32)
        val = self.val if use_val else None
        ws, nbytes = self._workspace(d)
        args = (_ptr(self._plan), _ptr(self.rowptr), _ptr(self.col), _ptr(val), _ptr(x), x.stride(0), _ptr(out),
                out.stride(0), self.n_rows, self.n_cols, self.e_cap, d, _ptr(row_scale), _ptr(bias), ACT[act],
                0 if skip_empty_rows else self.n_empty,
                self.max_row, _ptr(ws), nbytes, _stream())
        if x.dtype == torch.float32:
            _lib.check(lib.lpg_csr_spmm_f32(*args), "csr_spmm_f32")
        elif x.dtype == torch.bfloat16:
            _lib.check(lib.lpg_csr_spmm_bf16(*args), "csr_spmm_bf16")
        else:
            raise TypeError(f"spmm: unsupported dtype {x.dtype}")
        return out


# ------------------------------------------------------------------------------ dense / norm
def linear(x, weight, bias=None, w_is_km=False, out=None, act=None, n=None, ldx=None):
    """x [n,k] @ weight^T (+bias, act); weight [m,k] (nn.Linear) or [k,m] when w_is_km.
    n / ldx: treat x's storage as n rows of k values every ldx elements (overlapping windows, e.g.
    SEAL's width-5 conv1d as a matmul over a [B*k,16] buffer with ldx=16, k=80)."""
    lib = _lib.load()
    _need_gpu(x, weight, bias)
    bf16 = x.dtype == torch.bfloat16                 # bf16 storage: v_mfma_f32_32x32x16_bf16, fp32 accumulate
    dt = torch.bfloat16 if bf16 else torch.float32
    if not (x.dim() == 2 and x.dtype == dt and x.stride(1) == 1 and x.stride(0) >= x.shape[1]):
        x = _c(x, dt)                                # row-strided views (padded rows) are used in place
    weight = _c(weight, dt)
    bias = _c(bias, torch.float32)
    k = weight.shape[0] if w_is_km else weight.shape[1]
    m = weight.shape[1] if w_is_km else weight.shape[0]
    if n is None:
        if x.dim() != 2 or x.shape[1] != k:
            raise ValueError(f"linear: shape mismatch x {tuple(x.shape)} w {tuple(weight.shape)}")
        n, ldx = x.shape[0], x.stride(0)
    elif (n - 1) * ldx + k > x.numel():
        raise ValueError("linear: (n, ldx, k) window runs past the end of x")
    if out is None:
        out = torch.empty((n, m), dtype=dt, device=x.device)
    elif out.dtype != dt:
        raise TypeError("linear: `out` dtype must match x")
    fn = lib.lpg_linear_bf16 if bf16 else lib.lpg_linear_f32
    _lib.check(fn(_ptr(x), ldx, _ptr(weight), int(w_is_km), _ptr(bias), ACT[act], _ptr(out), out.stride(0), n, k, m,
                  _stream()), "linear")
    return out


def layer_norm_rows(x, eps=1e-5, weight=None, bias=None, relu=False):
    """torch.nn.LayerNorm over the last dimension of a [n, d] float32 tensor (d <= 1024), optional affine and ReLU"""
    lib = _lib.load()
    _need_gpu(x, weight, bias)
    x = _c(x, torch.float32)
    n, d = x.shape
    y = torch.empty_like(x)
    _lib.check(lib.lpg_layernorm_rows_f32(_ptr(x), x.stride(0), n, d, float(eps), _ptr(_c(weight, torch.float32)),
                                          _ptr(_c(bias, torch.float32)), int(relu), _ptr(y), y.stride(0), _stream()),
               "layernorm_rows")
    return y


_gn_ws = {}


def graph_norm(x, weight, bias, mean_scale, eps=1e-5, relu=False, add_to=None, out=None):
    lib = _lib.load()
    _need_gpu(x, weight, bias, mean_scale, add_to)
    x = _c(x, torch.float32)
    n, d = x.shape
    nbytes = _lib.check(lib.lpg_graphnorm_bytes(n, d), "graphnorm_bytes")
    key = (x.device, _stream())               # one scratch buffer per (device, stream): concurrent streams must not share it
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _gn_ws[key] = ws
    if out is None:
        out = torch.empty_like(x)
    add_to = _c(add_to, torch.float32)
    _lib.check(lib.lpg_graphnorm_f32(_ptr(x), n, d, _ptr(_c(weight, torch.float32)), _ptr(_c(bias, torch.float32)),
                                     _ptr(_c(mean_scale, torch.float32)), float(eps), int(relu), _ptr(add_to),
                                     _ptr(out), _ptr(ws), ws.numel(), _stream()), "graphnorm_f32")
    return out


def graph_norm_weighted(x, row_weight, n_total, weight, bias, mean_scale, eps=1e-5, relu=False):
    """GraphNorm of a [V, d] table whose row v stands for row_weight[v] (int64) identical rows of the real input,
    n_total rows in all (lpg_graphnorm_weighted_f32)."""
    lib = _lib.load()
    _need_gpu(x, row_weight, weight, bias, mean_scale)
    x = _c(x, torch.float32)
    row_weight = _c(row_weight, torch.int64)
    n, d = x.shape
    nbytes = _lib.check(lib.lpg_graphnorm_bytes(n, d), "graphnorm_bytes")
    key = (x.device, _stream())
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _gn_ws[key] = ws
    out = torch.empty_like(x)
    _lib.check(lib.lpg_graphnorm_weighted_f32(_ptr(x), n, d, _ptr(row_weight), int(n_total), _ptr(_c(weight, torch.float32)),
                                              _ptr(_c(bias, torch.float32)), _ptr(_c(mean_scale, torch.float32)), float(eps),
                                              int(relu), _ptr(out), _ptr(ws), ws.numel(), _stream()), "graphnorm_weighted")
    return out


# ------------------------------------------------------------------------------ pair ops
def embedding(table, idx, row_scale=None):
    """table[idx] (row_scale: every gathered row i times row_scale[i], fp32 [n])"""
    lib = _lib.load()
    _need_gpu(table, idx, row_scale)
    table = _c(table, torch.float32)
    idx = _c(idx.reshape(-1), torch.int64)
    out = torch.empty((idx.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
    if row_scale is not None:
        row_scale = _c(row_scale.reshape(-1), torch.float32)
        if row_scale.numel() != idx.numel():
            raise ValueError("embedding: row_scale needs one value per index")
        _lib.check(lib.lpg_embedding_scaled_f32(_ptr(table), table.shape[0], table.shape[1], _ptr(idx), idx.numel(),
                                                _ptr(row_scale), _ptr(out), _ptr(err_flag(table.device)), _stream()),
                   "embedding_scaled")
        return out
    _lib.check(lib.lpg_embedding_f32(_ptr(table), table.shape[0], table.shape[1], _ptr(idx), idx.numel(), _ptr(out),
                                     _ptr(err_flag(table.device)), _stream()), "embedding")
    return out


def pair_gather_mul(x, pos):
    lib = _lib.load()
    _need_gpu(x, pos)
    x = _c(x, torch.float32)
    pos = _c(pos, torch.int64)
    P = pos.shape[0]
    h = torch.empty((P, x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(lib.lpg_pair_gather_mul_f32(_ptr(x), x.shape[0], x.shape[1], _ptr(pos), P, _ptr(h),
                                           _ptr(err_flag(x.device)), _stream()), "pair_gather_mul")
    return h


def pair_readout(h, idx, w, b):
    lib = _lib.load()
    _need_gpu(h, idx, w, b)
    h = _c(h, torch.float32)
    idx = _c(idx.reshape(-1), torch.int64)
    B = idx.numel() // 2
    out = torch.empty((B, 1), dtype=torch.float32, device=h.device)
    _lib.check(lib.lpg_pair_readout_f32(_ptr(h), h.shape[0], h.shape[1], _ptr(idx), B,
                                        _ptr(_c(w.reshape(-1), torch.float32)), _ptr(_c(b, torch.float32)),
                                        _ptr(out), _ptr(err_flag(h.device)), _stream()), "pair_readout")
    return out


# ------------------------------------------------------------------------------ small-graph GCN blocks
_small_limits = None
SG_ROW_BLOCKS = None                # tests: force the number of row blocks of the small-graph aggregation


def small_max_rows():
    global _small_limits
    if _small_limits is None:
        lib = _lib.load()
        _small_limits = (int(lib.lpg_small_max_rows()), int(lib.lpg_small_max_row_blocks()))
    return _small_limits[0]


def small_row_blocks(e_cap):
    """row blocks a graph's aggregation is cut into (more workgroups, shorter edge strips per thread)."""
    small_max_rows()
    rb = int(SG_ROW_BLOCKS) if SG_ROW_BLOCKS else (1 if e_cap < 16384 else 2 if e_cap < 32768 else 4 if e_cap < 131072 else 8)
    return max(1, min(rb, _small_limits[1]))


class Deferred:
    """an un-normalised aggregation output y [n, C] plus what its consumers need to apply GraphNorm (+ ReLU) on
    load: per-row-block column sums `part` [rb, 2, C] (fp64) and the norm's parameters."""

    def __init__(self, y, part=None, norm=None, relu=False):
        self.y, self.part, self.norm, self.relu = y, part, norm, relu

    def args(self):
        if self.part is None:
            return [0, 0, 0, 0, 1, 0]
        nm = self.norm
        w, b, ms = _c(nm.weight, torch.float32), _c(nm.bias, torch.float32), _c(nm.mean_scale, torch.float32)
        _lib.keepalive(self.part, w, b, ms)
        return [_ptr(self.part), _ptr(w), _ptr(b), _ptr(ms), self.part.shape[0], int(self.relu)]

    @property
    def eps(self):
        return float(self.norm.eps) if self.norm is not None else 1e-5

    def materialize(self):
        """the normalised tensor (tests / debugging): plain GraphNorm math on the device in fp64."""
        if self.part is None:
            return self.y
        nm, n = self.norm, self.y.shape[0]
        s, s2 = self.part[:, 0, :].sum(0), self.part[:, 1, :].sum(0)
        mean, ms = s / n, nm.mean_scale.double()
        var = (s2 / n - mean * mean * ms * (2.0 - ms)).clamp_min(0.0)
        o = (self.y.double() - mean * ms) * (nm.weight.double() / torch.sqrt(var + nm.eps)) + nm.bias.double()
        return (o.clamp_min(0.0) if self.relu else o).float()


def embedding_graph_norm(table, idx, weight, bias, mean_scale, eps=1e-5, relu=False):
    """GraphNorm(table[idx]) in one launch (n <= 8192 rows, d % 4 == 0)."""
    lib = _lib.load()
    _need_gpu(table, idx, weight, bias, mean_scale)
    table = _c(table, torch.float32)
    idx = _c(idx.reshape(-1), torch.int64)
    n, d = idx.numel(), table.shape[1]
    out = torch.empty((n, d), dtype=torch.float32, device=table.device)
    weight, bias, mean_scale = _c(weight, torch.float32), _c(bias, torch.float32), _c(mean_scale, torch.float32)
    _lib.keepalive(table, idx, weight, bias, mean_scale, out)
    _lib.call("lpg_embedding_graphnorm_f32", _ptr(table), table.shape[0], d, _ptr(idx), n, _ptr(weight), _ptr(bias),
              _ptr(mean_scale), float(eps), int(relu), _ptr(out), _ptr(err_flag(table.device)), _stream())
    return out


def small_linear_t(xa, weight, weight2=None, xb=None, pos=None):
    """(X W^T)^T as [m, ldt] (columns contiguous, ldt = n rounded up to 4); X = N(xa), N(xa)[pos[:,0]] * N(xa)[pos[:,1]]
    (pos given) or N(xa) + N(xb) (xb given), N = the deferred GraphNorm when the operand is a `Deferred`.
    weight2: a second [m,k] matrix over the same X (second return value)."""
    lib = _lib.load()
    xa = xa if isinstance(xa, Deferred) else Deferred(xa)
    xb = xb if (xb is None or isinstance(xb, Deferred)) else Deferred(xb)
    ya, yb = _c(xa.y, torch.float32), (None if xb is None else _c(xb.y, torch.float32))
    _need_gpu(ya, yb, pos, weight, weight2)
    weight, weight2 = _c(weight, torch.float32), _c(weight2, torch.float32)
    m, k = weight.shape
    if ya.dim() != 2 or ya.shape[1] != k or (weight2 is not None and weight2.shape != weight.shape):
        raise ValueError("small_linear_t: shape mismatch")
    mode, n, n_src = 0, ya.shape[0], ya.shape[0]
    if pos is not None:
        pos = _c(pos, torch.int64)
        mode, n = 1, pos.shape[0]
    elif yb is not None:
        if yb.shape != ya.shape:
            raise ValueError("small_linear_t: xa / xb shape mismatch")
        mode = 2
    ldt = (n + 3) // 4 * 4
    hT = torch.empty((m, ldt), dtype=torch.float32, device=ya.device)
    h2T = torch.empty((m, ldt), dtype=torch.float32, device=ya.device) if weight2 is not None else None
    nb = xb.args() if xb is not None else Deferred(None).args()
    _lib.keepalive(ya, yb, pos, weight, weight2, hT, h2T)
    _lib.call("lpg_small_linear_t_f32", _ptr(ya), _ptr(yb), _ptr(pos), mode, n_src, n, k, _ptr(weight), _ptr(weight2),
              m, _ptr(hT), _ptr(h2T), ldt, *xa.args(), *nb, xa.eps, _ptr(err_flag(ya.device)), _stream())
    return (hT, h2T) if weight2 is not None else hT


def small_gcn(n, sides):
    """sides: one or two tuples (hT [c, ldt], graph: gcn-normalised unit-weight CSRGraph of n rows, conv bias, norm
    module, relu); one launch; returns a `Deferred` (un-normalised A_hat h + bias, column-sum partials) per side."""
    lib = _lib.load()
    c, ldt = sides[0][0].shape
    e_cap = sides[0][1].e_cap
    rb = small_row_blocks(e_cap)
    outs, args = [], []
    for hT, g, b, norm, relu in sides:
        _need_gpu(hT, b)
        if (hT.shape != (c, ldt) or not hT.is_contiguous() or g.n_rows != n or g.n_cols != n or not g.unit_gcn
                or g.e_cap != e_cap):
            raise ValueError("small_gcn: side does not match (square unit-weight gcn-normalised graph of n rows)")
        y = torch.empty((n, c), dtype=torch.float32, device=hT.device)
        part = torch.empty((rb, 2, c), dtype=torch.float64, device=hT.device)
        outs.append(Deferred(y, part, norm, relu))
        b = _c(b, torch.float32)
        pack = g.small_pack(rb)
        _lib.keepalive(hT, pack, g.rowptr, g.dinv, b, y, part)
        args += [_ptr(hT), _ptr(pack), _ptr(g.rowptr), _ptr(g.dinv), _ptr(b), _ptr(y), _ptr(part)]
    if len(sides) == 1:
        args += [0] * 7
    _lib.call("lpg_small_gcn_f32", n, c, ldt, c, rb, e_cap, *args, _stream())
    return outs


SMALL_EMBED_MAX = 1024      # table rows / columns lpg_small_gcn_embed_f32 takes (a 1024-thread workgroup: one thread each)


def small_gcn_embed(n, table, idx, gn_weight, gn_bias, gn_mean_scale, eps, lin_weight, g, bias, norm, relu):
    """conv1 of a model without node features in ONE launch: A_hat (GraphNorm(table[idx]) W^T) + bias, un-normalised,
    as a `Deferred` (lpg_small_gcn_embed_f32).  table [n_table, k] (both <= 1024), idx int64 [n], lin_weight [c, k],
    g: gcn-normalised unit-weight CSRGraph of n rows."""
    table, lin_weight = _c(table, torch.float32), _c(lin_weight, torch.float32)
    idx = _c(idx.reshape(-1), torch.int64)
    gw, gb, gms, b = (_c(v, torch.float32) for v in (gn_weight, gn_bias, gn_mean_scale, bias))
    _need_gpu(table, idx, lin_weight, gw, gb, gms, b)
    c, k = lin_weight.shape
    if table.shape[1] != k or idx.numel() != n or g.n_rows != n or g.n_cols != n or not g.unit_gcn:
        raise ValueError("small_gcn_embed: shapes do not match (table [*, k], idx [n], W [c, k], square gcn graph of n rows)")
    rb = small_row_blocks(g.e_cap)
    y = torch.empty((n, c), dtype=torch.float32, device=table.device)
    part = torch.empty((rb, 2, c), dtype=torch.float64, device=table.device)
    pack = g.small_pack(rb)
    _lib.keepalive(table, idx, lin_weight, gw, gb, gms, b, pack, g.rowptr, g.dinv, y, part)
    _lib.call("lpg_small_gcn_embed_f32", n, c, c, rb, g.e_cap, _ptr(table), table.shape[0], k, _ptr(idx), _ptr(gw), _ptr(gb),
              _ptr(gms), float(eps), _ptr(lin_weight), _ptr(pack), _ptr(g.rowptr), _ptr(g.dinv), _ptr(b), _ptr(y), _ptr(part),
              _ptr(err_flag(table.device)), _stream())
    return Deferred(y, part, norm, relu)


def small_readout(a, r, idx, w, b):
    """pair readout over N(a) + N(r) (r may be None); a, r: `Deferred` [P, C]."""
    lib = _lib.load()
    ya, yr = _c(a.y, torch.float32), (None if r is None else _c(r.y, torch.float32))
    _need_gpu(ya, yr, idx, w, b)
    if yr is not None and yr.shape != ya.shape:
        raise ValueError("small_readout: shape mismatch")
    idx = _c(idx.reshape(-1), torch.int64)
    B = idx.numel() // 2
    out = torch.empty((B, 1), dtype=torch.float32, device=ya.device)
    nr = r.args() if r is not None else Deferred(None).args()
    w, b = _c(w.reshape(-1), torch.float32), _c(b, torch.float32)
    _lib.keepalive(ya, yr, idx, w, b, out)
    _lib.call("lpg_small_readout_f32", _ptr(ya), _ptr(yr), ya.shape[0], ya.shape[1], *a.args(), *nr, a.eps, _ptr(idx),
              B, _ptr(w), _ptr(b), _ptr(out), _ptr(err_flag(ya.device)), _stream())
    return out


def flip_index(ei2):
    lib = _lib.load()
    _need_gpu(ei2)
    ei2 = _c(ei2, torch.int64)
    e, er = torch.empty_like(ei2), torch.empty_like(ei2)
    _lib.check(lib.lpg_flip_index_i64(_ptr(ei2), ei2.shape[1], _ptr(e), _ptr(er), _stream()), "flip_index")
    return e, er


# ------------------------------------------------------------------------------ tuple graph
def get_ei2(n_node, pos_edge, pred_edge):
    """TwoWL/utils.py:36-45 on the device; returns int64 [2, E2] in the reference's order."""
    lib = _lib.load()
    _need_gpu(pos_edge, pred_edge)
    import ctypes
    pos_edge = _c(pos_edge, torch.int64)
    pred_edge = _c(pred_edge, torch.int64)
    E1, Ep = pos_edge.shape[1], pred_edge.shape[1]
    dev = pos_edge.device
    nbytes = _lib.check(lib.lpg_get_ei2_bytes(n_node, E1, Ep), "get_ei2_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    e2 = ctypes.c_int64(0)
    _lib.check(lib.lpg_get_ei2_count(_ptr(pos_edge), E1, _ptr(pred_edge), Ep, n_node, _ptr(ws), nbytes,
                                     ctypes.addressof(e2), _ptr(err_flag(dev)), _stream()), "get_ei2_count")
    raise_if_flagged(dev, "get_ei2 node id")
    E2 = int(e2.value)
    ei2 = torch.empty((2, E2), dtype=torch.int64, device=dev)
    _lib.check(lib.lpg_get_ei2_fill(_ptr(ws), n_node, E1, Ep, E2, _ptr(ei2), _stream()), "get_ei2_fill")
    return ei2


def mark(idx, n_mask):
    lib = _lib.load()
    idx = _c(idx.reshape(-1), torch.int64)
    mask = torch.zeros(n_mask, dtype=torch.uint8, device=idx.device)
    _lib.check(lib.lpg_mark_i64(_ptr(idx), idx.numel(), _ptr(mask), n_mask, _ptr(err_flag(idx.device)), _stream()),
               "mark")
    return mask


def keep_from_blocked(key, E, blocked, invert=False):
    lib = _lib.load()
    key = _c(key, torch.int64)
    keep = torch.empty(E, dtype=torch.uint8, device=blocked.device)
    _lib.check(lib.lpg_keep_from_blocked(_ptr(key), E, _ptr(blocked), blocked.numel(), int(invert), _ptr(keep),
                                         _ptr(err_flag(blocked.device)), _stream()), "keep_from_blocked")
    return keep


def compact_cols(mat, keep):
    """mat[:, keep != 0] with the column order preserved (stable)."""
    lib = _lib.load()
    import ctypes
    mat = _c(mat, torch.int64)
    rows, E = mat.shape
    nbytes = _lib.check(lib.lpg_compact_cols_bytes(E), "compact_cols_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=mat.device)
    out = torch.empty(max(rows * E, 1), dtype=torch.int64, device=mat.device)
    cnt = ctypes.c_int64(0)
    _lib.check(lib.lpg_compact_cols_i64(_ptr(mat), rows, E, _ptr(keep), _ptr(out), _ptr(ws), nbytes,
                                        ctypes.addressof(cnt), _stream()), "compact_cols")
    c = int(cnt.value)
    return out[: rows * c].view(rows, c)


# ------------------------------------------------------------------------------ 2-WL layers without the tuple graph
TUPLE_MAX_C = 32          # channels the tuple-layer kernels take (lanes 0-31 / 32-63 = the two orientations)
TUPLE_MAX_PART = 64       # partial-sum blocks a consumer folds by itself
TUPLE_AX_SMALL_MAX_ROW = 2048  # longest member row for which A x takes the workgroup-per-node kernel (small graphs)


class TuplePlan:
    """device-side structure of a twowl.TupleGraph (include/lpg_hip.h: lpg_tuple_plan_build): the member adjacency
    `A` (unit-weight CSRGraph, rows = start node), cnt [N], qpos [P], qinfo [Q, 4], and for the operator form
    tnode [P, 2], seg_a / seg_r [E1].  ok: pos_edge is in double()'s layout and the blocked ids come in pairs."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def tuple_plan_build(n_node, pos_edge, pred_edge, blocked=None, n_blocked_max=0, operator_form=False):
    """blocked: uint8 mask [E1] or None; n_blocked_max: an upper bound of its set entries (sizes qinfo)."""
    lib = _lib.load()
    _need_gpu(pos_edge, pred_edge, blocked)
    pos_edge, pred_edge = _c(pos_edge, torch.int64), _c(pred_edge, torch.int64)
    dev = pos_edge.device
    E1, Ep = pos_edge.shape[1], pred_edge.shape[1]
    P = E1 + Ep
    if E1 % 2 or Ep % 2 or n_node < 1:
        return TuplePlan(ok=False, n_node=n_node, E1=E1, Ep=Ep, P=P, Q=Ep)
    if blocked is not None:
        keep = keep_from_blocked(None, E1, blocked)
        members = compact_cols(pos_edge, keep)
    else:
        members = pos_edge
    A = CSRGraph.from_edge_index(members, n_node, n_node)               # symmetric: rows = end = start side
    q_cap = Ep + (min(int(n_blocked_max), E1) if blocked is not None else 0)
    cnt = torch.empty(n_node, dtype=torch.int32, device=dev)
    qpos = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
    qinfo = torch.empty((max(q_cap, 1), 4), dtype=torch.int32, device=dev)
    tnode = torch.empty((max(P, 1), 2), dtype=torch.int32, device=dev) if operator_form else None
    seg_a = torch.empty(max(E1, 1), dtype=torch.int64, device=dev) if operator_form else None
    seg_r = torch.empty(max(E1, 1), dtype=torch.int64, device=dev) if operator_form else None
    head = torch.empty(4, dtype=torch.int32, device=dev)
    nbytes = _lib.check(lib.lpg_tuple_plan_bytes(P), "tuple_plan_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.lpg_tuple_plan_build(_ptr(pos_edge), E1, _ptr(pred_edge), Ep, n_node, _ptr(blocked), _ptr(A.rowptr),
                                        _ptr(cnt), _ptr(qpos), _ptr(qinfo), q_cap, _ptr(tnode), _ptr(seg_a), _ptr(seg_r),
                                        _ptr(head), _ptr(ws), nbytes, _ptr(err_flag(dev)), _stream()), "tuple_plan_build")
    raise_if_flagged(dev, "tuple graph node id")                        # the build's one sync
    h = head.tolist()
    return TuplePlan(ok=(h[1] == 0), n_node=int(n_node), E1=E1, Ep=Ep, P=P, Q=int(h[0]), A=A, cnt=cnt, qpos=qpos,
                     qinfo=qinfo, tnode=tnode, seg_a=seg_a[:E1] if operator_form else None,
                     seg_r=seg_r[:E1] if operator_form else None, pos_edge=pos_edge, pred_edge=pred_edge, q_cap=q_cap)


def _norm_args(norm, relu):
    w, b, ms = _c(norm.weight, torch.float32), _c(norm.bias, torch.float32), _c(norm.mean_scale, torch.float32)
    _lib.keepalive(w, b, ms)
    return [_ptr(w), _ptr(b), _ptr(ms), float(norm.eps), int(relu)]


def tuple_ax_small(plan, x1):
    """(x1n, A x1n) over a TuplePlan's member adjacency for graphs with short member rows: a workgroup per node
    (lpg_tuple_ax_small_f32).  x1: the 1-WL output [N, C], a tensor or a `Deferred` whose GraphNorm (+ ReLU) is applied
    on load."""
    dfr = x1 if isinstance(x1, Deferred) else None
    y1 = _c(dfr.y if dfr is not None else x1, torch.float32)
    _need_gpu(y1)
    N, C = y1.shape
    if N != plan.n_node or C > TUPLE_MAX_C:
        raise ValueError("tuple_ax_small: x1 must be [n_node, C <= 32]")
    if dfr is not None and dfr.part is not None:
        nm = dfr.norm
        gw, gb, gms = _c(nm.weight, torch.float32), _c(nm.bias, torch.float32), _c(nm.mean_scale, torch.float32)
        din = [_ptr(dfr.part), _ptr(gw), _ptr(gb), _ptr(gms), dfr.part.shape[0], int(dfr.relu), float(nm.eps)]
        _lib.keepalive(dfr.part, gw, gb, gms)
    else:
        din = [0, 0, 0, 0, 1, 0, 1e-5]
    x1n, ax = torch.empty_like(y1), torch.empty_like(y1)
    _lib.keepalive(y1, plan.A.rowptr, plan.A.col, x1n, ax)
    _lib.call("lpg_tuple_ax_small_f32", N, C, _ptr(y1), y1.stride(0), _ptr(plan.A.rowptr), _ptr(plan.A.col), *din, _ptr(x1n),
              _ptr(ax), _stream())
    return x1n, ax


def tuple_layer(plan, mode, x, ax, w_a, w_r, b_a, b_r, hq_prev=None, want_t=True, reduce_q=None):
    """one 2-WL layer over a TuplePlan (lpg_tuple_layer_f32).  mode 1: x = the 1-WL output [N, C], ax = A x;
    mode 2: x = (F_a | F_r) [N, 2C], ax = A x, hq_prev [Q, C].
    reduce_q: callable(tensor [2, 64] float64) summing it in place over the ranks of a sharded candidate list.
    -> (t_node [N, 2C] or None, y_q [Q, 2C], part [n_part, 2, 64])"""
    lib = _lib.load()
    xt, ax = _c(x, torch.float32), _c(ax, torch.float32)
    w_a, w_r, b_a, b_r = (_c(v, torch.float32) for v in (w_a, w_r, b_a, b_r))
    _need_gpu(xt, ax, w_a, w_r, b_a, b_r, hq_prev)
    C = w_a.shape[0]
    N, Q = plan.n_node, plan.Q
    if w_a.shape != (C, C) or w_r.shape != (C, C) or C > TUPLE_MAX_C or xt.shape != (N, C if mode == 1 else 2 * C):
        raise ValueError("tuple_layer: shapes do not match (square [C, C] weights, C <= 32, x [N, C] / [N, 2C])")
    if ax.shape != xt.shape:
        raise ValueError("tuple_layer: A x must have x's shape")
    dev = xt.device
    nb = int(_lib.check(lib.lpg_tuple_layer_blocks(N, Q), "tuple_layer_blocks"))
    t_node = torch.empty((N, 2 * C), dtype=torch.float32, device=dev) if want_t else None
    y_q = torch.empty((max(Q, 1), 2 * C), dtype=torch.float32, device=dev)
    part = torch.empty((2 * nb, 2, 64), dtype=torch.float64, device=dev)     # even blocks: member rows, odd: non-member rows
    hq_prev = _c(hq_prev, torch.float32)
    _lib.keepalive(xt, ax, hq_prev, plan.cnt, plan.qinfo, w_a, w_r, b_a, b_r, t_node, y_q, part)
    _lib.call("lpg_tuple_layer_f32", int(mode), N, Q, C, _ptr(xt), xt.stride(0), _ptr(ax), ax.stride(0), _ptr(hq_prev),
              _ptr(plan.cnt), _ptr(plan.qinfo), _ptr(w_a), _ptr(w_r), _ptr(b_a), _ptr(b_r), _ptr(t_node), _ptr(y_q),
              _ptr(part), nb, _stream())
    if 2 * nb > TUPLE_MAX_PART or reduce_q is not None:     # big graphs (or sharded candidates): one more small launch folds the partial sums
        n_out = 2 if reduce_q is not None else 16
        folded = torch.empty((n_out, 2, 64), dtype=torch.float64, device=dev)
        _lib.check(lib.lpg_tuple_fold_f64(_ptr(part), 2 * nb, _ptr(folded), n_out, _stream()), "tuple_fold")
        part = folded
        if reduce_q is not None:
            reduce_q(part[1])                   # the non-member rows' sums, summed over the ranks that hold the other candidates
    return t_node, y_q, part


def tuple_apply(plan, t_node, y_q, part, norm_a, relu_a, norm_r, relu_r, b_a, b_r, n_rows=None):
    """GraphNorm (+ ReLU) of a 2-WL layer's two outputs, for the next layer: ff [N, 2C] = (F_a | F_r), hq [Q, C]"""
    lib = _lib.load()
    C = t_node.shape[1] // 2
    b_a, b_r = _c(b_a, torch.float32), _c(b_r, torch.float32)
    ff = torch.empty_like(t_node)
    hq = torch.empty((max(plan.Q, 1), C), dtype=torch.float32, device=t_node.device)
    _lib.check(lib.lpg_tuple_apply_f32(plan.n_node, plan.Q, C, _ptr(t_node), _ptr(y_q), _ptr(plan.cnt), _ptr(part),
                                       part.shape[0], plan.P if n_rows is None else int(n_rows), *_norm_args(norm_a, relu_a),
                                       *_norm_args(norm_r, relu_r), _ptr(b_a), _ptr(b_r), _ptr(ff), _ptr(hq), _stream()), "tuple_apply")
    return ff, hq


def tuple_readout(plan, t_node, y_q, part, norm_a, relu_a, norm_r, relu_r, b_a, b_r, idx, w, b0, n_rows=None):
    """TwoWL/model/model.py:78-83 over the last 2-WL layer (GraphNorm + ReLU applied on load) -> [B, 1] logits"""
    C = t_node.shape[1] // 2
    idx = _c(idx.reshape(-1), torch.int64)
    if idx.data_ptr() % 16:
        idx = idx.clone()                    # the kernel reads a pair's two ids as one 16-byte piece
    _need_gpu(t_node, y_q, idx, w, b0)
    B = idx.numel() // 2
    out = torch.empty((B, 1), dtype=torch.float32, device=t_node.device)
    b_a, b_r, w, b0 = _c(b_a, torch.float32), _c(b_r, torch.float32), _c(w.reshape(-1), torch.float32), _c(b0, torch.float32)
    na, nr = _norm_args(norm_a, relu_a), _norm_args(norm_r, relu_r)
    _lib.keepalive(plan.pos_edge, t_node, y_q, plan.cnt, plan.qpos, part, b_a, b_r, idx, w, b0, out)
    _lib.call("lpg_tuple_readout_f32", plan.n_node, plan.Q, C, plan.E1, plan.P, _ptr(plan.pos_edge), _ptr(t_node), _ptr(y_q),
              _ptr(plan.cnt), _ptr(plan.qpos), _ptr(part), part.shape[0], plan.P if n_rows is None else int(n_rows), *na, *nr,
              _ptr(b_a), _ptr(b_r), _ptr(idx), B,
              _ptr(w), _ptr(b0), _ptr(out), _ptr(err_flag(t_node.device)), _stream())
    return out


def tuple_gcn_operator(plan, orient, x, weight, bias):
    """GCNConv(x, edge2 | edge2_r) for an ARBITRARY [P, C_in] input without the tuple graph (the operator form:
    TwoWL/model/model.py:77 called module by module): group sums of the member rows (the atomic-free scatter_add),
    the layer's linear on N + Q rows instead of P, and one broadcast launch."""
    lib = _lib.load()
    x = _c(x, torch.float32)
    _need_gpu(x, weight, bias)
    P, N, E1 = plan.P, plan.n_node, plan.E1
    if x.shape[0] != P:
        raise ValueError(f"tuple_gcn_operator: x must have one row per tuple ({P}), got {x.shape[0]}")
    sums = torch.zeros((N + 1, x.shape[1]), dtype=torch.float32, device=x.device)      # row N collects the blocked tuples
    if E1:
        scatter_add_rows(x[:E1], plan.seg_a if orient == 0 else plan.seg_r, sums, fresh=False)
    t_node = linear(sums[:N], weight)
    C = t_node.shape[1]
    hq = None
    if plan.Q:
        qrows = plan.qinfo[: plan.Q, 3].to(torch.int64)
        hq = linear(embedding(x, qrows), weight)
    y = torch.empty((P, C), dtype=torch.float32, device=x.device)
    bias = _c(bias, torch.float32)
    _lib.check(lib.lpg_tuple_bcast_f32(P, C, int(orient), _ptr(plan.tnode), _ptr(plan.cnt), _ptr(plan.qpos), _ptr(t_node),
                                       t_node.stride(0), _ptr(hq), 0 if hq is None else hq.stride(0), _ptr(bias), _ptr(y),
                                       y.stride(0), _stream()), "tuple_bcast")
    return y
