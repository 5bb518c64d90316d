"""Oracle (TEST INFRASTRUCTURE): fp64 numpy restatement of the float aggregation path.

PARITY UNPINNED at the third-party boundary (see oracle/__init__.py): torch_scatter,
torch_sparse, torch_geometric and tensorflow are absent and the reference stores no outputs
for them.  Restated from the published definitions of the pinned versions
(requirements.txt:1,18-20), anchored on the reference's call sites, which are cited per
function.  Everything runs in float64 so the HIP fp32 path can be held to 1e-5 relative.
"""
import numpy as np


# ----------------------------------------------------------------------------- scatter / spmm
def scatter_add(src, index, dim_size=None, dtype=None):
    """torch_scatter.scatter_add(src, index, dim=0, dim_size=N) [3P, torch-scatter 2.1.1]
    as called at TwoWL/utils.py:10: out = zeros(N, ...); out[index[e]] += src[e]."""
    src = np.asarray(src)
    index = np.asarray(index, dtype=np.int64)
    n = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.size else 0)
    out = np.zeros((n,) + src.shape[1:], dtype=dtype or src.dtype)
    np.add.at(out, index, src)
    return out


def coo_to_csr(row, col, n_rows, val=None):
    """stable sort by row -> (rowptr[n+1], col, val, perm); within a row the original edge
    order is kept (the order torch's CPU scatter_add_ accumulates in)."""
    row = np.asarray(row, dtype=np.int64)
    perm = np.argsort(row, kind="stable")
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=n_rows), out=rowptr[1:])
    col = np.asarray(col, dtype=np.int64)[perm]
    val = None if val is None else np.asarray(val)[perm]
    return rowptr, col, val, perm


def csr_spmm(rowptr, col, val, x, row_scale=None, bias=None, act=None):
    """Y = act(row_scale * (A @ X) + bias), A in CSR; float64.
    torch_sparse.spmm / matmul(reduce='sum') [3P, torch-sparse 0.6.17]; SEAL's
    tanh(D^-1 (A~ (XW))) at SEAL/model/gnn.py:47-48 is row_scale = D^-1, act = tanh."""
    x = np.asarray(x, dtype=np.float64)
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    msg = x[np.asarray(col, dtype=np.int64)]
    if val is not None:
        msg = msg * np.asarray(val, dtype=np.float64)[:, None]
    y = np.zeros((n, x.shape[1]), dtype=np.float64)
    np.add.at(y, rows, msg)
    if row_scale is not None:
        y = y * np.asarray(row_scale, dtype=np.float64)[:, None]
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float64)[None, :]
    if act == "tanh":
        y = np.tanh(y)
    elif act == "relu":
        y = np.maximum(y, 0.0)
    return y


def spmm_abs_rowsum(rowptr, col, val, x):
    """sum_e |w_e x_e| per output element: the scale the 1e-5 tolerance is relative to."""
    v = None if val is None else np.abs(val)
    return csr_spmm(rowptr, col, v, np.abs(np.asarray(x, dtype=np.float64)))


# ----------------------------------------------------------------------------- GCNConv pieces
def add_remaining_self_loops(edge_index, edge_weight, n, fill_value=1.0):
    """torch_geometric.utils.add_remaining_self_loops [3P, PyG 2.3.1], used by gcn_norm:
    existing self loops are REMOVED from the edge list and every node gets exactly one loop,
    appended after the non-loop edges, weight = the node's existing loop weight (the last one
    if several) else fill_value."""
    ei = np.asarray(edge_index, dtype=np.int64)
    w = np.ones(ei.shape[1], dtype=np.float64) if edge_weight is None else np.asarray(edge_weight, dtype=np.float64)
    mask = ei[0] != ei[1]
    loop_w = np.full(n, fill_value, dtype=np.float64)
    inv = ~mask
    loop_w[ei[0][inv]] = w[inv]          # sequential assignment: last one wins
    loops = np.arange(n, dtype=np.int64)
    ei2 = np.concatenate([ei[:, mask], np.stack([loops, loops])], axis=1)
    return ei2, np.concatenate([w[mask], loop_w])


def gcn_norm(edge_index, n, edge_weight=None):
    """torch_geometric.nn.conv.gcn_conv.gcn_norm [3P] with the defaults GCNConv(insize,
    outsize) uses at TwoWL/model/model.py:37 (improved=False, add_self_loops=True,
    flow='source_to_target'): deg by DESTINATION (row 1); w' = deg^-1/2[src] w deg^-1/2[dst],
    inf -> 0."""
    ei, w = add_remaining_self_loops(edge_index, edge_weight, n, 1.0)
    deg = np.zeros(n, dtype=np.float64)
    np.add.at(deg, ei[1], w)
    with np.errstate(divide="ignore"):
        dinv = deg ** -0.5
    dinv[np.isinf(dinv)] = 0.0
    return ei, dinv[ei[0]] * w * dinv[ei[1]]


def gcn_conv(x, edge_index, weight, bias, edge_weight=None):
    """GCNConv.forward [3P]: out = A_hat (x W^T) + b; out[dst] += w' * h[src]
    (call sites TwoWL/model/model.py:73,77).  weight is [out,in] as nn.Linear stores it."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    h = x @ np.asarray(weight, dtype=np.float64).T
    ei, w = gcn_norm(edge_index, n, edge_weight)
    out = np.zeros((n, h.shape[1]), dtype=np.float64)
    np.add.at(out, ei[1], h[ei[0]] * w[:, None])
    if bias is not None:
        out = out + np.asarray(bias, dtype=np.float64)[None, :]
    return out


def graph_norm(x, weight, bias, mean_scale, eps=1e-5):
    """torch_geometric.nn.GraphNorm.forward(x, batch=None) [3P] (TwoWL/model/model.py:38,54;
    single graph because Seq.forward passes only `out` on, :92-96):
    o = x - mean * mean_scale;  y = weight * o / sqrt(mean(o^2) + eps) + bias   (column-wise)."""
    x = np.asarray(x, dtype=np.float64)
    mean = x.mean(axis=0, keepdims=True)
    o = x - mean * np.asarray(mean_scale, dtype=np.float64)[None, :]
    var = (o * o).mean(axis=0, keepdims=True)
    return np.asarray(weight, dtype=np.float64)[None, :] * o / np.sqrt(var + eps) + np.asarray(bias, dtype=np.float64)[None, :]


def layer_norm(x, eps=1e-5):
    """torch.nn.LayerNorm(d, elementwise_affine=False) (TwoWL/model/model.py:30): per row, biased variance."""
    x = np.asarray(x, dtype=np.float64)
    mu = x.mean(axis=1, keepdims=True)
    return (x - mu) / np.sqrt(((x - mu) ** 2).mean(axis=1, keepdims=True) + eps)


def input_features(p, x_deg):
    """TwoWL/model/model.py:71 -- lin1(node_feat) (Dropout -> Linear -> LayerNorm -> Dropout, eval: :28-33,47-51) when
    the model was built with use_node_feat, else GraphNorm(Embedding(x)) (:53-55)."""
    if "lin1.weight" in p:
        h = np.asarray(p["node_feat"], dtype=np.float64) @ np.asarray(p["lin1.weight"], dtype=np.float64).T
        return layer_norm(h + np.asarray(p["lin1.bias"], dtype=np.float64)[None, :], p.get("lin1.eps", 1e-5))
    x = np.asarray(p["emb.weight"], dtype=np.float64)[np.asarray(x_deg, dtype=np.int64)]
    return graph_norm(x, p["emb_norm.weight"], p["emb_norm.bias"], p["emb_norm.mean_scale"])


# ----------------------------------------------------------------------------- TwoWL forward
def pair_hadamard(x, pos):
    """TwoWL/model/model.py:75 -- h[t] = x[pos[t,0]] * x[pos[t,1]]."""
    pos = np.asarray(pos, dtype=np.int64)
    return x[pos[:, 0]] * x[pos[:, 1]]


def pair_readout(h, idx, w, b):
    """TwoWL/model/model.py:78-83 -- x = h[idx]; s = Linear(x[even] * x[odd])."""
    idx = np.asarray(idx, dtype=np.int64)
    xx = h[idx]
    z = xx[0::2] * xx[1::2]
    return z @ np.asarray(w, dtype=np.float64).T + np.asarray(b, dtype=np.float64)[None, :]


def localwl_forward(params, x_deg, edge1, pos, idx, ei2):
    """LocalWLNet.forward in eval mode (TwoWL/model/model.py:68-84; dropout = identity).
    params: dict from lpgnn_amd.twowl.LocalWLNet.export_params() (plain numpy arrays)."""
    from .twowl_int import reverse
    edge2, edge2_r = reverse(ei2)
    p = params
    x = input_features(p, x_deg)
    for l in range(p["depth1"]):
        x = gcn_conv(x, edge1, p[f"conv1s.{l}.lin.weight"], p[f"conv1s.{l}.bias"])
        x = graph_norm(x, p[f"conv1s.{l}.norm.weight"], p[f"conv1s.{l}.norm.bias"], p[f"conv1s.{l}.norm.mean_scale"])
        if p[f"conv1s.{l}.act"]:
            x = np.maximum(x, 0.0)
    x = pair_hadamard(x, pos)
    for l in range(p["depth2"]):
        outs = []
        for name, e in (("conv2s", edge2), ("conv2s_r", edge2_r)):
            y = gcn_conv(x, e, p[f"{name}.{l}.lin.weight"], p[f"{name}.{l}.bias"])
            y = graph_norm(y, p[f"{name}.{l}.norm.weight"], p[f"{name}.{l}.norm.bias"], p[f"{name}.{l}.norm.mean_scale"])
            outs.append(np.maximum(y, 0.0))
        x = outs[0] + outs[1]
    return pair_readout(x, idx, p["pred.weight"], p["pred.bias"])


# ----------------------------------------------------------------------------- SEAL graph conv
def seal_graph_conv(a_tilde, d_inv, x, weights):
    """SEAL/model/gnn.py:46-63 -- Z_l = tanh(D^-1 (A~ (Z_{l-1} W_l))) for l = 1..4 on ONE
    subgraph (dense operands as the reference feeds them); returns (concat(Z1,Z2,Z3), Z4[:,0])."""
    a = np.asarray(a_tilde, dtype=np.float64)
    d = np.asarray(d_inv, dtype=np.float64)
    z = np.asarray(x, dtype=np.float64)
    zs = []
    for w in weights:
        z = np.tanh(d[:, None] * (a @ (z @ np.asarray(w, dtype=np.float64))))
        zs.append(z)
    return np.concatenate(zs[:3], axis=1), zs[3][:, 0]


def auc(y_true, score):
    """rank-based ROC AUC (ties get average rank) == sklearn.metrics.roc_auc_score used at
    TwoWL/model/train.py:66 and SEAL/operators/seal_link_predict.py:91."""
    y = np.asarray(y_true).reshape(-1).astype(bool)
    s = np.asarray(score, dtype=np.float64).reshape(-1)
    order = np.argsort(s, kind="mergesort")
    ranks = np.empty(len(s), dtype=np.float64)
    ss = s[order]
    i = 0
    while i < len(ss):
        j = i
        while j + 1 < len(ss) and ss[j + 1] == ss[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    npos, nneg = y.sum(), (~y).sum()
    return (ranks[y].sum() - npos * (npos + 1) / 2.0) / (npos * nneg)


def seal_head(z, key, hw):
    """SEAL/model/gnn.py:82-137 on ONE subgraph, fp64 (PARITY UNPINNED: TensorFlow absent).
    z = concat(Z1,Z2,Z3) [n,96], key = Z4[:,0]; hw = dict of the checkpoint's head weights.
    SortPooling: tf.nn.top_k(key, n) order (descending, ties -> lower index), first k rows, zero pad;
    conv1d(96 -> 16, stride 96) == per-row matmul; conv1d(width 5, 16 -> 32, VALID); flatten;
    relu(x W1 + b1); x W2 + b2; softmax -> P(class 1)."""
    z = np.asarray(z, dtype=np.float64)
    key = np.asarray(key, dtype=np.float64)
    w1 = np.asarray(hw["weight_1"], dtype=np.float64)
    k = w1.shape[0] // 32 + 4
    order = np.lexsort((np.arange(len(key)), -key))            # key descending, index ascending
    pooled = np.zeros((k, z.shape[1]))
    m = min(k, len(key))
    pooled[:m] = z[order[:m]]
    a = pooled @ np.asarray(hw["conv1d_kernel_1"], dtype=np.float64).reshape(96, 16)
    k2 = np.asarray(hw["conv1d_kernel_2"], dtype=np.float64)    # [5, 16, 32]
    b = np.stack([sum(a[t + dt] @ k2[dt] for dt in range(5)) for t in range(k - 4)])
    dense = np.maximum(b.reshape(-1) @ w1 + np.asarray(hw["bias_1"], dtype=np.float64), 0.0)
    pre_y = dense @ np.asarray(hw["weight_2"], dtype=np.float64) + np.asarray(hw["bias_2"], dtype=np.float64)
    e = np.exp(pre_y - pre_y.max())
    return pre_y, float(e[1] / e.sum())


# ----------------------------------------------------------------------------- stage-by-stage forward + error bound
def graph_norm_bound(x, err, weight, mean_scale, eps=1e-5):
    """first-order bound on |GraphNorm(x + dx) - GraphNorm(x)| for |dx| <= err (elementwise):
    y = w o / s + b with o = x - ms * mean(x), s = sqrt(mean(o^2) + eps), so
    |d mean| <= mean(err) =: em, |d o| <= err + |ms| em =: eo, |d s| <= mean(|o| eo) / s,
    |d y| <= |w| / s * eo + |w| |o| / s^2 * |d s|."""
    x = np.asarray(x, dtype=np.float64)
    w, ms = np.abs(np.asarray(weight, dtype=np.float64))[None, :], np.asarray(mean_scale, dtype=np.float64)[None, :]
    o = x - x.mean(axis=0, keepdims=True) * ms
    s = np.sqrt((o * o).mean(axis=0, keepdims=True) + eps)
    eo = err + np.abs(ms) * err.mean(axis=0, keepdims=True)
    es = (np.abs(o) * eo).mean(axis=0, keepdims=True) / s
    return w / s * eo + w * np.abs(o) / (s * s) * es


def graph_norm_mag(x, weight, bias, mean_scale, eps=1e-5):
    """the magnitude GraphNorm's own arithmetic works on: |w| (|x| + |ms mean|) / s + |b|"""
    x = np.asarray(x, dtype=np.float64)
    w, ms = np.abs(np.asarray(weight, dtype=np.float64))[None, :], np.asarray(mean_scale, dtype=np.float64)[None, :]
    mean = x.mean(axis=0, keepdims=True)
    o = x - mean * ms
    s = np.sqrt((o * o).mean(axis=0, keepdims=True) + eps)
    return w * (np.abs(x) + np.abs(mean * ms)) / s + np.abs(np.asarray(bias, dtype=np.float64))[None, :]


def localwl_forward_stages(params, x_deg, edge1, pos, idx, ei2, tol=1e-5):
    """LocalWLNet.forward (TwoWL/model/model.py:68-84) one stage at a time, fp64, with a running first-order error
    bound for an implementation whose every stage is accurate to tol * (the magnitude that stage sums).

    Returns a list of stage records {name, kind, inputs..., out, mag, bound}:
      out   -- the stage's fp64 output,
      mag   -- the magnitude of the stage's own arithmetic (what `tol` is relative to when the stage is fed exact input),
      bound -- elementwise bound on the accumulated error of `out` when every stage so far errs by tol * its mag:
               bound_out = (stage's first-order sensitivity applied to bound_in) + tol * mag.
    The per-stage parity tests use (inputs, out, mag); the whole-forward test uses the last record's bound."""
    from .twowl_int import reverse
    edge2, edge2_r = reverse(ei2)
    p = params
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    stages = []

    def gn_stage(name, x, err, prefix, relu):
        w, b, ms = p[f"{prefix}.weight"], p[f"{prefix}.bias"], p[f"{prefix}.mean_scale"]
        out = graph_norm(x, w, b, ms)
        mag = graph_norm_mag(x, w, b, ms)
        bound = graph_norm_bound(x, err, w, ms) + tol * mag
        if relu:
            out = np.maximum(out, 0.0)                       # 1-Lipschitz: the bound carries over
        stages.append({"name": name, "kind": "graph_norm", "x": x, "norm": prefix, "relu": relu, "out": out, "mag": mag,
                       "bound": bound})
        return out, bound

    def conv_stage(name, x, err, prefix, edge):
        wt, b = f64(p[f"{prefix}.lin.weight"]).T, f64(p[f"{prefix}.bias"])
        n = x.shape[0]
        ei, w = gcn_norm(edge, n)
        h, mag_h = x @ wt, np.abs(x) @ np.abs(wt)
        err_h = err @ np.abs(wt) + tol * mag_h

        def aggregate(v):
            o = np.zeros((n, v.shape[1]))
            np.add.at(o, ei[1], v[ei[0]] * w[:, None])      # w >= 0
            return o
        out = aggregate(h) + b[None, :]
        mag = aggregate(mag_h) + np.abs(b)[None, :]
        bound = aggregate(err_h) + tol * (aggregate(np.abs(h)) + np.abs(b)[None, :])
        stages.append({"name": name, "kind": "gcn_conv", "x": x, "conv": prefix, "edge": edge, "out": out, "mag": mag,
                       "bound": bound})
        return out, bound

    x0 = f64(p["emb.weight"])[np.asarray(x_deg, dtype=np.int64)]
    x, err = gn_stage("emb+norm", x0, np.zeros_like(x0), "emb_norm", False)
    stages[-1]["kind"], stages[-1]["x_deg"] = "embedding_graph_norm", np.asarray(x_deg, dtype=np.int64)
    for l in range(p["depth1"]):
        y, ey = conv_stage(f"conv1s.{l}.conv", x, err, f"conv1s.{l}", edge1)
        x, err = gn_stage(f"conv1s.{l}.norm", y, ey, f"conv1s.{l}.norm", bool(p[f"conv1s.{l}.act"]))
    pos = np.asarray(pos, dtype=np.int64)
    a, b_ = x[pos[:, 0]], x[pos[:, 1]]
    ea, eb = err[pos[:, 0]], err[pos[:, 1]]
    h = a * b_
    err = np.abs(a) * eb + np.abs(b_) * ea + ea * eb + tol * np.abs(h)
    stages.append({"name": "pair", "kind": "pair", "x": x, "pos": pos, "out": h, "mag": np.abs(h), "bound": err})
    x = h
    for l in range(p["depth2"]):
        outs, ys = [], []
        for name, e in (("conv2s", edge2), ("conv2s_r", edge2_r)):
            y, ey = conv_stage(f"{name}.{l}.conv", x, err, f"{name}.{l}", e)
            ys.append(y)
            outs.append(gn_stage(f"{name}.{l}.norm", y, ey, f"{name}.{l}.norm", True))
        x = outs[0][0] + outs[1][0]
        err = outs[0][1] + outs[1][1] + tol * (np.abs(outs[0][0]) + np.abs(outs[1][0]))
        # "a" + GraphNorm_r(y_r): an implementation may fuse the sum into the second normalisation (y_r, norm_r given)
        stages.append({"name": f"sum.{l}", "kind": "sum", "a": outs[0][0], "b": outs[1][0], "y_r": ys[1],
                       "norm_r": f"conv2s_r.{l}.norm", "out": x,
                       "mag": np.abs(outs[0][0]) + stages[-1]["mag"], "bound": err})
    idx = np.asarray(idx, dtype=np.int64)
    xx, ee = x[idx], err[idx]
    z = xx[0::2] * xx[1::2]
    ez = np.abs(xx[0::2]) * ee[1::2] + np.abs(xx[1::2]) * ee[0::2] + ee[0::2] * ee[1::2] + tol * np.abs(z)
    wt, b = f64(p["pred.weight"]).T, f64(p["pred.bias"])
    out = z @ wt + b[None, :]
    mag = np.abs(z) @ np.abs(wt) + np.abs(b)[None, :]
    stages.append({"name": "readout", "kind": "readout", "x": x, "idx": idx, "out": out, "mag": mag,
                   "bound": ez @ np.abs(wt) + tol * mag})
    return stages


def seal_head_mag(z, key, hw):
    """the magnitude chain of seal_head: the same four linear levels on |inputs| and |weights| (ReLU of a
    non-negative value is the identity), i.e. sum |w x| at the pre-softmax logits.  A head whose every level is
    accurate to tol * (its magnitude) is within 4 * tol * seal_head_mag of the fp64 logits: the level's own error
    plus the previous levels' errors carried through |weights| are each <= tol * (this chain)."""
    az = {k: np.abs(np.asarray(v, dtype=np.float64)) for k, v in hw.items()}
    pre_y, _ = seal_head(np.abs(np.asarray(z, dtype=np.float64)), key, az)
    return pre_y


# ----------------------------------------------------------------------------- TwoWL forward without the tuple graph
def tuple_structure(n_node, pos_edge, pred_edge, blocked=None):
    """what the tuple graph of TwoWL/utils.py:36-45 (get_ei2), after blockei2 (:48-50) and reverse (:71-78), looks like
    when pos_edge is in double()'s layout (:81-90: columns 2k / 2k + 1 are the two directions of one edge) and the
    blocked tuple ids come in such pairs too (TwoWL/model/train.py:26: double(idx1, for_index=True)):

      orientation a (edge2   = (flip(a), b)):  the in-neighbours of tuple b are the MEMBER tuples (graph tuples that are
                                               not blocked) that START at node_a(b) = start(b);
      orientation r (edge2_r = (a, flip(b))):  the in-neighbours of b are the members that END at node_r(b) = start(b^1)
                                               (= end(b) when the candidates are doubled as well);
      add_remaining_self_loops gives b one more in-edge (itself) exactly when b is not a member;
      every member in a group has the same degree cnt(node) = the group's size.

    Returns dict(start, end, node_a, node_r, member, cnt, ok); ok = the layout conditions hold."""
    pos_edge = np.asarray(pos_edge, dtype=np.int64)
    edge = np.concatenate([pos_edge, np.asarray(pred_edge, dtype=np.int64)], axis=1)
    E1, P = pos_edge.shape[1], edge.shape[1]
    member = np.zeros(P, dtype=bool)
    member[:E1] = True
    if blocked is not None:
        member[np.asarray(blocked, dtype=np.int64)] = False
    ok = E1 % 2 == 0 and P % 2 == 0
    if ok:
        ok = bool(np.array_equal(pos_edge[0, 0::2], pos_edge[1, 1::2]) and np.array_equal(pos_edge[1, 0::2], pos_edge[0, 1::2])
                  and np.array_equal(member[0::2], member[1::2]))
    start, end = edge[0], edge[1]
    partner = np.arange(P) ^ 1 if P % 2 == 0 else np.arange(P)
    cnt = np.bincount(start[member], minlength=n_node).astype(np.int64)
    return {"start": start, "end": end, "node_a": start, "node_r": start[partner], "member": member, "cnt": cnt, "ok": ok}


def localwl_forward_collapsed(params, x_deg, edge1, pos_edge, pred_edge, idx, n_node, blocked=None):
    """the same function as localwl_forward(params, x_deg, edge1, cat(pos_edge, pred_edge).T, idx,
    blockei2(get_ei2(n_node, pos_edge, pred_edge), blocked)) -- TwoWL/model/model.py:68-84 -- computed WITHOUT the
    tuple graph: every per-node block of get_ei2 is a complete bipartite graph whose sources share one degree, so a
    2-WL layer is a per-node sum over the member tuples plus a broadcast, and because member tuples' features are
    themselves node-level (x[u] * x[v] at layer 0; F_a[start] + F_r[end] after a layer) those sums are products with
    the member adjacency A.  O((N + Q) C^2 + E1 C) instead of O(E2 C), Q = non-member tuples.  fp64."""
    p = params
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    s = tuple_structure(n_node, pos_edge, pred_edge, blocked)
    if not s["ok"]:
        raise ValueError("pos_edge / blocked are not in double() layout")
    start, end, member, cnt = s["start"], s["end"], s["member"], s["cnt"].astype(np.float64)
    P = start.shape[0]
    x = input_features(p, x_deg)
    for l in range(p["depth1"]):
        x = gcn_conv(x, edge1, p[f"conv1s.{l}.lin.weight"], p[f"conv1s.{l}.bias"])
        x = graph_norm(x, p[f"conv1s.{l}.norm.weight"], p[f"conv1s.{l}.norm.bias"], p[f"conv1s.{l}.norm.mean_scale"])
        if p[f"conv1s.{l}.act"]:
            x = np.maximum(x, 0.0)
    ms_, me_ = start[member], end[member]

    class _Adj:                                                # the member adjacency (symmetric by the layout conditions)
        def __matmul__(self, v):
            out = np.zeros((n_node, v.shape[1]))
            np.add.at(out, ms_, v[me_])
            return out
    A = _Adj()
    q = np.nonzero(~member)[0]
    na, nr = s["node_a"][q], s["node_r"][q]
    hq = x[start[q]] * x[end[q]]
    Fa = Fr = None
    safe = np.maximum(cnt, 1.0)
    for l in range(p["depth2"]):
        if l == 0:
            sx_a = sx_r = x * (A @ x)
        else:
            sx_a = cnt[:, None] * Fa + A @ Fr
            sx_r = A @ Fa + cnt[:, None] * Fr
        outs = []
        for name, sx, nq in (("conv2s", sx_a, na), ("conv2s_r", sx_r, nr)):
            wt, b = f64(p[f"{name}.{l}.lin.weight"]).T, f64(p[f"{name}.{l}.bias"])
            T = sx @ wt
            M = T / safe[:, None] + b[None, :]                                  # every member row of node i
            c = cnt[nq]
            yq = T[nq] * np.where(c > 0, 1.0 / np.sqrt(np.maximum(c, 1.0) * (c + 1.0)), 0.0)[:, None] \
                + (hq @ wt) / (c + 1.0)[:, None] + b[None, :]
            w, gb, ms = (f64(p[f"{name}.{l}.norm.{k}"]) for k in ("weight", "bias", "mean_scale"))
            s1 = (cnt[:, None] * M).sum(0) + yq.sum(0)
            mean = s1 / P
            o_n, o_q = M - mean * ms, yq - mean * ms
            var = ((cnt[:, None] * o_n * o_n).sum(0) + (o_q * o_q).sum(0)) / P
            sc = w / np.sqrt(var + 1e-5)
            outs.append((np.maximum(o_n * sc + gb, 0.0), np.maximum(o_q * sc + gb, 0.0)))
        Fa, Fr = outs[0][0], outs[1][0]
        hq = outs[0][1] + outs[1][1]
    idx = np.asarray(idx, dtype=np.int64)
    qpos = np.full(P, -1, dtype=np.int64)
    qpos[q] = np.arange(q.size)
    rows = np.where(member[idx][:, None], Fa[start[idx]] + Fr[end[idx]], hq[np.maximum(qpos[idx], 0)])
    z = rows[0::2] * rows[1::2]
    return z @ f64(p["pred.weight"]).T + f64(p["pred.bias"])[None, :]
