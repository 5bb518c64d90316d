"""Host-side mirror of the reference's SEAL hot path.

    build_train_csr / extract_subgraphs / link2subgraph   <-> SEAL/config/subgraph.py:11-141
    node labels (DRNL)                                     <-> SEAL/utils/utils.py:35-54
    gnn_inputs                                             <-> SEAL/config/data.py:94-140
    graph_conv_stack                                       <-> SEAL/model/gnn.py:46-63

The reference runs one candidate link at a time (networkx + one TF sess.run per subgraph with
dense [n,n] operands).  Here a whole batch of candidates is one launch per stage: subgraphs are
extracted and labelled on the device, stacked into ONE block-diagonal CSR, and each of the four
graph-conv layers is one MFMA linear + one fused SpMM (row scale D^-1 and tanh in the epilogue).
"""
import numpy as np
import torch

from . import _lib
from . import ops as _ops

MAXN = 128


class TrainGraph:
    """symmetric 0/1 adjacency of the train-positive graph as a CSR with ascending, unique
    columns per row (self loops kept) -- the `A` / `G` of SEAL/config/subgraph.py:27-31,70-72."""

    def __init__(self, rowptr, col, n_nodes):
        self.rowptr, self.col, self.n_nodes = rowptr, col, int(n_nodes)


def build_train_csr(positive, n_nodes, test_ratio=0.1, device="cuda"):
    positive = torch.as_tensor(np.asarray(positive), dtype=torch.int64, device=device)
    test_size = int(len(positive) * test_ratio)
    train_pos = positive[: len(positive) - test_size] if test_size > 0 else positive
    u, v = train_pos[:, 0], train_pos[:, 1]
    keys = torch.unique(torch.cat([u * n_nodes + v, v * n_nodes + u]))       # sorted, deduplicated
    rows = torch.div(keys, n_nodes, rounding_mode="floor")
    col = (keys - rows * n_nodes).to(torch.int32)
    rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_nodes), 0)
    return TrainGraph(rowptr, col.contiguous(), n_nodes)


class SubgraphBatch:
    def __init__(self, pairs, nodes, n, tags, adj_bits, capped):
        self.pairs, self.nodes, self.n, self.tags, self.adj_bits, self.capped = pairs, nodes, n, tags, adj_bits, capped
        self.B = pairs.shape[0]

    def node_off(self):
        off = torch.zeros(self.B + 1, dtype=torch.int64, device=self.n.device)
        off[1:] = torch.cumsum(self.n.to(torch.int64), 0)
        return off


def extract_subgraphs(pairs, graph, hop, max_neighbors=100, replay_nodes=None, replay_n=None):
    """Batched extract_subgraph + node_labeling.  pairs int64 [B,2].  Pairs for which the
    reference's RNG-dependent truncation would fire come back with capped=1 and n=0; pass their
    node lists through replay_nodes/replay_n to label them in a given order."""
    lib = _lib.load()
    dev = graph.rowptr.device
    pairs = torch.as_tensor(pairs, dtype=torch.int64, device=dev).contiguous()
    B = pairs.shape[0]
    replay = replay_nodes is not None
    if replay:
        nodes = torch.full((B, MAXN), -1, dtype=torch.int32, device=dev)
        w = min(replay_nodes.shape[1], MAXN)
        nodes[:, :w] = torch.as_tensor(replay_nodes, dtype=torch.int32, device=dev)[:, :w]
        n = torch.as_tensor(replay_n, dtype=torch.int32, device=dev).contiguous()
    else:
        nodes = torch.empty((B, MAXN), dtype=torch.int32, device=dev)
        n = torch.empty(B, dtype=torch.int32, device=dev)
    tags = torch.empty((B, MAXN), dtype=torch.int32, device=dev)
    adj_bits = torch.empty((B, MAXN, MAXN // 32), dtype=torch.int32, device=dev)
    capped = torch.empty(B, dtype=torch.uint8, device=dev)
    _lib.check(lib.lpg_seal_subgraphs(_ops._ptr(graph.rowptr), _ops._ptr(graph.col), graph.n_nodes, _ops._ptr(pairs),
                                      B, int(hop), int(max_neighbors), int(replay), _ops._ptr(nodes), _ops._ptr(n),
                                      _ops._ptr(tags), _ops._ptr(adj_bits), _ops._ptr(capped),
                                      _ops._ptr(_ops.err_flag(dev)), _ops._stream()), "seal_subgraphs")
    _ops.maybe_check(dev, "candidate pair / node id")
    return SubgraphBatch(pairs, nodes, n, tags, adj_bits, capped)


def _host_capped_nodes(pair, rowptr, col, max_neighbors, rng):
    """the reference's truncation branch (SEAL/config/subgraph.py:118-135, hop = 1) with an explicit
    numpy RandomState; only used for the few capped pairs."""
    u, v = int(pair[0]), int(pair[1])
    sub = {u, v}
    order = [u, v]
    rng.shuffle(order)
    for node in order:
        nb = list(col[rowptr[node]:rowptr[node + 1]])
        if len(sub) + len(nb) < max_neighbors:
            sub |= set(int(x) for x in nb)
        else:
            rng.shuffle(nb)
            sub |= set(int(x) for x in nb[: max_neighbors - len(sub)])
            break
    sub.discard(u)
    sub.discard(v)
    return [u, v] + sorted(sub)


def link2subgraph(positive, negative, nodes_size, test_ratio, hop, network_type=0, max_neighbors=100, seed=0,
                  device="cuda"):
    """SEAL/config/subgraph.py:11-102 for hop in {0, 1}: negatives first, then positives (:74).
    Returns (batch, labels, tags_size, tag_map, graph).  Tags in `batch.tags` are RAW; tag_map is
    the global compaction (:84-101) applied when features are built."""
    if network_type != 0:
        raise NotImplementedError("the reference is run undirected (network_type=0)")
    if hop == "auto":
        raise NotImplementedError("hop='auto' (CN/AA AUC heuristic, :44-63) is host-side model selection; "
                                  "fb-pages-food selects hop=1 (tests/golden/seal_int.npz: auto_hop_line)")
    graph = build_train_csr(positive, nodes_size, test_ratio, device)
    pairs = np.concatenate([np.asarray(negative), np.asarray(positive)]).astype(np.int64)
    batch = extract_subgraphs(pairs, graph, hop, max_neighbors)
    capped = batch.capped.bool()
    if bool(capped.any()):
        idx = torch.nonzero(capped).reshape(-1).cpu().numpy()
        rp, cl = graph.rowptr.cpu().numpy(), graph.col.cpu().numpy()
        rng = np.random.RandomState(seed)
        lists = [_host_capped_nodes(pairs[i], rp, cl, max_neighbors, rng) for i in idx]
        rn = np.full((len(idx), MAXN), -1, dtype=np.int32)
        for k, l in enumerate(lists):
            rn[k, : len(l)] = l
        sub = extract_subgraphs(pairs[idx], graph, hop, max_neighbors, replay_nodes=rn,
                                replay_n=np.asarray([len(l) for l in lists], dtype=np.int32))
        ii = torch.as_tensor(idx, device=batch.n.device)
        batch.nodes[ii], batch.n[ii], batch.tags[ii], batch.adj_bits[ii] = sub.nodes, sub.n, sub.tags, sub.adj_bits
    labels = torch.cat([torch.zeros(len(negative), dtype=torch.uint8), torch.ones(len(positive), dtype=torch.uint8)])
    tag_map, tags_size = compact_tag_map(batch)
    return batch, labels.reshape(-1, 1), tags_size, tag_map, graph


def compact_tag_map(batch):
    """Global tag compaction (SEAL/config/subgraph.py:84-101).  The reference maps a tag to its
    position in list(set(all tags)) -- CPython set order, which is ascending for the small
    non-negative tags DRNL produces; here: ascending order.  Identity when tags == range(T)."""
    valid = batch.tags[batch.tags >= 0]
    uniq = torch.unique(valid)
    T = int(uniq.numel())
    tag_map = torch.full((int(uniq.max().item()) + 1 if T else 1,), -1, dtype=torch.int32, device=batch.tags.device)
    tag_map[uniq.long()] = torch.arange(T, dtype=torch.int32, device=batch.tags.device)
    return tag_map, T


def gnn_inputs(batch, tag_map, tags_size, emb=None):
    """SEAL/config/data.py:94-140 batched: block-diagonal A~ = A + I (CSR, values 1 / 2),
    D^-1 = 1/rowsum(A~) per vertex, X = [onehot(tag) || emb[node]].  Returns (CSRGraph, d_inv, X)."""
    lib = _lib.load()
    dev = batch.n.device
    off = batch.node_off()
    total = int(off[-1].item())
    nbytes = _lib.check(lib.lpg_seal_block_csr_bytes(total), "seal_block_csr_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rowptr = torch.empty(total + 1, dtype=torch.int64, device=dev)
    d_inv = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
    P, S = _ops._ptr, _ops._stream
    _lib.check(lib.lpg_seal_block_csr(P(batch.n), P(off), P(batch.adj_bits), batch.B, P(rowptr), 0, 0, P(d_inv), 0,
                                      P(ws), nbytes, S()), "seal_block_csr(count)")
    nnz = int(rowptr[-1].item())
    col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
    _lib.check(lib.lpg_seal_block_csr(P(batch.n), P(off), P(batch.adj_bits), batch.B, P(rowptr), P(col), P(val),
                                      P(d_inv), nnz, P(ws), nbytes, S()), "seal_block_csr(fill)")
    g = _ops.CSRGraph(rowptr, col[:nnz] if nnz else col, val[:nnz] if nnz else val, total, total, nnz)
    g._build_plan()
    g.n_empty = 0                       # A~ = A + I: every row holds its diagonal
    emb_dim = 0 if emb is None else emb.shape[1]
    emb = None if emb is None else _ops._c(emb, torch.float32)
    x = torch.empty((total, tags_size + emb_dim), dtype=torch.float32, device=dev)
    _lib.check(lib.lpg_seal_features_f32(P(batch.nodes), P(batch.n), P(off), P(batch.tags), P(tag_map),
                                         tag_map.numel(), tags_size, P(emb), emb_dim, batch.B, P(x), x.stride(0),
                                         P(_ops.err_flag(dev)), S()), "seal_features")
    _ops.maybe_check(dev, "seal feature index")
    return g, d_inv[:total], x


def graph_conv_stack(g, d_inv, x, weights):
    """SEAL/model/gnn.py:46-63: Z_l = tanh(D^-1 (A~ (Z_{l-1} W_l))), l = 1..4.
    Returns (concat(Z1, Z2, Z3) [total, 96], Z4[:, 0] -- the SortPooling key, :82)."""
    z, zs = x, []
    for w in weights:
        h = _ops.linear(z, w, w_is_km=True)                       # K5 on the fp32 matrix cores
        z = g.spmm(h, row_scale=d_inv, act="tanh")                # K10: one fused kernel per layer
        zs.append(z)
    return torch.cat(zs[:3], dim=1), zs[3][:, 0]
