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
import os

import numpy as np
import torch

from . import _lib
from . import ops as _ops

MAXN = 128


class TrainGraph:
    """symmetric 0/1 adjacency of the train-positive graph as a CSR with ascending, unique
    columns per row (self loops kept) -- the `A` / `G` of SEAL/config/subgraph.py:27-31,70-72."""

    def __init__(self, rowptr, col, n_nodes, directed=False):
        self.rowptr, self.col, self.n_nodes = rowptr, col, int(n_nodes)
        self.directed = bool(directed)               # network_type = 1: rows list successors only (nx.DiGraph)
        self.hub_index = self.hub_bits = None        # hub bit matrix (build_hub_matrix): hub x hub adjacency in one load

    # neighbour lists longer than hub_degree are not streamed by the extraction; their nodes get rows in the hub bit matrix.
    # build_hub_matrix takes the smallest candidate that leaves at most MAX_HUBS hubs (512 MB of bits at most).  100 K pairs
    # on the collab-shaped graph (profiles/r02/seal_hub_degree.log): 256 (988 hubs) 0.46 ms, 128 (4048) 0.32, 64 (5574)
    # 0.32, 32 (12.7 K) 0.34 -- the graph has ~3000 nodes of degree 150-200 whose lists were streamed in every heavy pair.
    # ppa-shaped (mean degree 105; profiles/r02/seal_ppa_extract.log), 50 K pairs: no matrix (round 1's limit of 4096
    # hubs) 2.94 ms; 128 (61.7 K hubs, 476 MB) 1.11; 256 (35 K, 153 MB) 1.06; 1024 (6 K) 2.41
    HUB_DEGREES = (64, 128, 256, 512, 1024, 2048, 4096)
    MAX_HUBS = 1 << 16
    hub_degree = 256                                 # (until build_hub_matrix chose)

    def build_hub_matrix(self):
        """hub_index[node] = row in the bit matrix (-1: not a hub), hub_bits[row, c >> 5] bit (c & 31) = hubs row and c
        are adjacent; hubs = nodes with more than hub_degree neighbours.  Skipped for graphs with more than
        MAX_HUBS of them or fewer than 128 K stored edges (the kernel then binary-searches hub x hub entries as before)."""
        dev = self.rowptr.device
        if self.directed or self.col.numel() < (1 << 17):   # directed: no symmetry for the kernel to lean on; small graphs
            return self                                     # (fb-pages-food): no host sync, the few searches are cheap
        deg = self.rowptr[1:] - self.rowptr[:-1]
        counts = [int(v) for v in torch.stack([(deg > d).sum() for d in self.HUB_DEGREES]).tolist()]   # one host read
        self.hub_degree = next((d for d, cnt in zip(self.HUB_DEGREES, counts) if cnt <= self.MAX_HUBS), self.HUB_DEGREES[-1])
        hubs = torch.nonzero(deg > self.hub_degree).flatten()
        H = int(hubs.numel())
        if H == 0 or H > self.MAX_HUBS:
            return self
        idx = torch.full((self.n_nodes,), -1, dtype=torch.int32, device=dev)
        idx[hubs] = torch.arange(H, dtype=torch.int32, device=dev)
        W = (H + 31) // 32
        lens = deg[hubs]
        rows = torch.repeat_interleave(torch.arange(H, device=dev), lens)
        starts = torch.repeat_interleave(self.rowptr[hubs], lens)
        offs = torch.arange(int(lens.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
        cols = idx[self.col[starts + offs].long()].long()
        keep = cols >= 0
        rows, cols = rows[keep], cols[keep]
        # one bit per hub x hub entry, added into int64 words (H x W x 8 bytes; a dense H x H boolean image took
        # 1.2 GB at the ppa-shaped graph's 35 K hubs).  A repeated entry of a sorted list is adjacent: dropped first.
        if rows.numel() > 1:
            first = torch.ones_like(rows, dtype=torch.bool)
            first[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
            rows, cols = rows[first], cols[first]
        words = torch.zeros(H * W, dtype=torch.int64, device=dev)
        words.scatter_add_(0, rows * W + (cols >> 5), torch.ones_like(cols) << (cols & 31))
        self.hub_index = idx
        self.hub_bits = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).view(H, W).contiguous()   # uint32 bit patterns
        self.hub_words = W
        return self


def build_train_csr(positive, n_nodes, test_ratio=0.1, device="cuda", network_type=0):
    """positive: [P, 2] node pairs (numpy array or tensor, host or device); the last test_ratio of them are held out.
    network_type: 0 undirected (both orientations, SEAL/config/subgraph.py:29-31), 1 directed (row u lists v only)."""
    directed = int(network_type) != 0
    if not torch.is_tensor(positive):
        positive = torch.from_numpy(np.ascontiguousarray(np.asarray(positive), dtype=np.int64))
    positive = positive.to(device=device, dtype=torch.int64)
    test_size = int(len(positive) * test_ratio)
    train_pos = positive[: len(positive) - test_size] if test_size > 0 else positive
    if positive.is_cuda and n_nodes <= 8192:
        # small graphs (the reference's own: 620 nodes): a bit matrix and three launches, no host sync
        lib = _lib.load()
        train_pos = train_pos.contiguous()
        P = train_pos.shape[0]
        words = (n_nodes + 31) // 32
        bits = torch.empty(n_nodes * words, dtype=torch.int32, device=positive.device)
        rowptr = torch.empty(n_nodes + 1, dtype=torch.int64, device=positive.device)
        col = torch.empty(max(2 * P, 1), dtype=torch.int32, device=positive.device)
        _lib.check(lib.lpg_seal_train_csr_small(_ops._ptr(train_pos), P, n_nodes, int(directed), _ops._ptr(bits), bits.numel() * 4,
                                                _ops._ptr(rowptr), _ops._ptr(col), col.numel(),
                                                _ops._ptr(_ops.err_flag(positive.device)), _ops._stream()),
                   "seal_train_csr_small")
        return TrainGraph(rowptr, col, n_nodes, directed)      # (col's tail past rowptr[n] is unused capacity; hub matrix: n too small)
    if positive.is_cuda:
        # any size: sorted, deduplicated 64-bit keys (row * n + col) -> col / rowptr, no host sync (csrc/seal_graph.hip)
        lib = _lib.load()
        train_pos = train_pos.contiguous()
        P = train_pos.shape[0]
        dev = positive.device
        rowptr = torch.empty(n_nodes + 1, dtype=torch.int64, device=dev)
        col = torch.empty(max(2 * P, 1), dtype=torch.int32, device=dev)
        nbytes = _lib.check(lib.lpg_seal_train_csr_bytes(P, n_nodes), "seal_train_csr_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.lpg_seal_train_csr(_ops._ptr(train_pos), P, n_nodes, int(directed), _ops._ptr(rowptr), _ops._ptr(col), col.numel(),
                                          _ops._ptr(ws), nbytes, _ops._ptr(_ops.err_flag(dev)), _ops._stream()),
                   "seal_train_csr")
        return TrainGraph(rowptr, col, n_nodes, directed).build_hub_matrix()
    # CPU tensors only (tests/test_seal_host.py checks the hub matrix's host logic on them): the same CSR through torch;
    # nothing downstream runs without a ROCm device (extract_subgraphs raises on CPU tensors)
    u, v = train_pos[:, 0], train_pos[:, 1]
    keys = torch.unique(u * n_nodes + v if directed else torch.cat([u * n_nodes + v, v * n_nodes + u]))   # sorted, deduplicated
    rows = torch.div(keys, n_nodes, rounding_mode="floor")
    col = (keys - rows * n_nodes).to(torch.int32)
    rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_nodes), 0)
    return TrainGraph(rowptr, col.contiguous(), n_nodes, directed).build_hub_matrix()


class SubgraphBatch:
    def __init__(self, pairs, nodes, n, tags, adj_bits, capped, tag_present=None):
        self.pairs, self.nodes, self.n, self.tags, self.adj_bits, self.capped = pairs, nodes, n, tags, adj_bits, capped
        self.tag_present = tag_present            # uint8 [TAG_BOUND + 1], byte [tag] set by the extraction kernel
        self.B = pairs.shape[0]
        self._off = None
        self._total = None
        self._tag_map = None
        self._head = None
        self._n_tags = None
        self._max_tag = -1
        self._first = None

    def _index(self):
        """node offsets, tag map and the head {total, error flag, distinct tags} in ONE launch (lpg_seal_batch_index)."""
        if self._off is None:
            lib = _lib.load()
            dev = self.n.device
            P = _ops._ptr
            present = self.tag_present
            K = present.numel() if present is not None else 0
            self._off = torch.empty(self.B + 1, dtype=torch.int64, device=dev)
            self._tag_map = torch.empty(K, dtype=torch.int32, device=dev) if K else None
            self._head = torch.empty(4, dtype=torch.int64, device=dev)
            n = self.n if self.n.is_contiguous() else self.n.contiguous()
            nbytes = _lib.check(lib.lpg_seal_batch_index_bytes(self.B), "seal_batch_index_bytes")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
            # small batches are launch-bound: the conv stack's 128-row windows ride along (sizes <= 128 => at most B + 1
            # of them; one workgroup searches them, so not for big batches -- 100 K pairs: +0.1 ms, measured)
            self._first = torch.empty(self.B + 2, dtype=torch.int32, device=dev) if self.B <= 8192 else None
            _lib.check(lib.lpg_seal_batch_index(P(n), self.B, P(self._off), P(present), K, P(self._tag_map),
                                                P(_ops.err_flag(dev)), P(self._head), P(self._first), CONV_WINDOW,
                                                P(ws), nbytes, _ops._stream()), "seal_batch_index")
            self._total = self._n_tags = None

    def _read_head(self):
        """the batch's one host read: vertex total, the extraction's index check, number of distinct tags."""
        if self._total is None:
            self._index()
            total, flag, n_tags, max_tag = self._head.tolist()
            self._total, self._n_tags, self._max_tag = int(total), int(n_tags), int(max_tag)
            if flag and _ops._CHECK:
                _ops.err_flag(self._head.device).zero_()
                raise IndexError(f"lpgnn_amd: candidate pair / node id out of range (flag={int(flag)})")

    def total(self):
        """number of vertices in the batch (one host sync, cached).  The extraction's index check rides on the same
        transfer: the error flag comes back with the total instead of in a sync of its own."""
        self._read_head()
        return self._total

    def node_off(self):
        """exclusive prefix sum of the subgraph sizes, int64 [B+1]; computed once per batch."""
        self._index()
        return self._off

    def tag_map(self):
        """(tag_map int32 [TAG_BOUND + 1], T): ascending present tags -> 0..T-1, -1 elsewhere; needs tag_present."""
        self._read_head()
        return self._tag_map, self._n_tags

    def invalidate(self):
        """the sizes / tags were edited in place (capped pairs replayed): recompute the index on next use"""
        self._off = self._total = self._tag_map = self._head = self._n_tags = self._first = None


CAP_POLICY = {"report": 0, "device": 1}
CONV_WINDOW = 128               # rows per workgroup window of the dense conv stack (csrc/seal_conv.hip)
DENSE_STACK = True              # False: graph_conv_stack always takes the generic block-CSR path (A/B, tests)


def extract_subgraphs(pairs, graph, hop, max_neighbors=100, replay_nodes=None, replay_n=None, cap_policy="device",
                      seed=0):
    """Batched extract_subgraph + node_labeling.  pairs int64 [B,2].  `capped` marks the pairs for
    which the reference's RNG-dependent truncation (np.random.shuffle, subgraph.py:122,128) fires:
      cap_policy="device": the kernel follows the same branch with a hash of (seed, u, v) in
                           place of the RNG -- deterministic, no host round trip;
      cap_policy="report": those pairs come back with n = 0; pass node lists through
                           replay_nodes / replay_n to label them in a given order (tests replay the
                           reference's own lists this way)."""
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
    present = torch.zeros(_TAG_BOUND + 1, dtype=torch.uint8, device=dev)
    _lib.check(lib.lpg_seal_subgraphs(_ops._ptr(graph.rowptr), _ops._ptr(graph.col), graph.n_nodes, _ops._ptr(pairs),
                                      B, int(hop), int(max_neighbors), int(replay), CAP_POLICY[cap_policy],
                                      int(seed) & 0xFFFFFFFFFFFFFFFF, _ops._ptr(nodes), _ops._ptr(n),
                                      _ops._ptr(tags), _ops._ptr(adj_bits), _ops._ptr(capped), _ops._ptr(present),
                                      _ops._ptr(graph.hub_index), _ops._ptr(graph.hub_bits),
                                      getattr(graph, "hub_words", 0) if graph.hub_bits is not None else 0,
                                      int(getattr(graph, "hub_degree", 256)),
                                      int(getattr(graph, "directed", False)), _ops._ptr(_ops.err_flag(dev)), _ops._stream()),
               "seal_subgraphs")
    # no sync of its own: the flag is read together with the batch's vertex total (SubgraphBatch.total(), which every
    # consumer needs first) or by the next checking op; out-of-range pairs were skipped, never dereferenced
    return SubgraphBatch(pairs, nodes, n, tags, adj_bits, capped, present)


def _insertion_order_neighbours(train_pos, wanted, directed):
    """{node: its neighbour list in networkx's adjacency-dict order} for the nodes in `wanted`: G.add_edges_from(train_pos)
    (SEAL/config/subgraph.py:69) appends b to adj[a] (and a to adj[b] for nx.Graph) the FIRST time the pair shows up,
    so list(nx.neighbors(G, x)) is x's partners in order of first appearance over the rows.  train_pos: [P, 2] numpy."""
    a, b = train_pos[:, 0].astype(np.int64), train_pos[:, 1].astype(np.int64)
    if directed:
        src, dst = a, b
    else:                                          # row i contributes (a -> b) then (b -> a)
        src = np.stack([a, b], 1).reshape(-1)
        dst = np.stack([b, a], 1).reshape(-1)
    pick = np.isin(src, np.fromiter(wanted, dtype=np.int64, count=len(wanted)))
    src, dst = src[pick], dst[pick]
    out = {int(x): [] for x in wanted}
    seen = set()
    for x, y in zip(src.tolist(), dst.tolist()):
        if (x, y) not in seen:
            seen.add((x, y))
            out[x].append(y)
    return out


def host_cap_node_lists(pairs, flagged, train_pos, max_neighbors, seed, directed=False):
    """The reference's node sets for the pairs its 100-node truncation can touch, SEED FOR SEED (SEAL/config/subgraph.py:
    118-135 at hop = 1 under np.random.seed(seed)): every pair -- negatives first, then positives, the order link2subgraph
    walks them in (:74-82) -- draws np.random.shuffle([u, v]) (:122); a pair whose running size reaches the cap at a
    visit shuffles that node's neighbour list, taken in networkx insertion order, and keeps its head (:127-130).
    `flagged`: the pairs the device reported as capped in EITHER visiting order (a superset of those that truncate in the
    drawn order; only they can consume more of the stream).  Returns {pair index: [u, v] + sorted(other vertices)}."""
    rng = np.random.RandomState(seed)
    flagged = set(int(i) for i in flagged)
    wanted = set()
    for i in flagged:
        wanted.update((int(pairs[i][0]), int(pairs[i][1])))
    nbrs = _insertion_order_neighbours(np.asarray(train_pos), wanted, directed) if wanted else {}
    out = {}
    for i in range(len(pairs)):
        u, v = int(pairs[i][0]), int(pairs[i][1])
        order = [u, v]
        rng.shuffle(order)                         # drawn for EVERY pair
        if i not in flagged:
            continue
        sub = {u, v}
        for node in order:
            nb = list(nbrs[node])
            if len(sub) + len(nb) < max_neighbors:
                sub |= set(nb)
            else:
                rng.shuffle(nb)
                sub |= set(nb[: max_neighbors - len(sub)])
                break
        sub.discard(u)
        sub.discard(v)
        out[i] = [u, v] + sorted(sub)
    return out


def _host_capped_nodes(pair, rowptr, col, max_neighbors, rng):     # (rowptr / col: successors for a directed graph)
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


from .metrics import rank_auc as _rank_auc  # noqa: E402  (public: lpgnn_amd.metrics.rank_auc)


def auto_hop(positive, negative, graph, test_ratio=0.1):
    """SEAL/config/subgraph.py:44-63 -- hop = 1 if AUC(common neighbours) > AUC('aa') else 2 on the
    held-out pairs.  Restated as the reference computes it: cn = A @ A in **uint8** (wraps mod 256),
    aa = A @ (A / log(rowsum A)) with nan/inf -> 0, i.e. aa[u,v] = cn[u,v] / log(deg v) (the
    reference divides COLUMN j by log(deg j))."""
    dev = graph.rowptr.device
    positive, negative = np.asarray(positive), np.asarray(negative)
    test_size = int(len(positive) * test_ratio)
    pairs = torch.as_tensor(np.concatenate([positive[-test_size:], negative[-test_size:]]), dtype=torch.int64, device=dev)
    y = torch.cat([torch.ones(test_size, device=dev), torch.zeros(test_size, device=dev)])
    n = graph.n_nodes
    deg = (graph.rowptr[1:] - graph.rowptr[:-1])
    col = graph.col[: int(graph.rowptr[-1])].long()                  # (the col buffer may have spare capacity)
    keys = torch.repeat_interleave(torch.arange(n, device=dev), deg) * n + col    # sorted (row, col) keys
    u, v = pairs[:, 0], pairs[:, 1]
    du = deg[u]
    pid = torch.repeat_interleave(torch.arange(pairs.shape[0], device=dev), du)
    off = torch.arange(int(du.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(du, 0) - du, du)
    w = col[graph.rowptr[u][pid] + off]                               # every neighbour w of u, per pair
    probe = w * n + v[pid]
    pos = torch.searchsorted(keys, probe).clamp(max=keys.numel() - 1)
    hit = keys[pos] == probe                                          # A[w, v] == 1
    cn = torch.zeros(pairs.shape[0], dtype=torch.int64, device=dev).index_add_(0, pid, hit.long())
    cn_u8 = (cn % 256).double()                                       # np.matmul on uint8 wraps
    logd = torch.log(deg[v].double())
    aa = cn.double() / logd                                           # float64 path: no wrap
    aa[torch.isnan(aa) | torch.isinf(aa)] = 0.0
    return 1 if _rank_auc(y, cn_u8) > _rank_auc(y, aa) else 2


def link2subgraph(positive, negative, nodes_size, test_ratio, hop, network_type=0, max_neighbors=100, seed=0,
                  device="cuda", cap_policy="device"):
    """SEAL/config/subgraph.py:11-102 for hop in {0, 1, "auto"}: negatives first, then positives (:74).
    Returns (batch, labels, tags_size, tag_map, graph).  Tags in `batch.tags` are RAW; tag_map is
    the global compaction (:84-101) applied when features are built.
    cap_policy: "device" (default: a counter-based hash of (seed, u, v) stands in for np.random.shuffle) or "host" --
    the pairs that can hit the 100-node cap are rebuilt on the host SEED FOR SEED with the reference's own stream
    (host_cap_node_lists: np.random.RandomState(seed), one shuffle([u, v]) per pair in the reference's pair order,
    neighbour lists in networkx insertion order), then replayed on the device: under the same seed the truncated
    subgraphs are the reference's own (tests/golden/seal_int.npz was generated under np.random.seed(2))."""
    graph = build_train_csr(positive, nodes_size, test_ratio, device, network_type)
    if hop == "auto":
        hop = auto_hop(positive, negative, graph, test_ratio)
        if hop != 1:
            raise NotImplementedError("the heuristic selected hop=2, which the reference itself cannot run "
                                      "(SEAL/config/subgraph.py:122,131 shuffles a set)")
    pairs = np.concatenate([np.asarray(negative), np.asarray(positive)]).astype(np.int64)
    if cap_policy == "device":
        batch = extract_subgraphs(pairs, graph, hop, max_neighbors, cap_policy="device", seed=seed)
    else:
        batch = extract_subgraphs(pairs, graph, hop, max_neighbors, cap_policy="report")
        capped = batch.capped.bool()
        if bool(capped.any()):
            idx = torch.nonzero(capped).reshape(-1).cpu().numpy()
            pos_h = positive.cpu().numpy() if torch.is_tensor(positive) else np.asarray(positive)
            test_size = int(len(pos_h) * test_ratio)
            train_pos = pos_h[: len(pos_h) - test_size] if test_size > 0 else pos_h          # as build_train_csr takes it
            by_pair = host_cap_node_lists(pairs, idx, train_pos, max_neighbors, seed, directed=int(network_type) != 0)
            lists = [by_pair[int(i)] for i in idx]
            rn = np.full((len(idx), MAXN), -1, dtype=np.int32)
            for k, l in enumerate(lists):
                rn[k, : len(l)] = l
            sub = extract_subgraphs(pairs[idx], graph, hop, max_neighbors, replay_nodes=rn,
                                    replay_n=np.asarray([len(l) for l in lists], dtype=np.int32))
            ii = torch.as_tensor(idx, device=batch.n.device)
            batch.nodes[ii], batch.n[ii], batch.tags[ii], batch.adj_bits[ii] = sub.nodes, sub.n, sub.tags, sub.adj_bits
            batch.tag_present |= sub.tag_present          # capped pairs came back with n = 0 (no tags) the first time
            batch.invalidate()                                 # the sizes changed
    labels = torch.cat([torch.zeros(len(negative), dtype=torch.uint8), torch.ones(len(positive), dtype=torch.uint8)])
    tag_map, tags_size = compact_tag_map(batch)
    return batch, labels.reshape(-1, 1), tags_size, tag_map, graph


_TAG_BOUND = 1 + 127 + 127 * 127


_SET_MIN_SLOTS = 8            # CPython's PySet_MINSIZE: every set has at least 8 slots and hash(i) == i for small ints


def _cpython_set_order(tag_lists):
    """the reference's tag order, by executing the reference's own set algebra (SEAL/config/subgraph.py:90-94):
    s = s.union(set(tags)) per subgraph, then list(s).  Host-side by nature: the order IS CPython's set iteration
    order.  tag_lists: an iterable of per-subgraph tag sequences (padding already removed)."""
    s = set()
    for tags in tag_lists:
        s = s.union(set(int(t) for t in tags))
    return list(s)


def _distinct_in_first_occurrence_order(tags_h, n_h):
    """per subgraph, its distinct tags in the order they first appear -- all set(tags) depends on (duplicates do not change
    a set) -- for the whole [B, 128] batch at once: one np.unique over (row, tag) keys instead of a Python pass over every
    vertex.  -> list of B short Python lists."""
    B, W = tags_h.shape
    valid = np.arange(W)[None, :] < n_h[:, None]
    rows, cols = np.nonzero(valid)
    key = rows.astype(np.int64) * (_TAG_BOUND + 2) + tags_h[rows, cols].astype(np.int64)
    uniq, first = np.unique(key, return_index=True)              # first = first flat position (row-major: vertex order)
    order = np.lexsort((cols[first], uniq // (_TAG_BOUND + 2)))
    r, t = (uniq // (_TAG_BOUND + 2))[order], (uniq % (_TAG_BOUND + 2))[order]
    cuts = np.searchsorted(r, np.arange(1, B))
    return [part.tolist() for part in np.split(t, cuts)]


def compact_tag_map(batch):
    """Global tag compaction (SEAL/config/subgraph.py:84-101): (tag_map int32 [TAG_BOUND + 1], T) with
    tag_map[raw tag] = compacted tag, -1 for absent tags.  The reference maps a tag to its position in
    list(set(all tags)), i.e. CPython set iteration order.  For non-negative ints below 8 that order is ascending
    whatever the history of the set (hash(i) = i, >= 8 slots, no collision can happen), and UNDIRECTED hop <= 1
    subgraphs only produce DRNL tags {0, 1, 2, 3, 4, 6}: every vertex is adjacent to a target, so its distance to the
    other one is <= 4 or infinite.  That case is one device launch (ascending rank of the presence table).
    Directed graphs (network_type = 1: a vertex reached from one target may be many hops from the other), hop 2 or
    replayed / hand-built batches do produce tags >= 8; then collisions in the set's hash table can reorder it, and
    the map is rebuilt on the host by running the reference's own set operations over the batch's per-subgraph tag
    sets -- same order as oracle.seal_int.compact_tags by construction.  That branch copies the [B, 128] tags to the
    host and takes ~0.3 s per 100 K subgraphs (a vectorised distinct-tags pass + one Python set union per subgraph):
    the "no host sync per forward" property of score_links / score_stack holds for the tag-< 8 case only."""
    dev = batch.tags.device
    if batch.tag_present is not None:                        # filled by the extraction kernel as it wrote the tags:
        tag_map, T = batch.tag_map()                         # ranked with the batch's offsets, one launch, one host read
        max_tag = batch._max_tag
    else:
        # presence table instead of a sort-based unique over [B, 128] tags: a DRNL tag on <= 128 vertices is at most
        # 1 + 127 + 127 * 127; slot 0 takes the -1 padding
        present = torch.zeros(_TAG_BOUND + 2, dtype=torch.uint8, device=dev)
        present[batch.tags.reshape(-1).long() + 1] = 1
        present = present[1:]
        rank = torch.cumsum(present, 0, dtype=torch.int32)
        tag_map = torch.where(present != 0, rank - 1, torch.full_like(rank, -1))
        idx = torch.arange(present.numel(), device=dev)
        T, max_tag = (int(v) for v in torch.stack([rank[-1].long(), (idx * (present != 0)).max()]).tolist())
    if max_tag < _SET_MIN_SLOTS or T == 0:
        return tag_map, T
    tags_h, n_h = batch.tags.cpu().numpy(), batch.n.cpu().numpy()
    order = _cpython_set_order(_distinct_in_first_occurrence_order(tags_h, n_h))
    if order == sorted(order) or set(order) == set(range(len(order))):
        # ascending already -- or the reference's "already contiguous" branch (:98), which leaves the tags alone
        return tag_map, T
    tm = torch.full((tag_map.numel(),), -1, dtype=torch.int32)
    tm[torch.as_tensor(order, dtype=torch.int64)] = torch.arange(len(order), dtype=torch.int32)
    return tm.to(dev), T


def gnn_inputs(batch, tag_map, tags_size, emb=None, fold_dinv=True, dtype=torch.float32, build_csr=True):
    """SEAL/config/data.py:94-140 batched: block-diagonal A~ = A + I (CSR, values 1 / 2),
    D^-1 = 1/rowsum(A~) per vertex, X = [onehot(tag) || emb[node]].  Returns (CSRGraph, d_inv, X).
    fold_dinv (default): the CSR values are D^-1 A~, so graph_conv_stack needs no per-row scale in the
    SpMM epilogue (a load at a row flush drains the gathers in flight); d_inv is still returned.
    X's rows are padded to a multiple of 16 bytes (a [total, T + dim] VIEW of a wider buffer), which is
    what lets the first linear read them as 16-byte pieces; dtype=torch.bfloat16 stores X in bf16.
    build_csr=False: only X is built (returns (None, None, X)) -- the dense conv stack
    (graph_conv_stack(..., batch=batch)) works from the bit adjacency and needs no CSR."""
    lib = _lib.load()
    dev = batch.n.device
    off = batch.node_off()
    total = batch.total()
    P, S = _ops._ptr, _ops._stream
    if not build_csr:
        g, d_inv = None, None
    else:
        g, d_inv = _block_csr(lib, batch, off, total, fold_dinv)
        d_inv = d_inv[:total]
    emb_dim = 0 if emb is None else emb.shape[1]
    emb = None if emb is None else _ops._c(emb, torch.float32)
    width = tags_size + emb_dim
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("gnn_inputs: dtype must be float32 or bfloat16")
    x = torch.empty((total, -(-width // 8) * 8), dtype=dtype, device=dev)[:, :width]
    fn = lib.lpg_seal_features_f32 if dtype == torch.float32 else lib.lpg_seal_features_bf16   # bf16: stored directly
    _lib.check(fn(P(batch.nodes), P(batch.n), P(off), P(batch.tags), P(tag_map), tag_map.numel(), tags_size, P(emb),
                  emb_dim, batch.B, P(x), x.stride(0), 1, P(_ops.err_flag(dev)), S()), "seal_features")
    _ops.maybe_check(dev, "seal feature index")
    return g, d_inv, x


def _block_csr(lib, batch, off, total, fold_dinv):
    """block-diagonal CSR of A~ = A + I with its SpMM plan, and D^-1 (the generic aggregation path)."""
    dev = batch.n.device
    nbytes = _lib.check(lib.lpg_seal_block_csr_bytes(total), "seal_block_csr_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rowptr = torch.empty(total + 1, dtype=torch.int64, device=dev)
    d_inv = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
    P, S = _ops._ptr, _ops._stream
    _lib.check(lib.lpg_seal_block_csr(P(batch.n), P(off), P(batch.adj_bits), batch.B, P(rowptr), 0, 0, P(d_inv), 0, 0,
                                      P(ws), nbytes, S()), "seal_block_csr(count)")
    nnz = int(rowptr[-1].item())
    col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
    _lib.check(lib.lpg_seal_block_csr(P(batch.n), P(off), P(batch.adj_bits), batch.B, P(rowptr), P(col), P(val),
                                      P(d_inv), nnz, int(fold_dinv), P(ws), nbytes, S()), "seal_block_csr(fill)")
    g = _ops.CSRGraph(rowptr, col[:nnz] if nnz else col, val[:nnz] if nnz else val, total, total, nnz)
    g._build_plan()
    g.n_empty = 0                       # A~ = A + I: every row holds its diagonal
    g.max_row = 129                     # <= LPG_SEAL_MAXN neighbours + the diagonal (known without a sync)
    g.rows_prescaled = bool(fold_dinv)
    return g, d_inv


_p_table_cache = {}


def _h1_tables(tags_size, emb, w1):
    """(p_table [n_graph, 32] fp32, w1_tag [T, 32] fp32, bf16?) -- the two tables layer 1 is gathered from, built once
    per (emb, W_1, T) version (for bf16 inputs: the fp32 products of the bf16 values)."""
    T = int(tags_size)
    bf16 = emb.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16
    if w1.shape[1] != 32 or w1.shape[0] != T + emb.shape[1]:
        raise ValueError(f"first_layer_h1: W_1 must be [{T + emb.shape[1]}, 32], got {tuple(w1.shape)}")
    ck = (emb.data_ptr(), emb._version, w1.data_ptr(), w1._version, T, tuple(emb.shape), bf16)
    hit = _p_table_cache.get(ck)
    if hit is None:
        if len(_p_table_cache) > 4:
            _p_table_cache.clear()
        w1f, embf = _ops._c(w1, torch.float32), _ops._c(emb, torch.float32)
        hit = _p_table_cache[ck] = (_ops.linear(embf, w1f[T:], w_is_km=True), w1f[:T].contiguous(), emb, w1)   # keeps the keys alive
    return hit[0], hit[1], bf16


def first_layer_h1(batch, tag_map, tags_size, emb, w1, check=True):
    """H_1 = X W_1 for X = [onehot(tag) || emb[node]] WITHOUT building X (SEAL/config/data.py:129-140 +
    SEAL/model/gnn.py:46): a one-hot row times W_1 is a row of W_1, and emb[node] @ W_1[T:] is row `node` of
    P = emb @ W_1[T:] (one small matmul), so H_1[g] = W_1[tag_g] + P[node_g].  W_1 [T + dim, 32].
    bf16 emb / W_1 (bf16 storage, configs[2]): the tables are the fp32 products of the bf16 values and H_1 is rounded
    once to bf16 -- bf16(X W_1) with fp32 accumulation."""
    lib = _lib.load()
    T = int(tags_size)
    p_table, w1_tag, bf16 = _h1_tables(T, emb, w1)
    off = batch.node_off()
    total = batch.total()
    h1 = torch.empty((total, 32), dtype=torch.bfloat16 if bf16 else torch.float32, device=emb.device)
    P = _ops._ptr
    fn = lib.lpg_seal_h1_tables_bf16 if bf16 else lib.lpg_seal_h1_tables_f32
    _lib.check(fn(P(batch.nodes), P(off), P(batch.tags), P(tag_map), tag_map.numel(), T, P(w1_tag), P(p_table),
                  emb.shape[0], batch.B, P(h1), h1.stride(0), P(_ops.err_flag(emb.device)), _ops._stream()), "seal_h1_tables")
    if check:                                         # check=False: the flag is read by the next checking op (head())
        _ops.maybe_check(emb.device, "seal first-layer index")
    return h1


def _dense_ok(weights, dtype=torch.float32):
    """the reference's architecture: graph weights [F,32], [32,32], [32,32], [32,1] (SEAL/model/gnn.py:32-42), all fp32
    or all bf16 (bf16 storage: configs[2])"""
    return (dtype in (torch.float32, torch.bfloat16) and len(weights) == 4
            and [tuple(w.shape)[1] for w in weights] == [32, 32, 32, 1]
            and [tuple(w.shape)[0] for w in weights[1:]] == [32, 32, 32] and all(w.dtype == dtype for w in weights)
            and DENSE_STACK)


def graph_conv_stack(g, d_inv, x, weights, batch=None, h1=None, head_k1=None):
    """SEAL/model/gnn.py:46-63: Z_l = tanh(D^-1 (A~ (Z_{l-1} W_l))), l = 1..4.
    Returns (concat(Z1, Z2, Z3) [total, 96], Z4[:, 0] -- the SortPooling key, :82).
    head_k1 ([96, 16], dense path only): conv1d #1 of the head is fused in and the first return value is
    concat(Z1, Z2, Z3) @ head_k1 [total, 16] instead -- the 96-wide concat is never written (head(a_all=...)).
    With `batch` (the SubgraphBatch) and the reference's fp32 architecture (widths 32/32/32/1) layers 1..4 run as
    dense 32 x 32 block products on the matrix cores straight from the bit adjacency
    (lpg_seal_conv_stack_f32; g / d_inv may then be None, and with h1 = first_layer_h1(...) so may x);
    otherwise the generic path:
    4 x (linear + block-diagonal CSR SpMM with the fused tanh epilogue)."""
    dt = (h1.dtype if h1 is not None else weights[0].dtype) if x is None else x.dtype
    if batch is not None and _dense_ok(weights, dt):
        lib = _lib.load()
        bf16 = dt == torch.bfloat16
        if h1 is None:
            h1 = _ops.linear(x, weights[0], w_is_km=True)           # the one big matmul: X W_1 (bf16 storage: bf16 MFMA)
        total = h1.shape[0]                                          # (h1 given: first_layer_h1, X never built)
        off = batch.node_off()
        win = CONV_WINDOW                                            # rows per workgroup window (<= 128)
        n_win = -(-total // win)
        if win == CONV_WINDOW and batch._first is not None and total == batch.total():
            first = batch._first                                     # computed with the node offsets (lpg_seal_batch_index)
        else:
            first = torch.empty(n_win + 1, dtype=torch.int32, device=h1.device)
            _lib.check(lib.lpg_seal_conv_windows(_ops._ptr(off), batch.B, n_win, win, _ops._ptr(first), _ops._stream()),
                       "seal_conv_windows")
        fuse = head_k1 is not None
        cat = None if fuse else torch.empty((total, 96), dtype=dt, device=h1.device)
        a16 = torch.empty((total, 16), dtype=torch.float32, device=h1.device) if fuse else None
        k1 = _ops._c(head_k1, torch.float32) if fuse else None
        if fuse and tuple(k1.shape) != (96, 16):
            raise ValueError("graph_conv_stack: head_k1 must be [96, 16]")
        key = torch.empty(total, dtype=torch.float32, device=h1.device)
        w2, w3, w4 = (_ops._c(w, dt) for w in weights[1:])
        P = _ops._ptr
        fn = lib.lpg_seal_conv_stack_bf16 if bf16 else lib.lpg_seal_conv_stack_f32
        _lib.check(fn(P(off), P(first), n_win, P(batch.adj_bits), P(h1), h1.stride(0), P(w2), P(w3), P(w4), P(cat), 96,
                      P(key), P(k1), P(a16), _ops._stream()), "seal_conv_stack")
        if bf16 and not fuse:
            key = key.to(torch.bfloat16)                             # the layer-by-layer bf16 path returns Z_4 as stored
        return (a16 if fuse else cat), key
    if head_k1 is not None:
        raise ValueError("graph_conv_stack: head_k1 is fused only on the dense path (batch=, fp32 widths 32/32/32/1)")
    if g is None:
        raise ValueError("graph_conv_stack: the generic path needs the block CSR (gnn_inputs(build_csr=True))")
    widths = [w.shape[1] for w in weights]
    cat = torch.empty((x.shape[0], sum(widths[:-1])), dtype=x.dtype, device=x.device)   # concat(Z1..Z3), written in place
    z, c0 = x, 0
    for li, w in enumerate(weights):                              # x bf16 => the whole stack runs in bf16 storage
        h = _ops.linear(z, w, w_is_km=True)                       # K5 on the matrix cores (fp32 or bf16 MFMA)
        rs = None if getattr(g, "rows_prescaled", False) else d_inv
        last = li == len(weights) - 1
        out = None if last else cat[:, c0:c0 + widths[li]]        # each layer lands in its column slice: no torch.cat
        z = g.spmm(h, row_scale=rs, act="tanh", out=out)          # K10: one fused kernel per layer
        c0 += widths[li]
    return cat, z[:, 0]


def score_stack(batch, tag_map, tags_size, emb, weights, head_k1, pool_k=None, fuse_tables=True):
    """The scoring form of the conv stack with both ends fused (lpg_seal_score_stack_f32 / _bf16):
    layer 1 gathered from the two tables in the kernel's prologue (H_1 never touches HBM), layers 1..4 + the head's
    conv1d #1 as graph_conv_stack(head_k1=...), and -- with pool_k -- SortPooling in the epilogue.
    Returns pooled [B, pool_k * 16] (pool_k given) or (a16 [total, 16], key [total]).  Needs the reference's layer
    widths (32 / 32 / 32 / 1), fp32 or bf16 storage."""
    lib = _lib.load()
    dt = weights[0].dtype
    if not _dense_ok(weights, dt) or emb.dtype != dt:
        raise ValueError("score_stack: needs the reference's architecture (widths 32/32/32/1), emb and weights of one dtype")
    T = int(tags_size)
    bf16 = dt == torch.bfloat16
    k1 = _ops._c(head_k1, torch.float32)
    if tuple(k1.shape) != (96, 16):
        raise ValueError("score_stack: head_k1 must be [96, 16]")
    dev = batch.n.device
    off = batch.node_off()
    # With pooled output nothing below needs the batch's vertex total on the HOST: the kernel reads the number of row
    # windows from node_off[B] on the device (n_windows = -1) and walks them with a bounded grid -- no sync between the
    # extraction and the scores.  (a16 / key outputs are sized by the total: that form reads it.)
    device_count = pool_k is not None and fuse_tables
    total = None if device_count else batch.total()
    n_win = -1 if device_count else -(-total // CONV_WINDOW)
    if batch._first is not None:
        first = batch._first
    else:
        first = torch.empty((batch.B + 2) if device_count else (n_win + 1), dtype=torch.int32, device=dev)
        _lib.check(lib.lpg_seal_conv_windows(_ops._ptr(off), batch.B, n_win, CONV_WINDOW, _ops._ptr(first), _ops._stream()),
                   "seal_conv_windows")
    P = _ops._ptr
    h1 = None
    p_table, w1_tag, _ = _h1_tables(T, emb, weights[0])
    if not fuse_tables:
        h1 = first_layer_h1(batch, tag_map, T, emb, weights[0], check=False)
    pooled = key = a16 = None
    if pool_k is not None:
        pooled = torch.empty((batch.B, int(pool_k) * 16), dtype=torch.float32, device=dev)
    else:
        key = torch.empty(total, dtype=torch.float32, device=dev)
        a16 = torch.empty((total, 16), dtype=torch.float32, device=dev)
    w2, w3, w4 = (_ops._c(w, dt) for w in weights[1:])
    fn = lib.lpg_seal_score_stack_bf16 if bf16 else lib.lpg_seal_score_stack_f32
    _lib.check(fn(P(off), P(first), n_win, P(batch.adj_bits), P(h1), 32 if h1 is None else h1.stride(0), P(batch.nodes),
                  P(batch.tags), P(tag_map), 0 if tag_map is None else tag_map.numel(), T, P(w1_tag), P(p_table), emb.shape[0],
                  P(w2), P(w3), P(w4), P(k1), P(key), P(a16), P(pooled), 0 if pool_k is None else int(pool_k), batch.B,
                  P(_ops.err_flag(dev)), _ops._stream()), "seal_score_stack")
    return pooled if pool_k is not None else (a16, key)


def sortpool(z, key, node_off, k):
    """SortPooling (SEAL/model/gnn.py:82-91): [B, k, C], rows ordered by key descending."""
    lib = _lib.load()
    z = _ops._c(z, torch.float32)
    key = _ops._c(key, torch.float32)
    B = node_off.numel() - 1
    out = torch.empty((B, k, z.shape[1]), dtype=torch.float32, device=z.device)
    _lib.check(lib.lpg_seal_sortpool_f32(_ops._ptr(z), z.stride(0), _ops._ptr(key), _ops._ptr(node_off), B, k,
                                         z.shape[1], _ops._ptr(out), _ops._stream()), "seal_sortpool")
    return out


_fold_cache = {}


def _folded_conv2_dense(hw, k):
    """[k * c1, dense] matrix of conv1d #2 (SEAL/model/gnn.py:113-118) composed with weight_1 (:124-129):
    Wc[(t + dt) * c1 + i, j] += sum_c K2[dt, i, c] * W1[t * c2 + c, j] for every window t and tap dt.
    Weight preprocessing, done once per (weights, version) in fp64 and rounded to fp32."""
    k2, w1 = hw["conv1d_kernel_2"], hw["weight_1"]
    key = (k2.data_ptr(), k2._version, w1.data_ptr(), w1._version, k)
    hit = _fold_cache.get(key)
    if hit is not None:
        return hit[0]
    fw, c1, c2 = k2.shape
    t_out = k - fw + 1
    w1r = w1.double().view(t_out, c2, -1)                               # [t, c, j]
    wc = torch.zeros((k, c1, w1.shape[1]), dtype=torch.float64, device=w1.device)
    for dt in range(fw):
        wc[dt:dt + t_out] += torch.einsum("ic,tcj->tij", k2[dt].double(), w1r)
    out = wc.view(k * c1, -1).float().contiguous()
    if len(_fold_cache) > 8:
        _fold_cache.clear()
    _fold_cache[key] = (out, k2, w1)                                    # keeps the key tensors alive
    return out


def head(z, key, node_off, hw, a_all=None, check=True, pooled=None):
    """The DGCNN head of SEAL/model/gnn.py:82-137, batched over all subgraphs:
    SortPooling -> conv1d(96->16, stride 96) -> conv1d(width 5, 16->32) -> dense 128 relu -> dense 2
    -> softmax.  Every stage is one launch on the whole batch.  Neither convolution has an activation: conv1d #1
    is a per-vertex linear applied BEFORE the pooling (or inside the conv stack), conv1d #2 is folded into the
    dense layer's matrix.
    hw: dict with conv1d_kernel_1 [96,1,16], conv1d_kernel_2 [5,16,32], weight_1 [448,128], bias_1,
    weight_2 [128,2], bias_2 (the shipped checkpoint's tensors).  Returns (pre_y [B,2], score [B]).
    a_all: z @ conv1d_kernel_1 [total, 16] when the conv stack already produced it (graph_conv_stack(head_k1=));
    z may then be None.  pooled: [B, k * 16] when the stack also pooled (score_stack(pool_k=k)); z, key, a_all unused.  check=False: the forward's deferred index check (one host sync) is left to the caller --
    ops.raise_if_flagged(device) after a scoring loop -- so consecutive forwards queue up without a gap."""
    lib = _lib.load()
    k1 = hw["conv1d_kernel_1"].reshape(-1, hw["conv1d_kernel_1"].shape[-1])          # [96, 16]
    fw, c1, c2 = hw["conv1d_kernel_2"].shape
    k = hw["weight_1"].shape[0] // c2 + fw - 1                                           # threshold_k (18)
    B = node_off.numel() - 1
    # conv1d #1 is a per-vertex linear (stride = filter width = one 96-wide row) with no activation, and SortPooling
    # only gathers rows (zero rows stay zero): conv1d(sortpool(z)) == sortpool(z @ k1) value for value.  Multiplying
    # first moves 16-float rows through the pooling instead of 96-float ones (1.25 GB less traffic per 100 K subgraphs).
    if pooled is not None:                                                               # pooled by the conv stack's epilogue
        a = pooled.view(B, k * c1)
    else:
        if a_all is None:
            a_all = _ops.linear(z, k1, w_is_km=True)
        a = sortpool(a_all, key, node_off, k).view(B, k * c1)                            # [B, k*16]
    # conv1d #2 (width 5, stride 1, VALID) has NO activation and neither has the flatten: conv1d #2 followed by the
    # dense layer is ONE linear map from the pooled [k, 16] block to the 128 dense units.  Its matrix is folded once
    # per set of weights; the [B*k, 32] convolution output (230 MB per 100 K subgraphs) is never formed.
    dense = _ops.linear(a, _folded_conv2_dense(hw, k), hw["bias_1"], w_is_km=True, act="relu")
    score = torch.empty(B, dtype=torch.float32, device=a.device)
    w2 = _ops._c(hw["weight_2"], torch.float32)
    if tuple(w2.shape) == (dense.shape[1], 2) and dense.shape[1] % 4 == 0 and dense.stride(0) % 4 == 0:
        pre_y = torch.empty((B, 2), dtype=torch.float32, device=a.device)       # last layer + softmax: one launch
        _lib.check(lib.lpg_seal_logits_f32(_ops._ptr(dense), dense.stride(0), B, dense.shape[1], _ops._ptr(w2),
                                           _ops._ptr(_ops._c(hw["bias_2"], torch.float32)), _ops._ptr(pre_y),
                                           _ops._ptr(score), _ops._stream()), "seal_logits")
    else:
        pre_y = _ops.linear(dense, hw["weight_2"], hw["bias_2"], w_is_km=True)
        _lib.check(lib.lpg_softmax2_pos_score_f32(_ops._ptr(pre_y), B, _ops._ptr(score), _ops._stream()), "softmax2")
    if check:
        _ops.maybe_check(a.device, "seal forward index")   # the one deferred check of the scoring forward
    return pre_y, score


def score_links(positive, negative, nodes_size, emb, weights, head_weights, test_ratio=0.1, hop=1, device="cuda",
                network_type=0):
    """the SEAL scoring forward end to end for a candidate set: structure (a10-a12), graph
    convolutions (a9) and the head.  Returns (scores [B], labels [B,1], batch)."""
    batch, labels, T, tag_map, _ = link2subgraph(positive, negative, nodes_size, test_ratio, hop, network_type, device=device)
    if emb is not None and _dense_ok(weights, weights[0].dtype) and emb.dtype == weights[0].dtype:   # X is never built: H_1 from two row gathers, then the dense stack
        k1 = head_weights["conv1d_kernel_1"]
        k1 = k1.reshape(-1, k1.shape[-1])
        fuse = tuple(k1.shape) == (96, 16) and _lib.get_option("seal_conv_fp32") == 0
        if fuse and _lib.get_option("seal_conv_variant") == 0:
            # tags / nodes in, pooled rows out: one launch between the extraction and the dense layers
            fw, c1, c2 = head_weights["conv1d_kernel_2"].shape
            pool_k = head_weights["weight_1"].shape[0] // c2 + fw - 1
            pooled = score_stack(batch, tag_map, T, emb, weights, k1, pool_k=pool_k)
            _, score = head(None, None, batch.node_off(), head_weights, pooled=pooled)
            return score, labels, batch
        h1 = first_layer_h1(batch, tag_map, T, emb, weights[0], check=False)       # head() reads the flag at the end
        z, key = graph_conv_stack(None, None, None, weights, batch=batch, h1=h1, head_k1=k1 if fuse else None)
        _, score = head(None if fuse else z, key, batch.node_off(), head_weights, a_all=z if fuse else None)
        return score, labels, batch
    g, d_inv, x = gnn_inputs(batch, tag_map, T, emb)
    z, key = graph_conv_stack(g, d_inv, x, weights, batch=batch)
    _, score = head(z, key, batch.node_off(), head_weights)
    return score, labels, batch


def score_links_from_file(path, emb, weights, head_weights, network_type=0, test_ratio=0.1, hop=1, seed=0, device="cuda"):
    """edge file -> scores: the candidate loader of SEAL/config/data.py:10-32 (data.seal_load_data: header row eaten,
    sorted non-edges shuffled under `seed`, positives shuffled) followed by the scoring forward above -- the
    reference's `execute()` (SEAL/operators/seal_link_predict.py) from load_data to the per-candidate P(link), minus
    training.  emb: [n_nodes, dim] node embedding (the reference trains node2vec for it; any table of that shape).
    Returns (scores [2P], labels [2P, 1], batch, (positive, negative, n_nodes))."""
    from . import data as _data
    positive, negative, n_nodes = _data.seal_load_data(path, network_type, seed)
    if emb is not None and emb.shape[0] != n_nodes:
        raise ValueError(f"score_links_from_file: emb has {emb.shape[0]} rows, the edge file has {n_nodes} nodes")
    score, labels, batch = score_links(positive, negative, n_nodes, emb, weights, head_weights, test_ratio, hop, device,
                                       network_type)
    return score, labels, batch, (positive, negative, n_nodes)
