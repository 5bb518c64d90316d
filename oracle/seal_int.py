"""Oracle (TEST INFRASTRUCTURE): restatement of SEAL's integer structure path.

Follows SEAL/config/subgraph.py:11-141 (link2subgraph / extract_subgraph),
SEAL/utils/utils.py:35-54 (node_labeling) and SEAL/config/data.py:94-140 (A~, D^-1, X).
Pinned bit-exactly against tests/golden/seal_int.npz (outputs of the reference itself).

Pure Python / numpy, no networkx: BFS on the dense 0/1 adjacency.
"""
import numpy as np


def build_train_adjacency(positive, n_nodes, test_ratio=0.1, network_type=0):
    """SEAL/config/subgraph.py:23-31 -- only the first len(positive)-test_size positives enter
    A (undirected: both orientations set; network_type = 1: the stored orientation only; self loops sit on the
    diagonal).  Everything below reads ROWS of A (successors), as nx.neighbors / nx.DiGraph(data=adj) do, so the
    same functions restate the directed case."""
    positive = np.asarray(positive, dtype=np.int64)
    test_size = int(len(positive) * test_ratio)
    train_pos = positive[:-test_size]
    A = np.zeros((n_nodes, n_nodes), dtype=np.uint8)
    A[train_pos[:, 0], train_pos[:, 1]] = 1
    if network_type == 0:
        A[train_pos[:, 1], train_pos[:, 0]] = 1
    return A


def neighbours(A, node):
    """nx.neighbors on the Graph built from train_pos == nonzero columns of row `node`
    (a self loop lists the node itself).  A: dense uint8 [N,N] or a scipy.sparse CSR matrix
    (large synthetic graphs)."""
    if hasattr(A, "indptr"):
        return np.sort(A.indices[A.indptr[node]:A.indptr[node + 1]])
    return np.nonzero(A[node])[0]


def enclosing_nodes(node_pair, A, hop, max_neighbors=100):
    """SEAL/config/subgraph.py:118-135 for hop in {0, 1}.

    Returns (nodes, capped).  When the 100-node cap does not bite the result is RNG-free:
    nodes = [u, v] + the rest in ASCENDING id order (canonical order; the reference's order is
    CPython set iteration order, which tests undo with a permutation).  When
    len(sub) + len(neighbours) >= max_neighbors for one of the two targets the reference
    truncates a *shuffled* neighbour list (:125-130) -- RNG dependent, so `capped` is returned
    True and nodes is None; callers replay the reference's node list instead.
    hop >= 2 raises in the reference itself (set is not shufflable, :122,131) -> not restated.
    """
    u, v = int(node_pair[0]), int(node_pair[1])
    sub = {u, v}
    if int(hop) not in (0, 1):
        raise ValueError("reference crashes for hop >= 2; only hop 0/1 are defined")
    if int(hop) == 1:
        # the reference visits [u, v] in a shuffled order (:122); report `capped` when the
        # truncation branch (:124-130) would fire in EITHER order
        for first, second in ((u, v), (v, u)):
            s = {u, v}
            for node in (first, second):
                nb = neighbours(A, node)
                if len(s) + len(nb) < max_neighbors:
                    s |= set(int(x) for x in nb)
                else:
                    return None, True
        sub = s
    sub.discard(u)
    sub.discard(v)
    return [u, v] + sorted(sub), False


def induced_adjacency(nodes, A):
    """SEAL/config/subgraph.py:136-137 -- A[nodes][:, nodes] with the target entries cleared."""
    nodes = np.asarray(nodes, dtype=np.int64)
    if hasattr(A, "indptr"):
        adj = (A[nodes, :][:, nodes].toarray() != 0).astype(np.uint8)
    else:
        adj = A[nodes, :][:, nodes].copy()
    adj[0, 1] = adj[1, 0] = 0
    return adj


def _bfs(adj, src):
    n = adj.shape[0]
    dist = np.full(n, -1, dtype=np.int64)
    dist[src] = 0
    frontier = [src]
    while frontier:
        nxt = []
        for a in frontier:
            for b in np.nonzero(adj[a])[0]:
                if dist[b] < 0:
                    dist[b] = dist[a] + 1
                    nxt.append(int(b))
        frontier = nxt
    return dist


def node_labeling(adj):
    """SEAL/utils/utils.py:35-54 (undirected) -- DRNL variant.  Targets (rows 0, 1) get 1;
    node i >= 2: dx = sp(0,i), dy = sp(1,i) on the subgraph AS GIVEN (the other target is not
    masked); unreachable from either => 0; else
    tag = 1 + min(dx,dy) + (d//2) * ((d//2) + (d%2) - 1),  d = dx + dy."""
    adj = np.asarray(adj)
    n = adj.shape[0]
    if n <= 2:
        return [1, 1]
    d0, d1 = _bfs(adj, 0), _bfs(adj, 1)
    tags = [1, 1]
    for i in range(2, n):
        dx, dy = int(d0[i]), int(d1[i])
        if dx < 0 or dy < 0:
            tags.append(0)
            continue
        d = dx + dy
        div, mod = divmod(d, 2)
        tags.append(1 + min(dx, dy) + div * (div + mod - 1))
    return tags


def compact_tags(vertex_tags):
    """SEAL/config/subgraph.py:84-101 -- global tag compaction.  The reference maps tag x to
    its position in ``list(set_of_all_tags)`` (CPython set iteration order) unless the tag set
    already equals range(T).  Restated with the same Python set operations."""
    s = set()
    for tags in vertex_tags:
        s = s.union(set(int(t) for t in tags))
    lst = list(s)
    if set(range(len(lst))) != set(lst):
        m = {x: lst.index(x) for x in lst}
        vertex_tags = [[m[int(t)] for t in tags] for tags in vertex_tags]
    else:
        vertex_tags = [[int(t) for t in tags] for tags in vertex_tags]
    return vertex_tags, len(lst)


def gnn_inputs(adj, tags, tags_size, emb_rows=None):
    """SEAL/config/data.py:94-140 -- A~ = A + I (uint8; a self loop gives a 2 on the
    diagonal), D^-1 = 1/rowsum(A~) (float64, the diagonal of inv(diag(.))),
    X = [onehot(tag, T) || emb[sub_nodes]]."""
    adj = np.asarray(adj, dtype=np.uint8)
    a_tilde = adj + np.eye(adj.shape[0], dtype=np.uint8)
    d_inv = 1.0 / a_tilde.sum(axis=1).astype(np.float64)
    x = np.eye(tags_size, dtype=np.uint8)[np.asarray(tags, dtype=np.int64)].astype(np.float64)
    if emb_rows is not None:
        x = np.concatenate([x, np.asarray(emb_rows, dtype=np.float64)], axis=1)
    return a_tilde, d_inv, x
