"""Oracle (TEST INFRASTRUCTURE): numpy restatement of the integer utilities of TwoWL/utils.py.

Pinned bit-exactly against tests/golden/twowl_int.npz (outputs of the reference functions
themselves, see tests/golden/make_golden.py).  All arrays int64 unless stated.
"""
import numpy as np


def degree(ei, num_node):
    """TwoWL/utils.py:8-10 -- scatter_add(ones_like(ei[1]), ei[1], dim_size=num_node):
    in-degree histogram over the destination row of ``ei`` ([2,E])."""
    return np.bincount(np.asarray(ei[1], dtype=np.int64), minlength=num_node).astype(np.int64)


def set_mul(a, b):
    """TwoWL/utils.py:13-19 -- cartesian product, a-major / b-minor, shape [|a|*|b|, 2]."""
    a = np.asarray(a, dtype=np.int64).reshape(-1)
    b = np.asarray(b, dtype=np.int64).reshape(-1)
    return np.stack([np.repeat(a, b.size), np.tile(b, a.size)], axis=1)


def check_in_set(target, s):
    """TwoWL/utils.py:22-33 -- for each target element, how many elements of ``s`` equal it."""
    target = np.asarray(target, dtype=np.int64).reshape(-1)
    s = np.asarray(s, dtype=np.int64).reshape(-1)
    if s.size == 0:
        return np.zeros(target.shape, dtype=np.int64)
    vals, cnt = np.unique(s, return_counts=True)
    pos = np.searchsorted(vals, target)
    pos_c = np.minimum(pos, vals.size - 1)
    hit = vals[pos_c] == target
    return np.where(hit, cnt[pos_c], 0).astype(np.int64)


def get_ei2(n_node, pos_edge, pred_edge):
    """TwoWL/utils.py:36-45 -- 2-WL tuple graph.  Tuple ids are column indices into
    cat(pos_edge, pred_edge).  For node i = 0..n-1 (ascending): every pair (a, b) with
    pos_edge[1, a] == i (a ascending, a < |pos_edge|) and edge[0, b] == i (b ascending);
    a-major, b-minor.  Returns [2, E2] (row 0 = a, row 1 = b)."""
    pos_edge = np.asarray(pos_edge, dtype=np.int64)
    edge = np.concatenate([pos_edge, np.asarray(pred_edge, dtype=np.int64)], axis=1)
    out = []
    for i in range(n_node):
        a = np.nonzero(pos_edge[1] == i)[0]
        b = np.nonzero(edge[0] == i)[0]
        out.append(set_mul(a, b))
    if not out:
        return np.zeros((2, 0), dtype=np.int64)
    return np.concatenate(out, axis=0).T.copy()


def blockei2(ei2, blocked_idx):
    """TwoWL/utils.py:48-50 -- drop tuple-graph edges whose SOURCE tuple (row 0) is blocked."""
    keep = check_in_set(ei2[0], blocked_idx) == 0
    return np.asarray(ei2)[:, keep]


def idx2mask(num, idx):
    """TwoWL/utils.py:53-57."""
    m = np.zeros(num, dtype=bool)
    m[np.asarray(idx, dtype=np.int64)] = True
    return m


def sample_block(sample_idx, size, ei, ei2=None):
    """TwoWL/utils.py:61-68 -- remove the sampled edge columns from ``ei``; x_new = number of
    kept edges per SOURCE node (row sum of the sparse adjacency, float ones -> int64); filter
    ``ei2`` by blocked source tuple."""
    ei = np.asarray(ei, dtype=np.int64)
    keep = ~idx2mask(ei.shape[1], sample_idx)
    ei_new = ei[:, keep]
    ei2_new = blockei2(np.asarray(ei2, dtype=np.int64), sample_idx) if ei2 is not None else None
    x_new = np.bincount(ei_new[0], minlength=size).astype(np.int64)
    return ei_new, x_new, ei2_new


def flip(k):
    """partner-direction tuple id: k+1 for even k, k-1 for odd k (TwoWL/utils.py:73-74)."""
    k = np.asarray(k, dtype=np.int64)
    return k + 1 - 2 * (k & 1)


def reverse(edge_index):
    """TwoWL/utils.py:71-78 -- returns (edge, edge_r) = ((flip(e0), e1), (e0, flip(e1)))."""
    e = np.asarray(edge_index, dtype=np.int64)
    return np.stack([flip(e[0]), e[1]]), np.stack([e[0], flip(e[1])])


def double(x, for_index=False):
    """TwoWL/utils.py:81-90 -- edges: [2,E] -> [2,2E] with (r,c) at 2k and (c,r) at 2k+1;
    indices: [B] -> [2B] with 2i, 2i+1 interleaved."""
    x = np.asarray(x, dtype=np.int64)
    if not for_index:
        r, c = x[0], x[1]
        out = np.empty((2, 2 * x.shape[1]), dtype=np.int64)
        out[0, 0::2], out[0, 1::2] = r, c
        out[1, 0::2], out[1, 1::2] = c, r
        return out
    out = np.empty(2 * x.shape[0], dtype=np.int64)
    out[0::2], out[1::2] = 2 * x, 2 * x + 1
    return out
