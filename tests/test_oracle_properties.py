"""CPU: property tests (hypothesis) that the oracle's restatements agree with independent dense
formulations -- two derivations of the same definition, as SURVEY.md section 7 asks for the
float path whose third-party reference cannot be imported."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import aggregation as agg, seal_int, twowl_int


@st.composite
def graphs(draw, max_n=12, max_e=60):
    n = draw(st.integers(1, max_n))
    e = draw(st.integers(0, max_e))
    src = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    dst = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    return n, np.asarray([src, dst], dtype=np.int64).reshape(2, e)


@settings(max_examples=60, deadline=None)
@given(graphs(), st.integers(1, 5), st.booleans())
def test_csr_spmm_equals_dense(g, d, weighted):
    n, ei = g
    rng = np.random.default_rng(0)
    w = rng.standard_normal(ei.shape[1]) if weighted else None
    x = rng.standard_normal((n, d))
    A = np.zeros((n, n))
    np.add.at(A, (ei[1], ei[0]), 1.0 if w is None else w)
    rowptr, col, val, perm = agg.coo_to_csr(ei[1], ei[0], n, w)
    assert np.allclose(agg.csr_spmm(rowptr, col, val, x), A @ x)
    assert np.array_equal(np.sort(perm), np.arange(ei.shape[1]))
    assert np.all(np.diff(perm[np.argsort(ei[1][perm], kind="stable")]) != 0) or ei.shape[1] < 2


@settings(max_examples=60, deadline=None)
@given(graphs(), st.integers(1, 4))
def test_gcn_conv_equals_dense_normalised_adjacency(g, d):
    """GCNConv == D^-1/2 (A_noloops + I) D^-1/2 (X W^T) + b with duplicate edges counted."""
    n, ei = g
    rng = np.random.default_rng(1)
    x, wgt, b = rng.standard_normal((n, d)), rng.standard_normal((3, d)), rng.standard_normal(3)
    A = np.zeros((n, n))
    m = ei[0] != ei[1]
    np.add.at(A, (ei[1][m], ei[0][m]), 1.0)
    A += np.eye(n)
    dinv = A.sum(axis=1) ** -0.5
    want = (dinv[:, None] * A * dinv[None, :]) @ (x @ wgt.T) + b
    assert np.allclose(agg.gcn_conv(x, ei, wgt, b), want)


@settings(max_examples=40, deadline=None)
@given(st.integers(2, 40), st.integers(1, 6))
def test_graph_norm_moments(n, d):
    rng = np.random.default_rng(n * 7 + d)
    x = rng.standard_normal((n, d)) * 3 + 1
    y = agg.graph_norm(x, np.ones(d), np.zeros(d), np.ones(d), eps=0.0) if n > 1 and np.all(x.std(0) > 0) else None
    if y is not None:
        assert np.allclose(y.mean(0), 0, atol=1e-9) and np.allclose((y * y).mean(0), 1, atol=1e-9)


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 8), st.integers(0, 20), st.integers(0, 20), st.integers(0, 2 ** 31))
def test_get_ei2_is_the_tuple_line_graph(n, e1, ep, seed):
    rng = np.random.default_rng(seed)
    pos = rng.integers(0, n, size=(2, e1))
    pred = rng.integers(0, n, size=(2, ep))
    ei2 = twowl_int.get_ei2(n, pos, pred)
    edge = np.concatenate([pos, pred], axis=1)
    want = {(a, b) for a in range(e1) for b in range(e1 + ep) if pos[1, a] == edge[0, b]}
    got = list(map(tuple, ei2.T.tolist()))
    assert set(got) == want and len(got) == len(want)
    heads = pos[1][ei2[0]] if ei2.shape[1] else np.zeros(0)
    assert np.all(np.diff(heads) >= 0)                       # node-major order
    e, er = twowl_int.reverse(ei2)
    assert np.array_equal(twowl_int.flip(twowl_int.flip(ei2[0])), ei2[0])
    assert np.array_equal(e[1], ei2[1]) and np.array_equal(er[0], ei2[0])


@settings(max_examples=60, deadline=None)
@given(st.integers(2, 14), st.integers(0, 2 ** 31))
def test_drnl_invariants(n, seed):
    rng = np.random.default_rng(seed)
    a = (rng.random((n, n)) < 0.3).astype(np.uint8)
    a = np.triu(a, 1)
    a = a + a.T
    a[0, 1] = a[1, 0] = 0
    tags = seal_int.node_labeling(a)
    assert tags[:2] == [1, 1] and len(tags) == n and min(tags) >= 0
    p = np.concatenate([[1, 0], np.arange(2, n)])             # swapping the two targets keeps the labels
    assert seal_int.node_labeling(a[p][:, p]) == tags
    for i in range(2, n):                                     # direct common neighbour of both targets -> 2
        if a[0, i] and a[1, i]:
            assert tags[i] == 2


def test_c_restatement_matches_numpy_oracle():
    """oracle/spmm_ref.c (fp32, kernel association order) vs the fp64 numpy restatement."""
    from oracle import spmm_c
    rng = np.random.default_rng(5)
    n, e, d = 200, 5000, 12
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    dst[:1500] = 7                                        # one row spanning many chunks
    x = rng.standard_normal((n, d)).astype(np.float32)
    for w in (None, rng.standard_normal(e).astype(np.float32)):
        rp, col, val, _ = agg.coo_to_csr(dst, src, n, w)
        want = agg.csr_spmm(rp, col, val, x)
        mag = agg.spmm_abs_rowsum(rp, col, val, x)
        for chunk in (16, 64, 512):
            got = spmm_c.spmm_f32_chunked(rp, col, val, x, chunk)
            assert np.all(np.abs(got - want) <= 1e-5 * mag + 1e-30)
        got = spmm_c.scatter_add_f32(src, dst, w, x, n)
        assert np.all(np.abs(got - want) <= 1e-5 * mag + 1e-30)
