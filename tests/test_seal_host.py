"""CPU: host-side logic of the SEAL scoring path that needs no kernel -- the conv1d #2 / dense fold of the head, the
per-graph hub bit matrix, the train-graph CSR.  (The kernels that consume them are held to the oracle in
tests/test_gpu_parity.py.)"""
import os

import numpy as np
import pytest
import torch

from lpgnn_amd import seal


def test_folded_conv2_dense_equals_the_two_layers():
    """head(): conv1d #2 (width 5, stride 1, VALID, no activation) followed by flatten + dense is ONE linear map of
    the pooled [k, 16] block (SEAL/model/gnn.py:113-129); the folded matrix against the two layers in fp64."""
    rng = np.random.default_rng(0)
    k, fw, c1, c2, dn = 18, 5, 16, 32, 128
    hw = {"conv1d_kernel_2": torch.from_numpy(rng.standard_normal((fw, c1, c2)).astype(np.float32)),
          "weight_1": torch.from_numpy(rng.standard_normal(((k - fw + 1) * c2, dn)).astype(np.float32))}
    wc = seal._folded_conv2_dense(hw, k)
    assert wc.shape == (k * c1, dn) and wc.dtype == torch.float32
    a = rng.standard_normal((7, k, c1))
    k2, w1 = hw["conv1d_kernel_2"].double().numpy(), hw["weight_1"].double().numpy()
    conv = np.zeros((7, k - fw + 1, c2))
    for t in range(k - fw + 1):
        for dt in range(fw):
            conv[:, t] += a[:, t + dt] @ k2[dt]
    want = conv.reshape(7, -1) @ w1
    got = a.reshape(7, -1) @ wc.double().numpy()
    assert np.abs(got - want).max() <= 1e-5 * (np.abs(want).max() + 1.0)
    assert seal._folded_conv2_dense(hw, k) is wc                      # cached per weights version
    hw["weight_1"].add_(1.0)                                          # an in-place update must miss the cache
    assert seal._folded_conv2_dense(hw, k) is not wc


def test_hub_bit_matrix_on_cpu():
    """TrainGraph.build_hub_matrix: bit (a, b) == hubs a and b are adjacent, hub_index maps nodes to rows."""
    rng = np.random.default_rng(1)
    n = 3000
    hubs = rng.choice(n, size=40, replace=False)
    edges = [rng.integers(0, n, size=(70_000, 2))]
    for h in hubs:                                                    # 40 nodes with > 256 neighbours
        edges.append(np.stack([np.full(400, h), rng.integers(0, n, size=400)], 1))
    edges.append(np.stack([hubs[:20], hubs[20:]], 1))                 # some hub-hub edges for sure
    pos = np.concatenate(edges).astype(np.int64)
    pos = pos[pos[:, 0] != pos[:, 1]]
    g = seal.build_train_csr(pos, n, test_ratio=0.0, device="cpu")
    assert g.hub_bits is not None and g.col.numel() >= (1 << 17)
    deg = (g.rowptr[1:] - g.rowptr[:-1]).numpy()
    hub_nodes = np.nonzero(deg > g.hub_degree)[0]
    assert set(hubs.tolist()) <= set(hub_nodes.tolist()) and g.hub_bits.shape[0] == len(hub_nodes)
    idx = g.hub_index.numpy()
    assert np.array_equal(np.nonzero(idx >= 0)[0], hub_nodes) and np.array_equal(idx[hub_nodes], np.arange(len(hub_nodes)))
    bits = g.hub_bits.numpy().view(np.uint32)
    rp, cl = g.rowptr.numpy(), g.col.numpy()
    for a, na in enumerate(hub_nodes):
        nb = set(cl[rp[na]:rp[na + 1]].tolist())
        row = [(bits[a, b >> 5] >> (b & 31)) & 1 for b in range(len(hub_nodes))]
        assert row == [int(int(nbn) in nb) for nbn in hub_nodes]
    small = seal.build_train_csr(pos[:1000], n, test_ratio=0.0, device="cpu")
    assert small.hub_bits is None and small.hub_index is None          # small graphs keep the searches (no sync)


def test_train_csr_is_symmetric_sorted_unique():
    rng = np.random.default_rng(2)
    pos = rng.integers(0, 50, size=(400, 2)).astype(np.int64)
    g = seal.build_train_csr(pos, 50, test_ratio=0.1, device="cpu")
    rp, cl = g.rowptr.numpy(), g.col.numpy()
    adj = np.zeros((50, 50), dtype=bool)
    for u, v in pos[: len(pos) - int(len(pos) * 0.1)]:
        adj[u, v] = adj[v, u] = True
    for r in range(50):
        assert cl[rp[r]:rp[r + 1]].tolist() == np.nonzero(adj[r])[0].tolist()


def test_three_way_bf16_split_is_exact():
    """the split the SEAL conv kernel applies to H (csrc/seal_conv.hip, split3): hi = top 16 bits of v, mid = top 16
    bits of (v - hi), lo = (v - hi - mid) truncated to 16 bits -- restated in numpy: every term is a bf16 value and
    hi + mid + lo == v EXACTLY, so sum_k a_k * v_k with small-integer a_k (the 0 / 1 / 2 adjacency entries) is the
    sum of three exact bf16 products per term.  (Exact while the third term stays a NORMAL bf16 number: |v| >= 2^-110;
    below that the loss is under 2^-133 absolute -- the last two asserts pin that too.)"""
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.standard_normal(200_000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 200_000),
                        np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -110, 65504.0, 1e38, -3e-33], dtype=np.float32)])
    v = v.astype(np.float32)
    tiny = v[(np.abs(v) < 2.0 ** -110) & (v != 0)]
    v = v[(np.abs(v) >= 2.0 ** -110) | (v == 0)]

    def top16(x):
        return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)

    hi = top16(v)
    r1 = (v - hi).astype(np.float32)
    mid = top16(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = top16(r2)
    assert np.array_equal(lo, r2)                                    # nothing is lost in the third term
    back = (hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64))
    assert np.array_equal(back, v.astype(np.float64))                # exact, not just close
    th = top16(tiny)
    tr1 = (tiny - th).astype(np.float32)
    tm = top16(tr1)
    tl = top16((tr1 - tm).astype(np.float32))
    assert np.abs(th.astype(np.float64) + tm + tl - tiny).max(initial=0.0) < 2.0 ** -133
    for a in (1.0, 2.0):                                             # products with the adjacency's 1 / 2 are exact too
        for t in (hi, mid, lo):
            p = (np.float32(a) * t).astype(np.float32)
            assert np.array_equal(p.astype(np.float64), a * t.astype(np.float64))


def _hand_batch(tag_lists, device="cpu"):
    """a SubgraphBatch carrying only what compact_tag_map reads: per-subgraph sizes and (-1 padded) raw tags"""
    B = len(tag_lists)
    tags = np.full((B, seal.MAXN), -1, dtype=np.int32)
    for b, t in enumerate(tag_lists):
        tags[b, : len(t)] = t
    n = torch.tensor([len(t) for t in tag_lists], dtype=torch.int32, device=device)
    pairs = torch.zeros((B, 2), dtype=torch.int64, device=device)
    return seal.SubgraphBatch(pairs, None, n, torch.from_numpy(tags).to(device), None, None, None)


# tag sets the reference's `list(set(...))` (SEAL/config/subgraph.py:90-94) does NOT list in ascending order: values
# at or above the set's slot count collide (hash(i) = i), e.g. 8 and 16 land on slot 0 of an 8-slot table
NONCONTIGUOUS_TAG_LISTS = [
    [[1, 1, 2, 3], [1, 1, 0, 10], [1, 1, 16, 2]],                       # the verdict's {0,1,2,3,10,16}
    [[1, 1, 9], [1, 1, 17, 8], [1, 1, 16]],                             # collisions modulo 8
    [[1, 1, 33, 2], [1, 1, 65, 3, 4], [1, 1, 6, 8, 11, 14, 18, 22, 27, 32]],
    [[1, 1, 2, 3, 4, 6]],                                               # fb-pages-food's own set: ascending
    [[1, 1], [1, 1, 0]],                                                # already contiguous: identity branch (:98)
    [[1, 1, 1000, 24], [1, 1, 16257, 40, 8]],
]


def test_compact_tag_map_follows_the_reference_set_order():
    """seal.compact_tag_map against oracle.seal_int.compact_tags (= the reference's set algebra) on tag sets that are
    not contiguous and not in ascending set order: every raw tag must map to the same compacted tag."""
    from oracle import seal_int
    saw_unsorted = False
    for lists in NONCONTIGUOUS_TAG_LISTS:
        want_tags, want_T = seal_int.compact_tags(lists)
        tag_map, T = seal.compact_tag_map(_hand_batch(lists))
        assert T == want_T
        tm = tag_map.numpy()
        for raw, want in zip(lists, want_tags):
            assert [int(tm[t]) for t in raw] == want
        present = sorted({t for l in lists for t in l})
        assert (tm >= 0).sum() == len(present) and all(tm[t] >= 0 for t in present)
        order = seal._cpython_set_order(lists)
        saw_unsorted |= order != sorted(order)
    assert saw_unsorted, "no case exercised a non-ascending set order"


@pytest.mark.parametrize("name,directed", [("seal_int.npz", False), ("seal_dir.npz", True)])
def test_host_cap_policy_reproduces_the_reference_stream(golden_dir, name, directed):
    """seal.host_cap_node_lists under RandomState(2) == the node sets the reference itself produced under
    np.random.seed(2) (tests/golden/*.npz, hop 1): every pair draws shuffle([u, v]) in the reference's pair order,
    neighbour lists come in networkx insertion order, the truncation keeps the shuffled list's head
    (SEAL/config/subgraph.py:118-135) -- nothing of the reference's output is replayed.  Flagged = every pair that
    could truncate in EITHER visiting order (what the device kernel reports)."""
    z = np.load(os.path.join(golden_dir, name))
    pos, neg = z["positive"].astype(np.int64), z["negative"].astype(np.int64)
    pairs = np.concatenate([neg, pos])
    off, nodes = z["hop1_off"], z["hop1_nodes"]
    train = pos[: len(pos) - int(len(pos) * 0.1)]
    nb = seal._insertion_order_neighbours(train, set(pairs.reshape(-1).tolist()), directed)
    flagged = []
    for i, (u, v) in enumerate(pairs.tolist()):
        du, dv, s0 = len(nb[u]), len(nb[v]), (1 if u == v else 2)
        eu, ev = len(set(nb[u]) - {u, v}), len(set(nb[v]) - {u, v})
        if s0 + du >= 100 or s0 + dv >= 100 or s0 + eu + dv >= 100 or s0 + ev + du >= 100:
            flagged.append(i)
    got = seal.host_cap_node_lists(pairs, flagged, train, 100, 2, directed)
    truncated = 0
    for i in flagged:
        ref = nodes[off[i]:off[i + 1]]
        assert ref[:2].tolist() == got[i][:2] and sorted(ref[2:].tolist()) == got[i][2:], i
        full = ({*nb[int(pairs[i][0])], *nb[int(pairs[i][1])]} | {int(pairs[i][0]), int(pairs[i][1])})
        truncated += len(ref) < len(full)
    if not directed:
        assert truncated >= 4                       # the undirected run has pairs the cap really bites
    wrong = seal.host_cap_node_lists(pairs, flagged, train, 100, 3, directed)      # another seed: another truncation
    if truncated:
        assert any(wrong[i] != got[i] for i in flagged)
