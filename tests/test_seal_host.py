"""CPU: host-side logic of the SEAL scoring path that needs no kernel -- the conv1d #2 / dense fold of the head, the
per-graph hub bit matrix, the train-graph CSR.  (The kernels that consume them are held to the oracle in
tests/test_gpu_parity.py.)"""
import numpy as np
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
    hub_nodes = np.nonzero(deg > seal.TrainGraph.HUB_DEGREE)[0]
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
