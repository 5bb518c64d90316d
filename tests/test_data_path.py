"""CPU: the host-side data path.  random_split_edges draws from the torch CPU RNG
in the reference's order, so under the same seed it must reproduce the reference's split bit for bit
(golden: tests/golden/twowl_int.npz `split_*`, produced by TwoWL/utils.py::random_split_edges)."""
import os

import numpy as np
import torch

from lpgnn_amd import data, twowl


def test_random_split_edges_matches_reference(golden_dir):
    tw = np.load(os.path.join(golden_dir, "twowl_int.npz"))
    ei = data.load_edges(os.path.join(golden_dir, "fb-pages-food.edges"))
    assert ei.shape == (2, 2102) and int(ei.max()) == 619
    torch.manual_seed(0)
    sp = data.random_split_edges(ei, 620, 0.05, 0.1)
    for k in ("train_pos", "val_pos", "test_pos", "val_neg", "test_neg"):
        assert np.array_equal(sp[k].numpy(), tw["split_" + k]), k
    assert sp["train_pos"].shape[1] == 1644 and sp["val_pos"].shape[1] == 96 and sp["test_pos"].shape[1] == 193
    assert np.array_equal(twowl.double(sp["train_pos"]).numpy(), tw["double_train_pos"])


def test_negative_sampling_contract():
    g = torch.Generator().manual_seed(3)
    n = 50
    ei = torch.randint(0, n, (2, 400), generator=g)
    ei = torch.cat([ei, torch.arange(n).repeat(2, 1)], 1)
    neg = data.negative_sampling(ei, n, 300, generator=g)
    assert neg.shape == (2, 300)
    have = set((ei[0] * n + ei[1]).tolist())
    keys = (neg[0] * n + neg[1]).tolist()
    assert len(set(keys)) == 300 and not (set(keys) & have)
    assert bool((neg[0] != neg[1]).all())                 # self loops were passed in, so none are sampled
    g2 = torch.Generator().manual_seed(3)
    torch.randint(0, n, (2, 400), generator=g2)
    assert torch.equal(data.negative_sampling(ei, n, 300, generator=g2), neg)      # seeded => reproducible


def test_seal_load_data_matches_reference(golden_dir):
    """data.seal_load_data against the candidate pairs the REFERENCE's own statements produced
    (tests/golden/make_golden.py runs SEAL/config/data.py:17-27 under np.random.seed(0), non-edges sorted):
    2101 positives (header row eaten), 2101 negatives, 620 nodes -- value for value, in order."""
    sl = np.load(os.path.join(golden_dir, "seal_int.npz"))
    pos, neg, n = data.seal_load_data(os.path.join(golden_dir, "fb-pages-food.edges"), network_type=0, seed=0)
    assert n == int(sl["n_nodes"]) == 620 and pos.shape == neg.shape == (2101, 2)
    assert np.array_equal(pos, sl["positive"]) and np.array_equal(neg, sl["negative"])
    have = {(min(a, b), max(a, b)) for a, b in pos.tolist()}
    assert all(a < b and (a, b) not in have for a, b in neg.tolist())
    # directed: ordered pairs, no edge u -> v, never u == v
    posd, negd, nd = data.seal_load_data(os.path.join(golden_dir, "fb-pages-food.edges"), network_type=1, seed=0)
    haved = set(map(tuple, posd.tolist()))
    assert nd == 620 and all(a != b and (a, b) not in haved for a, b in negd.tolist())


def test_metrics_equal_sklearn_with_ties():
    """lpgnn_amd.metrics.rank_auc / roc_curve (what twowl.test and seal.auto_hop use, TwoWL/model/train.py:66-67) against
    sklearn.metrics on scores with many ties, few levels and tiny sizes."""
    import numpy as np
    import torch
    from sklearn.metrics import roc_auc_score, roc_curve
    from lpgnn_amd import metrics
    rng = np.random.default_rng(0)
    for n, levels in [(50, 5), (400, 1000), (7, 2), (1000, 13), (386, 3)]:
        y = rng.integers(0, 2, n)
        y[0], y[1] = 0, 1
        s = rng.integers(0, levels, n) / levels
        assert abs(metrics.rank_auc(torch.from_numpy(y), torch.from_numpy(s)) - roc_auc_score(y, s)) < 1e-12
        f, t, th = metrics.roc_curve(torch.from_numpy(y), torch.from_numpy(s))
        f0, t0, th0 = roc_curve(y, s)
        assert np.allclose(f.numpy(), f0) and np.allclose(t.numpy(), t0) and np.allclose(th.numpy()[1:], th0[1:])
