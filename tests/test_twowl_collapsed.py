"""CPU: the per-node form of LocalWLNet's 2-WL stage (oracle.aggregation.localwl_forward_collapsed -- what
csrc/tuple_wl.hip computes) is the SAME function as the reference's forward over the explicit tuple graph
(oracle.aggregation.localwl_forward over get_ei2 / blockei2 / reverse: TwoWL/utils.py:36-50,71-78,
TwoWL/model/model.py:68-84), in fp64, on the reference-generated fb-pages-food split and on random graphs with
self-loop tuples, duplicate edges, nodes that only appear in candidates, blocked tuples and un-doubled candidates."""
import os

import numpy as np
import pytest

from oracle import aggregation as agg, twowl_int


def rand_params(rng, C1, C2, d1, d2, maxx):
    p = {"depth1": d1, "depth2": d2, "emb.weight": rng.standard_normal((maxx + 1, C1))}
    for k in ("weight", "bias", "mean_scale"):
        p[f"emb_norm.{k}"] = rng.uniform(0.5, 1.5, C1)
    for name, L in (("conv1s", d1), ("conv2s", d2), ("conv2s_r", d2)):
        for l in range(L):
            ci = C1 if name == "conv1s" else C2
            co = C1 if (name == "conv1s" and l < L - 1) else C2
            p[f"{name}.{l}.lin.weight"] = rng.standard_normal((co, ci)) / np.sqrt(ci)
            p[f"{name}.{l}.bias"] = rng.uniform(-.3, .3, co)
            for k in ("weight", "bias", "mean_scale"):
                p[f"{name}.{l}.norm.{k}"] = rng.uniform(0.5, 1.5, co)
            p[f"{name}.{l}.act"] = True
    p["pred.weight"], p["pred.bias"] = rng.standard_normal((1, C2)), rng.standard_normal(1)
    return p


def random_case(rng, trial):
    n = int(rng.integers(3, 40))
    e, ep = int(rng.integers(1, 80)), int(rng.integers(2, 30))
    pos_edge = twowl_int.double(rng.integers(0, n, size=(2, e)))                 # self loops and duplicates included
    cand = rng.integers(0, n + 3, size=(2, ep))                                  # nodes n..n+2 only appear in candidates
    pred_edge = twowl_int.double(cand) if trial % 3 else cand[:, : (ep // 2) * 2]
    blocked = twowl_int.double(rng.choice(e, size=max(1, e // 4), replace=False), for_index=True) if trial % 2 else None
    return n + 3, pos_edge, pred_edge, blocked


@pytest.mark.parametrize("trial", range(24))
def test_collapsed_equals_explicit_random(trial):
    rng = np.random.default_rng(100 + trial)
    N, pos_edge, pred_edge, blocked = random_case(rng, trial)
    E1, P = pos_edge.shape[1], pos_edge.shape[1] + pred_edge.shape[1]
    ei2 = twowl_int.get_ei2(N, pos_edge, pred_edge)
    keep = np.ones(E1, bool)
    if blocked is not None:
        ei2 = twowl_int.blockei2(ei2, blocked)
        keep[blocked] = False
    edge1 = pos_edge[:, keep]
    x_deg = np.bincount(edge1[0], minlength=N)
    p = rand_params(rng, 12, 8, int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(x_deg.max()))
    pos1 = np.concatenate([pos_edge.T, pred_edge.T], 0)
    idx = np.concatenate([E1 + np.arange(pred_edge.shape[1]), rng.integers(0, P, size=6)])       # members and blocked tuples too
    want = agg.localwl_forward(p, x_deg, edge1, pos1, idx, ei2)
    got = agg.localwl_forward_collapsed(p, x_deg, edge1, pos_edge, pred_edge, idx, N, blocked)
    assert np.abs(want - got).max() <= 1e-10 * (np.abs(want).max() + 1e-30)


@pytest.mark.parametrize("split", ["train", "val", "test"])
def test_collapsed_equals_explicit_fb_pages_food(golden_dir, split):
    """on the reference's own tuple graphs (tests/golden/twowl_int.npz: get_ei2 outputs of the reference itself)"""
    tw = np.load(os.path.join(golden_dir, "twowl_int.npz"))
    rng = np.random.default_rng(5)
    ei, pred, ei2 = (tw[f"{split}_{k}"].astype(np.int64) for k in ("ei", "pred", "ei2"))
    x_deg = twowl_int.degree(ei, 620)
    p = rand_params(rng, 24, 16, 1, 2, int(x_deg.max()))
    pos1 = np.concatenate([ei.T, pred.T], 0)
    idx = ei.shape[1] + np.arange(pred.shape[1])
    want = agg.localwl_forward(p, x_deg, ei, pos1, idx, ei2)
    got = agg.localwl_forward_collapsed(p, x_deg, ei, ei, pred, idx, 620)
    assert np.abs(want - got).max() <= 1e-10 * np.abs(want).max()
    s = agg.tuple_structure(620, ei, pred)
    assert s["ok"] and int(s["cnt"].sum()) == ei.shape[1]
    # E2 of the explicit graph = sum over nodes of (members ending) x (tuples starting)
    starts = np.bincount(np.concatenate([ei[0], pred[0]]), minlength=620)
    assert int((s["cnt"] * starts).sum()) == ei2.shape[1]


def test_structure_rejects_other_layouts():
    rng = np.random.default_rng(0)
    und = rng.integers(0, 9, size=(2, 10))
    pos = twowl_int.double(und)
    pred = twowl_int.double(rng.integers(0, 9, size=(2, 4)))
    assert agg.tuple_structure(9, pos, pred)["ok"]
    shuffled = pos[:, rng.permutation(pos.shape[1])]
    assert not agg.tuple_structure(9, shuffled, pred)["ok"]                        # partners no longer adjacent
    assert not agg.tuple_structure(9, pos, pred, blocked=np.array([2]))["ok"]      # one direction blocked only
    assert agg.tuple_structure(9, pos, pred, blocked=np.array([2, 3]))["ok"]
    assert not agg.tuple_structure(9, pos[:, :-1], pred)["ok"]                     # odd number of graph tuples
