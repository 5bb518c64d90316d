"""Data path of the TwoWL pipeline: edge list -> splits -> the tensors
the hot path consumes.

    load_edges / random_split_edges / negative_sampling / do_edge_split   <-> TwoWL/utils.py:93-147,
                                                                              TwoWL/operators/datasets.py:154-206
    BaseGraph / dataset / load_dataset                                     <-> TwoWL/operators/datasets.py:9-151
    seal_load_data                                                         <-> SEAL/config/data.py:10-32

The split itself is host-side index shuffling in the reference (torch CPU RNG); it is restated with
the same sequence of ``torch.randperm`` draws, so under ``torch.manual_seed(s)`` it reproduces the
reference's split bit for bit (pinned by tests/golden/twowl_int.npz: ``split_*``).  Everything that
is computation on the split -- ``double``, ``get_ei2`` (x3), ``degree`` -- runs on the device kernels.
``negative_sampling`` is third-party in the reference (PyG, absent): restated from its contract
(uniform sample of directed non-edges without duplicates), not bit-compatible with PyG's RNG use.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import twowl


def load_edges(path: str) -> torch.Tensor:
    """``src,dst`` per line, no header (TwoWL/operators/datasets.py:156-163) -> int64 [2, E]."""
    e = np.loadtxt(path, delimiter=",", dtype=np.int64).reshape(-1, 2)
    return torch.from_numpy(e.T.copy())


def seal_load_data(path: str, network_type: int = 0, seed: Optional[int] = 0):
    """SEAL/config/data.py:10-32 -- the SEAL pipeline's candidate loader: (positive [P, 2], negative [P, 2], n_nodes).

    * positives: the edge file read the way ``pd.read_csv(path, delimiter=',', dtype=int)`` reads it, i.e. with the
      FIRST line taken as the column header (:17) -- fb-pages-food's 2102 lines give 2101 positives;
    * negatives: ``list(nx.non_edges(G))`` shuffled, the first P kept (:21-26).  networkx enumerates the non-edges by
      popping nodes from a Python set; the list is SORTED here before the shuffle, exactly as
      tests/golden/make_golden.py does when it runs the reference, so that the result does not depend on set iteration
      order (for ids below the set's slot count the reference's own order is already this one): pairs (u, v) with
      u < v that are not adjacent (undirected), or every ordered pair u != v with no edge u -> v (directed);
    * the two ``np.random.shuffle`` calls (negatives first, then positives, :24,27) are drawn from
      ``np.random.RandomState(seed)`` -- the stream ``np.random.seed(seed)`` gives the reference; seed=None uses the
      global numpy RNG like the reference itself;
    * 1-based ids are shifted to 0-based (:29-31).  n_nodes = number of distinct node ids (``len(G.nodes())``).
    Pinned by tests/golden/seal_int.npz ``positive`` / ``negative`` (reference-generated, seed 0)."""
    positive = np.loadtxt(path, delimiter=",", dtype=np.int64, skiprows=1).reshape(-1, 2)
    nodes = np.unique(positive)
    n = nodes.size
    rank = np.searchsorted(nodes, positive)                       # ids -> 0..n-1 positions (identity for 0..n-1 files)
    adj = np.zeros((n, n), dtype=bool)
    adj[rank[:, 0], rank[:, 1]] = True
    if network_type == 0:
        adj |= adj.T
        free = np.triu(~adj, k=1)                                 # u < v, not adjacent (self pairs never listed)
    else:
        free = ~adj
        np.fill_diagonal(free, False)                             # non_neighbors excludes the node itself
    cu, cv = np.nonzero(free)                                     # row-major == sorted lexicographically
    negative_all = np.stack([nodes[cu], nodes[cv]], axis=1)
    rng = np.random.RandomState(seed) if seed is not None else np.random
    rng.shuffle(negative_all)
    negative = negative_all[: len(positive)].copy()
    rng.shuffle(positive)
    if positive.size and positive.min() == 1:
        positive -= 1
        negative -= 1
    return positive, negative, int(n)


def random_split_edges(edge_index: torch.Tensor, num_nodes: int, val_ratio: float = 0.05,
                       test_ratio: float = 0.1) -> Dict[str, torch.Tensor]:
    """TwoWL/utils.py:93-147.  Keeps the ``row < col`` half, permutes it once (val | test | train),
    and draws val/test negatives from the strict upper triangle of non-edges with a second
    permutation.  Uses the global torch CPU RNG exactly like the reference (seed with
    torch.manual_seed).  Dense [N, N] mask as in the reference: meant for graphs of this size."""
    row, col = edge_index[0].cpu(), edge_index[1].cpu()
    upper = row < col
    row, col = row[upper], col[upper]
    n_v = int(math.floor(val_ratio * row.numel()))
    n_t = int(math.floor(test_ratio * row.numel()))
    order = torch.randperm(row.numel())
    row, col = row[order], col[order]
    out = {
        "val_pos": torch.stack([row[:n_v], col[:n_v]]),
        "test_pos": torch.stack([row[n_v:n_v + n_t], col[n_v:n_v + n_t]]),
        "train_pos": torch.stack([row[n_v + n_t:], col[n_v + n_t:]]),
    }
    free = torch.ones(num_nodes, num_nodes, dtype=torch.uint8).triu(diagonal=1).to(torch.bool)
    free[row, col] = False
    cand_r, cand_c = free.nonzero(as_tuple=False).t()
    pick = torch.randperm(cand_r.numel())[:n_v + n_t]
    cand_r, cand_c = cand_r[pick], cand_c[pick]
    free[cand_r, cand_c] = False
    out["train_neg_adj_mask"] = free
    out["val_neg"] = torch.stack([cand_r[:n_v], cand_c[:n_v]])
    out["test_neg"] = torch.stack([cand_r[n_v:n_v + n_t], cand_c[n_v:n_v + n_t]])
    return out


def negative_sampling(edge_index: torch.Tensor, num_nodes: int, num_neg_samples: int,
                      generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """contract of torch_geometric.utils.negative_sampling [3P, PyG 2.3.1] as called at
    TwoWL/operators/datasets.py:176-197: ``num_neg_samples`` distinct directed pairs (i, j) that are
    not columns of ``edge_index`` (the callers pass the self loops in, so i == j is excluded there)."""
    dev = edge_index.device
    have = torch.unique(edge_index[0] * num_nodes + edge_index[1])
    total = num_nodes * num_nodes
    if num_neg_samples > total - have.numel():
        raise ValueError("negative_sampling: not enough non-edges")
    chosen = torch.empty(0, dtype=torch.int64, device=dev)
    while chosen.numel() < num_neg_samples:
        cand = torch.randint(0, total, (int((num_neg_samples - chosen.numel()) * 1.3) + 32,), generator=generator,
                             device=dev)
        if have.numel():
            pos = torch.searchsorted(have, cand).clamp(max=have.numel() - 1)
            cand = cand[have[pos] != cand]
        chosen = torch.unique(torch.cat([chosen, cand]))          # sorted: deterministic given the generator
    keep = torch.randperm(chosen.numel(), generator=generator, device=dev)[:num_neg_samples]
    got = chosen[keep]
    return torch.stack([torch.div(got, num_nodes, rounding_mode="floor"), got % num_nodes])


def do_edge_split(edge_index, num_nodes, val_ratio=0.05, test_ratio=0.1, generator=None):
    """TwoWL/operators/datasets.py:171-206: the positive split of random_split_edges, negatives
    re-drawn with negative_sampling against (train + self loops [+ val [+ test]])."""
    sp = random_split_edges(edge_index, num_nodes, val_ratio, test_ratio)
    loops = torch.arange(num_nodes).repeat(2, 1)
    base = torch.cat([sp["train_pos"], loops], dim=1)                    # add_self_loops(train_pos)
    n_tr, n_v, n_t = sp["train_pos"].shape[1], sp["val_pos"].shape[1], sp["test_pos"].shape[1]
    return {
        "train": {"edge": sp["train_pos"], "edge_neg": negative_sampling(base, num_nodes, n_tr, generator)},
        "valid": {"edge": sp["val_pos"],
                  "edge_neg": negative_sampling(torch.cat([base, sp["val_pos"]], 1), num_nodes, n_v, generator)},
        "test": {"edge": sp["test_pos"],
                 "edge_neg": negative_sampling(torch.cat([base, sp["val_pos"], sp["test_pos"]], 1), num_nodes, n_t,
                                               generator)},
    }


class dataset:
    """TwoWL/operators/datasets.py:9-17 (same field names)."""

    def __init__(self, x, na, ei, ea, pos1, y, ei2):
        self.x, self.na, self.ei, self.ea, self.pos1, self.y, self.ei2 = x, na, ei, ea, pos1, y, ei2


class BaseGraph:
    """TwoWL/operators/datasets.py:20-120: assembles, per split (0 train / 1 val / 2 test), the
    message-passing graph ``ei``, the tuple list ``pos1`` (graph edges then candidates), labels
    ``y`` and the 2-WL tuple graph ``ei2`` -- on the device."""

    def __init__(self, x, node_attr, edge_pos, edge_neg, num_pos, num_neg, pattern="2wl_l"):
        """pattern: "2wl_l" (the reference's: explicit ei2), "2wl_l_implicit" (structured tuple graph, same scores)"""
        self.x, self.node_attr = x, node_attr
        self.edge_pos, self.edge_neg = edge_pos, edge_neg
        self.num_pos, self.num_neg = [int(v) for v in num_pos], [int(v) for v in num_neg]
        self.num_nodes = x.shape[0]
        self.max_x, self.pattern = None, pattern

    def preprocess(self):
        p, q = self.num_pos, self.num_neg
        ep, en = self.edge_pos, self.edge_neg
        self.edge_indexs = [ep[:, :p[0]], ep[:, :p[0]], ep[:, :p[0] + p[1]]]            # :45-49
        self.edge_attrs = [torch.ones_like(e[0], dtype=torch.float) for e in self.edge_indexs]
        pos = [ep[:, :p[0]], ep[:, p[0]:p[0] + p[1]], ep[:, ep.shape[1] - p[2]:]]
        neg = [en[:, :q[0]], en[:, q[0]:q[0] + q[1]], en[:, en.shape[1] - q[2]:]]
        pred = [neg[0]] + [torch.cat((pos[i], neg[i]), dim=1) for i in (1, 2)]           # :75-78
        self.pred_edges = pred
        self.pos1s = [torch.cat((self.edge_indexs[i].t(), pred[i].t()), dim=0).contiguous() for i in range(3)]
        dev = ep.device
        self.ys = [torch.zeros((neg[0].shape[1], 1), device=dev)] + [
            torch.cat((torch.ones((pos[i].shape[1], 1), device=dev), torch.zeros((neg[i].shape[1], 1), device=dev)))
            for i in (1, 2)]
        if self.pattern == "2wl_l":                                                       # :94-99, a3 on the device
            self.ei2s = [twowl.get_ei2(self.num_nodes, self.edge_indexs[i].contiguous(), pred[i].contiguous())
                         for i in range(3)]
        elif self.pattern == "2wl_l_implicit":
            # the same tuple graph as a structured handle (twowl.TupleGraph): nothing of size E2 = sum deg^2 is built,
            # LocalWLNet / reverse / sample_block take it where the reference passes ei2; pos1 is the handle's own list
            self.ei2s = [twowl.TupleGraph(self.num_nodes, self.edge_indexs[i].contiguous(), pred[i].contiguous())
                         for i in range(3)]
            self.pos1s = [g.pos1 for g in self.ei2s]
        else:
            self.ei2s = [None] * 3

    def setPosDegreeFeature(self):
        """:107-114 -- x[0] = x[1] = degree(train graph), x[2] = degree(edge_indexs[1]) (sic)."""
        d0 = twowl.degree(self.edge_indexs[0].contiguous(), self.num_nodes)
        d1 = twowl.degree(self.edge_indexs[1].contiguous(), self.num_nodes)
        self.x = [d0, d0.clone(), d1]
        self.max_x = max(int(t.max().item()) for t in self.x)

    def split(self, split: int):
        return (self.x[split], self.node_attr, self.edge_indexs[split], self.edge_attrs[split], self.pos1s[split],
                self.ys[split], self.ei2s[split])

    def to(self, device):
        self.edge_pos, self.edge_neg = self.edge_pos.to(device), self.edge_neg.to(device)
        return self


def load_dataset(path: str, pattern="2wl_l", val_ratio=0.05, test_ratio=0.1, device="cuda", generator=None):
    """TwoWL/operators/datasets.py:123-151: edge list -> split -> doubled (both directions at 2k, 2k+1)
    positive / negative tuples -> BaseGraph on ``device``."""
    edge_index = load_edges(path)
    n = int(edge_index.max()) + 1
    sp = do_edge_split(edge_index, n, val_ratio, test_ratio, generator)
    d = {k: {kk: twowl.double(vv) for kk, vv in v.items()} for k, v in sp.items()}
    edge_pos = torch.cat((d["train"]["edge"], d["valid"]["edge"], d["test"]["edge"]), dim=-1).to(device)
    edge_neg = torch.cat((d["train"]["edge_neg"], d["valid"]["edge_neg"], d["test"]["edge_neg"]), dim=-1).to(device)
    num_pos = [d[k]["edge"].shape[1] for k in ("train", "valid", "test")]
    num_neg = [d[k]["edge_neg"].shape[1] for k in ("train", "valid", "test")]
    n_node = int(max(edge_pos.max(), edge_neg.max())) + 1
    return BaseGraph(torch.zeros((n_node, 0)), None, edge_pos, edge_neg, num_pos, num_neg, pattern)
