#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the importable parts of the
reference (read-only, from /root/reference) on seeded inputs.

Run in the build container only:   python tests/golden/make_golden.py
(/root/reference does not exist on the GPU box; tests read only the committed .npz files.)

What is generated from REAL reference code:
  * seal_int.npz     SEAL/config/subgraph.py::link2subgraph / extract_subgraph (:11-141) and
                     SEAL/utils/utils.py::node_labeling (:35-54) on fb-pages-food, hop=1 and
                     hop=0, plus node_labeling on random small adjacency matrices.  Needs the
                     networkx-2.0 constructor shim below (the reference pins networkx 2.0 where
                     nx.Graph(data=adj) builds from an adjacency matrix).
  * twowl_int.npz    TwoWL/utils.py::double, get_ei2, reverse, sample_block, set_mul,
                     check_in_set, idx2mask, blockei2, random_split_edges (:13-147) on a seeded
                     fb-pages-food split.  TwoWL/utils.py imports torch_scatter at module import;
                     tests/golden/_import_stubs/torch_scatter lets the import succeed (see its
                     docstring).  `degree` goes through that stub => labelled restated, not pinned.
  * seal_weights.npz the four graph-conv weight matrices of the trained SEAL checkpoint
                     (raw little-endian fp32 slices of SEAL/model/model-1000.data-00000-of-00001
                     at the offsets decoded in SURVEY.md section 5; read as bytes, nothing
                     executed).  Data, not code.

No reference source text is copied anywhere; only input/output arrays are stored.
"""
import contextlib
import io
import os
import sys
import zlib

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def canon_perm(nodes):
    """canonical order = the two targets first, the rest ascending by node id."""
    nodes = np.asarray(nodes)
    rest = np.argsort(nodes[2:], kind="stable") + 2
    return np.concatenate([[0, 1], rest]).astype(np.int64)


def pack_adj(adj):
    return np.packbits(np.asarray(adj, dtype=np.uint8).reshape(-1))


def gen_seal():
    import networkx as nx

    class _G(nx.Graph):
        def __init__(self, incoming_graph_data=None, data=None, **attr):
            super().__init__(data if data is not None else incoming_graph_data, **attr)

    class _DG(nx.DiGraph):
        def __init__(self, incoming_graph_data=None, data=None, **attr):
            super().__init__(data if data is not None else incoming_graph_data, **attr)

    nx.Graph, nx.DiGraph = _G, _DG
    sys.path.insert(0, REF)
    from SEAL.config import subgraph as ref_sub
    from SEAL.utils import utils as ref_utils
    import pandas as pd

    # candidate pairs exactly as SEAL/config/data.py:17-27 makes them (header row eaten)
    np.random.seed(0)
    positive = pd.read_csv(os.path.join(HERE, "fb-pages-food.edges"), delimiter=",", dtype=int).to_numpy()
    G = nx.Graph()
    G.add_edges_from(positive)
    negative_all = sorted(nx.non_edges(G))          # sorted: independent of set iteration order
    np.random.shuffle(negative_all)
    negative = np.asarray(negative_all[:len(positive)])
    np.random.shuffle(positive)
    n_nodes = len(G.nodes())
    assert n_nodes == 620 and len(positive) == 2101

    out = {"positive": positive.astype(np.int32), "negative": negative.astype(np.int32),
           "n_nodes": np.int64(n_nodes)}

    for hop in (1, 0):
        np.random.seed(1 + hop)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            graphs_adj, labels, vertex_tags, node_size_list, sub_nodes, tags_size = \
                ref_sub.link2subgraph(positive.copy(), negative.copy(), n_nodes, 0.1, hop, 0, 100)
        n = np.asarray(node_size_list, dtype=np.int32)
        off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
        nodes_flat = np.concatenate([np.asarray(s, dtype=np.int32) for s in sub_nodes])
        tags_flat = np.concatenate([np.asarray(t, dtype=np.int32) for t in vertex_tags])
        nnz = np.asarray([int(np.asarray(a).sum()) for a in graphs_adj], dtype=np.int32)
        crc_canon = []
        for a, s in zip(graphs_adj, sub_nodes):
            p = canon_perm(s)
            a = np.asarray(a)[p][:, p]
            crc_canon.append(zlib.crc32(pack_adj(a).tobytes()))
        k = f"hop{hop}_"
        out[k + "n"] = n
        out[k + "off"] = off
        out[k + "nodes"] = nodes_flat            # reference order (python set iteration)
        out[k + "tags"] = tags_flat              # after the global tag compaction (:90-101)
        out[k + "tags_size"] = np.int64(tags_size)
        out[k + "nnz"] = nnz
        out[k + "adj_crc_canon"] = np.asarray(crc_canon, dtype=np.uint32)
        out[k + "labels"] = np.asarray(labels, dtype=np.uint8).reshape(-1)
        # full adjacency (reference order) for the first 256 pairs of each class
        keep = list(range(0, 256)) + list(range(len(negative), len(negative) + 256))
        out[k + "adj_keep"] = np.asarray(keep, dtype=np.int32)
        out[k + "adj_bits"] = np.concatenate([pack_adj(graphs_adj[i]) for i in keep])
        print(f"[seal] hop={hop}: {len(n)} subgraphs, tags_size={tags_size}, "
              f"n min/med/mean/max = {n.min()}/{int(np.median(n))}/{n.mean():.1f}/{n.max()}, "
              f"nnz mean {nnz.mean():.1f}")

    # hop="auto" -- record which hop the CN/AA heuristic picks (SEAL/config/subgraph.py:44-63)
    np.random.seed(3)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()), \
            np.errstate(all="ignore"):
        try:
            ref_sub.link2subgraph(positive.copy(), negative.copy(), n_nodes, 0.1, "auto", 0, 100)
        except Exception as e:      # hop==2 crashes in the reference
            print("auto raised", type(e).__name__)
    auto_hop = [ln for ln in buf.getvalue().splitlines() if ln.startswith("hop =")]
    out["auto_hop_line"] = np.asarray(auto_hop[0] if auto_hop else "")
    print("[seal] auto:", auto_hop)

    # node_labeling known answers on random small graphs (entry (0,1) already cleared, as the
    # caller does at SEAL/config/subgraph.py:137)
    rng = np.random.RandomState(7)
    kat_n, kat_adj, kat_tags = [], [], []
    fixed = [
        (5, [(0, 2), (2, 1), (1, 3), (3, 4)]),
        (4, [(0, 2), (2, 1)]),
        (3, [(0, 2)]),
        (2, []),
    ]
    for nn, edges in fixed:
        a = np.zeros((nn, nn), dtype=np.uint8)
        for i, j in edges:
            a[i, j] = a[j, i] = 1
        kat_n.append(nn); kat_adj.append(pack_adj(a)); kat_tags.append(ref_utils.node_labeling(a, 0))
    for _ in range(300):
        nn = int(rng.randint(2, 41))
        p = rng.choice([0.05, 0.1, 0.2, 0.5])
        a = (rng.rand(nn, nn) < p).astype(np.uint8)
        a = np.triu(a, 1); a = a + a.T
        a[0, 1] = a[1, 0] = 0
        t = ref_utils.node_labeling(a, 0)
        if nn == 2:
            assert list(t) == [1, 1]
        kat_n.append(nn); kat_adj.append(pack_adj(a)); kat_tags.append(t)
    out["kat_n"] = np.asarray(kat_n, dtype=np.int32)
    out["kat_adj_bits"] = np.concatenate(kat_adj)
    out["kat_tags"] = np.concatenate([np.asarray(t, dtype=np.int32) for t in kat_tags])
    print("[seal] fixed KATs:", [list(map(int, t)) for t in kat_tags[:4]])
    np.savez_compressed(os.path.join(HERE, "seal_int.npz"), **out)


def gen_seal_directed():
    """seal_dir.npz: the same reference functions with network_type = 1 (nx.DiGraph; A set in one orientation only,
    SEAL/config/subgraph.py:29-31,69; directed shortest paths in node_labeling, SEAL/utils/utils.py:38) on
    fb-pages-food read as a DIRECTED edge list, hop 1, plus node_labeling known answers on random directed graphs."""
    import networkx as nx

    class _G(nx.Graph):
        def __init__(self, incoming_graph_data=None, data=None, **attr):
            super().__init__(data if data is not None else incoming_graph_data, **attr)

    class _DG(nx.DiGraph):
        def __init__(self, incoming_graph_data=None, data=None, **attr):
            super().__init__(data if data is not None else incoming_graph_data, **attr)

    nx.Graph, nx.DiGraph = _G, _DG
    sys.path.insert(0, REF)
    from SEAL.config import subgraph as ref_sub
    from SEAL.utils import utils as ref_utils
    import pandas as pd

    # candidate pairs as SEAL/config/data.py:17-27 makes them for network_type = 1 (header row eaten, DiGraph non-edges)
    np.random.seed(0)
    positive = pd.read_csv(os.path.join(HERE, "fb-pages-food.edges"), delimiter=",", dtype=int).to_numpy()
    G = nx.DiGraph()
    G.add_edges_from(positive)
    negative_all = sorted(nx.non_edges(G))          # sorted: independent of iteration order
    np.random.shuffle(negative_all)
    negative = np.asarray(negative_all[:len(positive)])
    np.random.shuffle(positive)
    n_nodes = len(G.nodes())
    out = {"positive": positive.astype(np.int32), "negative": negative.astype(np.int32), "n_nodes": np.int64(n_nodes)}
    np.random.seed(2)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        graphs_adj, labels, vertex_tags, node_size_list, sub_nodes, tags_size = \
            ref_sub.link2subgraph(positive.copy(), negative.copy(), n_nodes, 0.1, 1, 1, 100)
    n = np.asarray(node_size_list, dtype=np.int32)
    out["hop1_n"] = n
    out["hop1_off"] = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    out["hop1_nodes"] = np.concatenate([np.asarray(s, dtype=np.int32) for s in sub_nodes])     # reference order
    out["hop1_tags"] = np.concatenate([np.asarray(t, dtype=np.int32) for t in vertex_tags])    # after compaction
    out["hop1_tags_size"] = np.int64(tags_size)
    out["hop1_nnz"] = np.asarray([int(np.asarray(a).sum()) for a in graphs_adj], dtype=np.int32)
    crc = []
    for a, s in zip(graphs_adj, sub_nodes):
        pm = canon_perm(s)
        crc.append(zlib.crc32(pack_adj(np.asarray(a)[pm][:, pm]).tobytes()))
    out["hop1_adj_crc_canon"] = np.asarray(crc, dtype=np.uint32)
    out["hop1_labels"] = np.asarray(labels, dtype=np.uint8).reshape(-1)
    print(f"[seal directed] hop=1: {len(n)} subgraphs, tags_size={tags_size}, n min/med/mean/max = "
          f"{n.min()}/{int(np.median(n))}/{n.mean():.1f}/{n.max()}, asymmetric subgraphs: "
          f"{sum(int((np.asarray(a) != np.asarray(a).T).any()) for a in graphs_adj)}")
    rng = np.random.RandomState(11)
    kat_n, kat_adj, kat_tags = [], [], []
    for _ in range(200):
        nn = int(rng.randint(2, 41))
        a = (rng.rand(nn, nn) < rng.choice([0.05, 0.1, 0.2, 0.5])).astype(np.uint8)
        a[0, 1] = a[1, 0] = 0
        kat_n.append(nn); kat_adj.append(pack_adj(a)); kat_tags.append(ref_utils.node_labeling(a, 1))
    out["kat_n"] = np.asarray(kat_n, dtype=np.int32)
    out["kat_adj_bits"] = np.concatenate(kat_adj)
    out["kat_tags"] = np.concatenate([np.asarray(t, dtype=np.int32) for t in kat_tags])
    np.savez_compressed(os.path.join(HERE, "seal_dir.npz"), **out)


def gen_twowl():
    import types
    import torch
    sys.path.insert(0, os.path.join(HERE, "_import_stubs"))
    sys.path.insert(0, REF)
    from TwoWL import utils as ref

    e = np.loadtxt(os.path.join(HERE, "fb-pages-food.edges"), delimiter=",", dtype=np.int64)
    edge_index = torch.from_numpy(e.T.copy())
    out = {}
    torch.manual_seed(0)
    data = types.SimpleNamespace(num_nodes=620, edge_index=edge_index, edge_attr=None)
    data = ref.random_split_edges(data, 0.05, 0.1)
    n = 620
    trn, val, tst = data.train_pos_edge_index, data.val_pos_edge_index, data.test_pos_edge_index
    vneg, tneg = data.val_neg_edge_index, data.test_neg_edge_index
    # train negatives: the reference uses PyG negative_sampling (absent); any non-edge sample of
    # the same size is an equally valid INPUT for the functions under test.
    cand = data.train_neg_adj_mask.nonzero(as_tuple=False).t()
    trneg = cand[:, torch.randperm(cand.shape[1])[:trn.shape[1]]]
    for k, v in dict(train_pos=trn, val_pos=val, test_pos=tst, train_neg=trneg,
                     val_neg=vneg, test_neg=tneg).items():
        out["split_" + k] = v.numpy().astype(np.int32)

    d = {k: ref.double(v) for k, v in dict(trn=trn, val=val, tst=tst, trneg=trneg,
                                           vneg=vneg, tneg=tneg).items()}
    out["double_train_pos"] = d["trn"].numpy().astype(np.int32)
    idx = torch.tensor([5, 0, 17, 3], dtype=torch.long)
    out["double_idx_in"] = idx.numpy()
    out["double_idx_out"] = ref.double(idx, for_index=True).numpy()

    # the three splits exactly as TwoWL/operators/datasets.py:44-99 assembles them
    ei = [d["trn"], d["trn"], torch.cat((d["trn"], d["val"]), dim=1)]
    pred = [d["trneg"], torch.cat((d["val"], d["vneg"]), 1), torch.cat((d["tst"], d["tneg"]), 1)]
    for s, name in enumerate(["train", "val", "test"]):
        ei2 = ref.get_ei2(n, ei[s], pred[s])
        out[f"{name}_ei"] = ei[s].numpy().astype(np.int32)
        out[f"{name}_pred"] = pred[s].numpy().astype(np.int32)
        out[f"{name}_ei2"] = ei2.numpy().astype(np.int32)
        out[f"{name}_degree_restated3p"] = ref.degree(ei[s], n).numpy()
        e2, e2r = ref.reverse(ei2)
        out[f"{name}_rev"] = e2.numpy().astype(np.int32)
        out[f"{name}_rev_r"] = e2r.numpy().astype(np.int32)
        print(f"[twowl] {name}: E1={ei[s].shape[1]} Ep={pred[s].shape[1]} E2={ei2.shape[1]}")

    # sample_block (TwoWL/utils.py:61-68) as TwoWL/model/train.py:18-32 calls it
    torch.manual_seed(1)
    bs = out["val_pred"].shape[1] // 2
    perm1 = torch.randperm(ei[0].shape[1] // 2)
    idx1 = ref.double(perm1[: bs // 2], for_index=True)
    ei2_train = torch.from_numpy(out["train_ei2"].astype(np.int64))
    ei_new, x_new, ei2_new = ref.sample_block(idx1, n, ei[0], ei2_train)
    out["sb_idx"] = idx1.numpy()
    out["sb_ei_new"] = ei_new.numpy().astype(np.int32)
    out["sb_x_new"] = x_new.numpy()
    out["sb_ei2_new"] = ei2_new.numpy().astype(np.int32)
    print(f"[twowl] sample_block: |idx|={idx1.numel()} ei_new={tuple(ei_new.shape)} ei2_new={tuple(ei2_new.shape)}")

    a = torch.tensor([3, 1, 4], dtype=torch.long); b = torch.tensor([9, 2], dtype=torch.long)
    out["set_mul_a"], out["set_mul_b"] = a.numpy(), b.numpy()
    out["set_mul_out"] = ref.set_mul(a, b).numpy()
    t = torch.tensor([0, 5, 5, 7, 2, 9], dtype=torch.long); s = torch.tensor([5, 9, 11], dtype=torch.long)
    out["cis_t"], out["cis_s"] = t.numpy(), s.numpy()
    out["cis_out"] = ref.check_in_set(t, s).numpy()
    out["idx2mask_out"] = ref.idx2mask(8, torch.tensor([1, 6, 1])).numpy()
    np.savez_compressed(os.path.join(HERE, "twowl_int.npz"), **out)


def gen_seal_weights():
    path = os.path.join(REF, "SEAL/model/model-1000.data-00000-of-00001")
    raw = np.fromfile(path, dtype=np.uint8)
    spec = {"graph_weight_1": ((106, 32), 50724), "graph_weight_2": ((32, 32), 91428),
            "graph_weight_3": ((32, 32), 103716), "graph_weight_4": ((32, 1), 116004),
            # the head (SEAL/model/gnn.py:103-137): conv1d kernels, dense layers
            "conv1d_kernel_1": ((96, 1, 16), 0), "conv1d_kernel_2": ((5, 16, 32), 18432),
            "bias_1": ((128,), 49164), "bias_2": ((2,), 50700),
            "weight_1": ((448, 128), 116132), "weight_2": ((128, 2), 804260)}
    out = {}
    for k, (shape, off) in spec.items():
        nb = int(np.prod(shape)) * 4
        w = raw[off:off + nb].view("<f4").reshape(shape).copy()
        assert np.isfinite(w).all() and 1e-4 < np.abs(w).mean() < 2.0, (k, np.abs(w).mean())
        out[k] = w
        print(f"[weights] {k}: {shape} mean|w|={np.abs(w).mean():.4f}")
    np.savez_compressed(os.path.join(HERE, "seal_weights.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["seal", "seal_dir", "twowl", "weights"]
    if "seal_dir" in which:
        gen_seal_directed()
    if "weights" in which:
        gen_seal_weights()
    if "twowl" in which:
        gen_twowl()
    if "seal" in which:
        gen_seal()
