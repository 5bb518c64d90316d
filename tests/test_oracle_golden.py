"""CPU: pin the oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Integer paths: bit-exact."""
import os
import zlib

import numpy as np
import pytest

from oracle import seal_int, twowl_int


@pytest.fixture(scope="module")
def tw(golden_dir):
    return np.load(os.path.join(golden_dir, "twowl_int.npz"))


@pytest.fixture(scope="module")
def sl(golden_dir):
    return np.load(os.path.join(golden_dir, "seal_int.npz"))


def test_double(tw):
    assert np.array_equal(twowl_int.double(tw["split_train_pos"]), tw["double_train_pos"])
    assert np.array_equal(twowl_int.double(tw["double_idx_in"], for_index=True), tw["double_idx_out"])


@pytest.mark.parametrize("split", ["train", "val", "test"])
def test_get_ei2_reverse_degree(tw, split):
    ei, pred = tw[f"{split}_ei"], tw[f"{split}_pred"]
    ei2 = twowl_int.get_ei2(620, ei, pred)
    assert ei2.shape == tw[f"{split}_ei2"].shape
    assert np.array_equal(ei2, tw[f"{split}_ei2"])          # incl. order
    e, er = twowl_int.reverse(ei2)
    assert np.array_equal(e, tw[f"{split}_rev"]) and np.array_equal(er, tw[f"{split}_rev_r"])
    # degree golden went through the torch_scatter import stub -> definition-level only
    assert np.array_equal(twowl_int.degree(ei, 620), tw[f"{split}_degree_restated3p"])


def test_sample_block(tw):
    ei_new, x_new, ei2_new = twowl_int.sample_block(tw["sb_idx"], 620, tw["train_ei"], tw["train_ei2"])
    assert np.array_equal(ei_new, tw["sb_ei_new"])
    assert np.array_equal(x_new, tw["sb_x_new"])
    assert np.array_equal(ei2_new, tw["sb_ei2_new"])


def test_small_set_ops(tw):
    assert np.array_equal(twowl_int.set_mul(tw["set_mul_a"], tw["set_mul_b"]), tw["set_mul_out"])
    assert np.array_equal(twowl_int.check_in_set(tw["cis_t"], tw["cis_s"]), tw["cis_out"])
    assert np.array_equal(twowl_int.idx2mask(8, [1, 6, 1]), tw["idx2mask_out"])


def test_node_labeling_known_answers(sl):
    n, bits, tags = sl["kat_n"], sl["kat_adj_bits"], sl["kat_tags"]
    bo = to = 0
    for nn in n:
        nb = (nn * nn + 7) // 8
        adj = np.unpackbits(bits[bo:bo + nb])[: nn * nn].reshape(nn, nn)
        want = tags[to:to + nn]
        assert list(seal_int.node_labeling(adj)) == list(want)
        bo += nb
        to += nn
    assert list(tags[:5]) == [1, 1, 2, 4, 9]      # SURVEY section 8c known answer


@pytest.fixture(scope="module")
def sd(golden_dir):
    return np.load(os.path.join(golden_dir, "seal_dir.npz"))


def test_node_labeling_known_answers_directed(sd):
    """SEAL/utils/utils.py:35-54 with network_type = 1 (nx.DiGraph): 200 random directed graphs, tags from the reference"""
    n, bits, tags = sd["kat_n"], sd["kat_adj_bits"], sd["kat_tags"]
    bo = to = 0
    for nn in n:
        nb = (nn * nn + 7) // 8
        adj = np.unpackbits(bits[bo:bo + nb])[: nn * nn].reshape(nn, nn)
        assert list(seal_int.node_labeling(adj)) == list(tags[to:to + nn])
        bo += nb
        to += nn


@pytest.mark.parametrize("hop,directed", [(1, False), (0, False), (1, True)])
def test_enclosing_subgraphs_all_pairs(sl, sd, hop, directed):
    """all 4202 candidate pairs: node set, canonical adjacency (crc) and tags, bit-exact -- undirected (hop 1, 0) and
    the reference run with network_type = 1 (fb-pages-food read as a directed edge list, hop 1)."""
    sl = sd if directed else sl
    pos, neg = sl["positive"], sl["negative"]
    A = seal_int.build_train_adjacency(pos, int(sl["n_nodes"]), network_type=int(directed))
    assert bool((A != A.T).any()) == directed
    k = f"hop{hop}_"
    off, nodes_ref, tags_ref = sl[k + "off"], sl[k + "nodes"], sl[k + "tags"]
    pairs = np.concatenate([neg, pos])            # link2subgraph: negatives first (:74)
    raw_tags, n_capped = [], 0
    for g, (u, v) in enumerate(pairs):
        ref_nodes = nodes_ref[off[g]:off[g + 1]]
        nodes, capped = seal_int.enclosing_nodes((u, v), A, hop)
        if capped:                                # RNG-dependent in the reference: replay its set
            n_capped += 1
            perm = np.concatenate([[0, 1], np.argsort(ref_nodes[2:], kind="stable") + 2])
            nodes = list(ref_nodes[perm])
            cand = set(seal_int.neighbours(A, u)) | set(seal_int.neighbours(A, v)) | {u, v}
            assert set(ref_nodes) <= cand and len(ref_nodes) <= 100
        assert sorted(nodes[2:]) == sorted(ref_nodes[2:]) and list(nodes[:2]) == list(ref_nodes[:2])
        adj = seal_int.induced_adjacency(nodes, A)
        assert int(adj.sum()) == int(sl[k + "nnz"][g])
        assert zlib.crc32(np.packbits(adj.reshape(-1)).tobytes()) == int(sl[k + "adj_crc_canon"][g])
        raw_tags.append((nodes, seal_int.node_labeling(adj)))
    tags_c, tags_size = seal_int.compact_tags([t for _, t in raw_tags])
    assert tags_size == int(sl[k + "tags_size"])
    for g in range(len(pairs)):
        ref_nodes = nodes_ref[off[g]:off[g + 1]]
        want = dict(zip(ref_nodes[2:].tolist(), tags_ref[off[g] + 2:off[g + 1]].tolist()))
        got = dict(zip(raw_tags[g][0][2:], tags_c[g][2:]))
        assert got == want and tags_c[g][:2] == list(tags_ref[off[g]:off[g] + 2])
    if hop == 1:
        assert n_capped > 0


def test_full_adjacency_reference_order(sl):
    """replay the reference's own node order: adjacency bit-for-bit."""
    A = seal_int.build_train_adjacency(sl["positive"], int(sl["n_nodes"]))
    off, nodes_ref = sl["hop1_off"], sl["hop1_nodes"]
    bits, bo = sl["hop1_adj_bits"], 0
    for g in sl["hop1_adj_keep"]:
        nodes = nodes_ref[off[g]:off[g + 1]]
        n = len(nodes)
        nb = (n * n + 7) // 8
        want = np.unpackbits(bits[bo:bo + nb])[: n * n].reshape(n, n)
        bo += nb
        assert np.array_equal(seal_int.induced_adjacency(nodes, A), want)
