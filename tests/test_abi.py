"""CPU: the C-ABI library loads, exports every symbol include/lpg_hip.h declares, answers the
workspace queries and rejects bad arguments -- no kernel is launched."""
import ctypes
import os
import subprocess

import pytest

import lpgnn_amd
from lpgnn_amd import _lib


def test_header_symbols_exported():
    protos = _lib.parse_header()
    assert len(protos) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/lpg_hip.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T lpg_" in ln}
    assert exported == set(protos), exported ^ set(protos)


def test_workspace_queries_and_errors():
    lib = lpgnn_amd.load_library()
    assert lib.lpg_abi_version() == 1
    assert lib.lpg_spmm_chunk_edges(1000) == 16 and lib.lpg_spmm_chunk_edges(100_000_000) == 512
    assert lib.lpg_spmm_fixup_groups(128, 4, 1) == 2 and lib.lpg_spmm_fixup_groups(32, 4, 1) == 8 and lib.lpg_spmm_fixup_groups(3, 4, 0) == 16
    assert lib.lpg_spmm_fixup_groups(256, 4, 1) == 1 and lib.lpg_spmm_fixup_groups(128, 2, 1) == 4 and lib.lpg_spmm_fixup_long_row() == 64
    T = lib.lpg_spmm_chunk_edges(5_000_000)
    assert T % 16 == 0 and lib.lpg_csr_spmm_bytes(5_000_000, 128) == 2 * (-(-5_000_000 // T)) * 128 * 4 + 256
    assert lib.lpg_csr_spmm_bytes(-1, 128) < 0
    for q in (lib.lpg_csr_plan_bytes(100, 1000), lib.lpg_graphnorm_bytes(100, 32), lib.lpg_coo_to_csr_bytes(1000, 100, 1),
              lib.lpg_get_ei2_bytes(100, 1000, 100), lib.lpg_compact_cols_bytes(1000), lib.lpg_seal_block_csr_bytes(1000)):
        assert q > 0
    rc = lib.lpg_csr_spmm_f32(0, 0, 0, 0, 0, 0, 0, 0, 5, 5, 10, 0, 0, 0, 0, -1, -1, 0, 0, 0)
    assert rc == -1 and b"bad sizes" in lib.lpg_last_error()
    rc = lib.lpg_seal_subgraphs(0, 0, 10, 0, 4, 2, 100, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 256, 0, 0, 0)
    assert rc == -1 and b"hop" in lib.lpg_last_error()
    rc = lib.lpg_linear_f32(0, 4, 0, 0, 0, 0, 0, 4, 8, 0, 4, 0)
    assert rc == -1


def test_product_path_has_no_cpu_fallback():
    """CPU tensors must raise -- the oracle is never reachable from the product package."""
    import torch
    from lpgnn_amd.torch_scatter import scatter_add
    with pytest.raises(lpgnn_amd.LpgError):
        scatter_add(torch.ones(3, dtype=torch.int64), torch.tensor([0, 1, 1]), dim_size=2)
    with pytest.raises(lpgnn_amd.LpgError):
        lpgnn_amd.CSRGraph.from_edge_index(torch.zeros(2, 3, dtype=torch.long), 4)
    pkg = os.path.dirname(_lib.LIB_PATH)
    src = os.path.dirname(pkg)
    for root, _, files in os.walk(src):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_scripted_degree_compiles():
    """TwoWL/utils.py:8-10 scripts `degree`; the drop-in scatter_add must be TorchScript-callable."""
    import torch
    from tests.scripted_degree import degree        # scripted at import
    assert isinstance(degree, torch.jit.ScriptFunction)
    assert "lpgnn::scatter_add" in str(degree.graph) or "scatter_add" in str(degree.inlined_graph)


def test_state_dict_keys_match_reference_layout():
    from lpgnn_amd.twowl import LocalWLNet
    m = LocalWLNet(10, channels_1wl=32, channels_2wl=16, depth1=2, depth2=1)
    keys = set(m.state_dict().keys())
    for k in ("emb.0.weight", "emb.1.mean_scale", "conv1s.0.modlist.0.lin.weight", "conv1s.1.modlist.0.bias",
              "conv2s.0.modlist.1.weight", "conv2s_r.0.modlist.1.mean_scale", "pred.weight", "pred.bias"):
        assert k in keys, k
