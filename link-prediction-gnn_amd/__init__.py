"""lpgnn_amd -- MI355X (gfx950) native message-passing aggregation behind the torch_scatter /
torch_sparse / PyG-GCNConv call signatures of NguyenTrieu903/Link-Prediction-GNN.

The package directory is ``link-prediction-gnn_amd/`` (not an importable name); ``lpgnn_amd.py``
at the repo root loads it under the name ``lpgnn_amd``.
"""
from . import _lib                                   # noqa: F401
from ._lib import LpgError, load as load_library    # noqa: F401
from . import ops, torch_scatter, torch_sparse, nn, twowl, dist   # noqa: F401
from .ops import CSRGraph                            # noqa: F401

__all__ = ["ops", "torch_scatter", "torch_sparse", "nn", "twowl", "CSRGraph", "load_library", "LpgError"]
from . import seal, data, metrics                    # noqa: F401,E402
