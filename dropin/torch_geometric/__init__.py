"""drop-in: only `torch_geometric.nn.{GCNConv, GraphNorm}` (TwoWL/model/model.py:3) is provided."""
from . import nn  # noqa: F401
