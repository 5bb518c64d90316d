"""`degree` exactly as a TorchScript caller would write it against the drop-in torch_scatter
(mirrors the call shape of TwoWL/utils.py:8-10)."""
import torch
from torch import Tensor

from lpgnn_amd.torch_scatter import scatter_add


@torch.jit.script
def degree(ei: Tensor, num_node: int):
    return scatter_add(torch.ones_like(ei[1]), ei[1], dim_size=num_node)
