"""drop-in: `from torch_scatter import scatter_add` (TwoWL/utils.py:5) -> HIP path."""
from lpgnn_amd.torch_scatter import scatter, scatter_add, scatter_sum  # noqa: F401
