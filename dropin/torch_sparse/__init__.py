"""drop-in for the torch_sparse names the north star keeps (requirements.txt:19) -> HIP path."""
from lpgnn_amd.torch_sparse import SparseTensor, matmul, spmm  # noqa: F401
