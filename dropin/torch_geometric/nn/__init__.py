from lpgnn_amd.nn import GCNConv, GraphNorm  # noqa: F401
