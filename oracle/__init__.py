"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (numpy; plain C -- spmm_ref.c, built by oracle/Makefile -- for the fp32
same-order SpMM loop) of the reference's
aggregation hot path, each function citing the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from here, and only as the checker.  The product package (``link-prediction-gnn_amd/``,
importable as ``lpgnn_amd``) never imports it and has no CPU fallback: without the HIP
extension it raises.

Pinning status (details in DESIGN.md):
  * integer paths (extract_subgraph / node_labeling / get_ei2 / reverse / double / sample_block):
    PINNED against golden vectors produced by running the reference itself
    (tests/golden/make_golden.py -> seal_int.npz, twowl_int.npz).
  * float paths (GCNConv / GraphNorm / scatter_add / SEAL graph conv): the arithmetic lives in
    third-party wheels that are neither under /root/reference nor installable here
    (torch-scatter 2.1.1, torch-sparse 0.6.17, torch-geometric 2.3.1, tensorflow 1.14;
    requirements.txt:1,18-20) and the reference holds no tests or stored outputs for them:
    PARITY UNPINNED at that boundary -- restated from the published definitions, anchored on the
    reference call sites (TwoWL/model/model.py:68-84, SEAL/model/gnn.py:46-63).
"""
