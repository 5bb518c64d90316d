"""CPU, build container only: the reference's own TwoWL files import and construct against the
drop-in packages (skipped where /root/reference is absent, e.g. on the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CODE = r'''
import sys, torch
sys.dont_write_bytecode = True
from TwoWL import utils as U                       # imports torch_scatter -> dropin
from TwoWL.model.model import LocalWLNet as RefNet   # imports torch_geometric.nn -> dropin
import torch_scatter, torch_geometric.nn as tgnn, lpgnn_amd
assert torch_scatter.scatter_add is lpgnn_amd.torch_scatter.scatter_add
assert tgnn.GCNConv is lpgnn_amd.nn.GCNConv and tgnn.GraphNorm is lpgnn_amd.nn.GraphNorm
assert isinstance(U.degree, torch.jit.ScriptFunction)     # TwoWL/utils.py:8 scripts through our op
ref = RefNet(12, False, None, channels_1wl=24, channels_2wl=16, depth1=2, depth2=2)
ours = lpgnn_amd.twowl.LocalWLNet(12, channels_1wl=24, channels_2wl=16, depth1=2, depth2=2)
a, b = ref.state_dict(), ours.state_dict()
assert list(a.keys()) == list(b.keys()), set(a) ^ set(b)
assert all(a[k].shape == b[k].shape for k in a)
ours.load_state_dict(a)                             # reference weights load unchanged
# integer helpers that need no device agree with the oracle-pinned goldens' generator
e = torch.tensor([[0, 2], [1, 3]])
assert U.double(e).tolist() == lpgnn_amd.twowl.double(e).tolist()
print("DROPIN_OK")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_files_run_on_dropins():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "dropin"), REF]),
               PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, r.stderr[-2000:]
