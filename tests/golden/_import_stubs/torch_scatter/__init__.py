"""Import stub used ONLY by tests/golden/make_golden.py.

The reference module ``TwoWL/utils.py`` does ``from torch_scatter import scatter_add`` at
import time (TwoWL/utils.py:5) and scripts ``degree`` (TwoWL/utils.py:8-10) with it.
torch-scatter 2.1.1 is not installable here, so the golden generator puts this directory on
``sys.path`` purely to let the module import; the integer functions we take goldens from
(``get_ei2``, ``reverse``, ``double``, ``sample_block``, ``set_mul``, ``check_in_set``,
``idx2mask``, ``blockei2``) never call it.  ``degree`` does, so its golden is the published
torch_scatter definition restated (out = zeros(dim_size).scatter_add_(dim, index, src)) and is
labelled "restated [3P]" in the fixture -- it pins nothing beyond the definition.

Never imported by the product, by tests, or on the GPU box.
"""
from typing import Optional

import torch
from torch import Tensor


def scatter_add(src: Tensor, index: Tensor, dim: int = -1,
                out: Optional[Tensor] = None,
                dim_size: Optional[int] = None) -> Tensor:
    if out is None:
        size = list(src.size())
        if dim_size is not None:
            size[dim] = dim_size
        elif index.numel() == 0:
            size[dim] = 0
        else:
            size[dim] = int(index.max()) + 1
        res = torch.zeros(size, dtype=src.dtype, device=src.device)
        return res.scatter_add_(dim, index, src)
    else:
        return out.scatter_add_(dim, index, src)
