"""The ctypes stub printed in INTEGRATION.md section 3, run as written: the reference-side binding a maintainer
would add to call the fused forward from the reference's own SpGraphAttentionLayer.forward (layers.py:141-170)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import gat_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_matches_oracle():
    import pygat_amd as pg                      # noqa: F401  (builds / checks the library)
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"## 3\..*?```python\n(.*?)```", md, flags=re.S).group(1)
    code = code.replace('"pygat_amd/libpygat_amd.so"', repr(os.path.join(ROOT, "pygat_amd", "libpygat_amd.so")))
    ns = {}
    exec(compile(code, "INTEGRATION.md#3", "exec"), ns)          # our own document, not reference code
    lib, C = ns["lib"], ns["C"]
    lib.pygat_partials_bytes.restype = C.c_size_t
    lib.pygat_last_error.restype = C.c_char_p
    N, Fin, F = 300, 12, 8
    rowptr, col = O.random_symmetric_csr(N, 6, 3, hub=(1, 200))
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(N, Fin, generator=gen)
    W = torch.randn(Fin, F, generator=gen) * 0.3
    a = torch.randn(1, 2 * F, generator=gen) * 0.3
    ref = O.sparse_head_forward(x.double(), rowptr, col, W.double(), a.double(), 0.2, True).numpy()
    dev = "cuda:0"
    Wh = (x @ W).to(dev)
    s = (Wh @ a[0, :F].to(dev)).reshape(N, 1).contiguous()
    a_pad = a.view(1, 2, F).to(dev).contiguous()
    rp = torch.as_tensor(rowptr, dtype=torch.int32, device=dev)
    rows = torch.repeat_interleave(torch.arange(N, dtype=torch.int32), torch.as_tensor(np.diff(rowptr)).long())
    edge_rc = torch.stack([rows, torch.as_tensor(col, dtype=torch.int32)], 1).contiguous().to(dev)
    out = ns["fused_attention"](Wh, s, a_pad, rp, edge_rc, 0.2, True)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-5
