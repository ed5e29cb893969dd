"""The C restatement (oracle/gat_oracle.c, the CPU baseline at scale) against the python oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import gat_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import c_oracle
    return c_oracle


@pytest.mark.parametrize("concat", [True, False])
def test_c_oracle_matches_python_oracle(clib, concat):
    N, Fin, F, H = 300, 24, 8, 4
    rowptr, col = O.random_symmetric_csr(N, 6, 1, hub=(2, 150))
    rng = np.random.default_rng(2)
    X = rng.standard_normal((N, Fin)); W = rng.standard_normal((H, Fin, F)) * 0.3
    a = rng.standard_normal((H, 2 * F)) * 0.3; G = rng.standard_normal((N, H * F if concat else F))
    ref = O.csr_layer_fwd_bwd(X, rowptr, col, W, a, 0.2, concat, G)
    got = clib.level(X, rowptr, col, W, a, 0.2, concat, G)
    for k in ("out", "dW", "da", "dX"):
        scale = max(1.0, np.abs(ref[k]).max())
        assert np.abs(got[k] - ref[k]).max() <= 1e-5 * scale, k


def test_c_oracle_asymmetric_pattern(clib):
    N, Fin, F, H = 80, 6, 4, 2
    rng = np.random.default_rng(3)
    dense = (rng.random((N, N)) < 0.1) | np.eye(N, dtype=bool)
    rowptr = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    col = np.nonzero(dense)[1].astype(np.int32)
    X = rng.standard_normal((N, Fin)); W = rng.standard_normal((H, Fin, F)) * 0.3
    a = rng.standard_normal((H, 2 * F)) * 0.3; G = rng.standard_normal((N, H * F))
    ref = O.csr_layer_fwd_bwd(X, rowptr, col, W, a, 0.2, True, G)
    got = clib.level(X, rowptr, col, W, a, 0.2, True, G)
    for k in ("out", "dW", "da", "dX"):
        assert np.abs(got[k] - ref[k]).max() <= 1e-5 * max(1.0, np.abs(ref[k]).max()), k


@pytest.mark.parametrize("concat", [True, False])
def test_c_oracle_f64_build_is_the_python_oracle_in_fp64(clib, concat):
    """The -DORACLE_F64 build (ground truth of the full-size GPU tests) against the python oracle run in
    fp64: same algorithm, both in double -> agreement to rounding."""
    N, Fin, F, H = 300, 24, 8, 4
    rowptr, col = O.random_symmetric_csr(N, 6, 4, hub=(7, 180))
    rng = np.random.default_rng(5)
    X = rng.standard_normal((N, Fin)); W = rng.standard_normal((H, Fin, F)) * 0.3
    a = rng.standard_normal((H, 2 * F)) * 0.3; G = rng.standard_normal((N, H * F if concat else F))
    ref = O.csr_layer_fwd_bwd(X, rowptr, col, W, a, 0.2, concat, G)
    got = clib.level(X, rowptr, col, W, a, 0.2, concat, G, dtype=np.float64)
    for k in ("out", "dW", "da", "dX"):
        assert got[k].dtype == np.float64
        assert np.abs(got[k] - ref[k]).max() <= 1e-11 * max(1.0, np.abs(ref[k]).max()), k


@pytest.mark.parametrize("concat", [True, False])
@pytest.mark.parametrize("symmetric", [True, False])
def test_c_oracle_v2_level_is_the_python_v2_oracle(clib, concat, symmetric):
    """gat_oracle_level_v2 (SpGraphAttentionLayerV2, layers.py:258-313) against sparse_head_forward_v2 of the python oracle
    run in fp64 with torch autograd for the gradients: the hand-derived per-edge backward of the C file agrees to rounding
    (fp64 build) and to fp32 accuracy (fp32 build).  This is what prices GATv2LevelFn at full size (test_gpu_fullsize)."""
    import torch
    N, Fin, F, H = 260, 12, 8, 3
    rng = np.random.default_rng(7)
    if symmetric:
        rowptr, col = O.random_symmetric_csr(N, 6, 8, hub=(5, 140))
    else:
        dense = (rng.random((N, N)) < 0.05) | np.eye(N, dtype=bool)
        rowptr = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32); col = np.nonzero(dense)[1].astype(np.int32)
    X = rng.standard_normal((N, Fin)); W = rng.standard_normal((H, 2 * Fin, F)) * 0.3
    a = rng.standard_normal((H, F)) * 0.5; G = rng.standard_normal((N, H * F if concat else F))
    xt = torch.tensor(X, requires_grad=True); Wt = torch.tensor(W, requires_grad=True); at = torch.tensor(a, requires_grad=True)
    out = O.level_forward_v2(xt, (rowptr, col), Wt, at, 0.2, concat)
    out.backward(torch.tensor(G))
    ref = dict(out=out.detach().numpy(), dW=Wt.grad.numpy(), da=at.grad.numpy(), dX=xt.grad.numpy())
    got64 = clib.level_v2(X, rowptr, col, W, a, 0.2, concat, G, dtype=np.float64)
    got32 = clib.level_v2(X, rowptr, col, W, a, 0.2, concat, G)
    for k in ("out", "dW", "da", "dX"):
        scale = max(1.0, np.abs(ref[k]).max())
        assert np.abs(got64[k] - ref[k]).max() <= 1e-11 * scale, (k, np.abs(got64[k] - ref[k]).max())
        assert np.abs(got32[k] - ref[k]).max() <= 2e-5 * scale, (k, np.abs(got32[k] - ref[k]).max())
