"""CPU tests of the oracle itself (parity unpinned: see oracle/gat_oracle.py header).

The oracle's two formulations (dense layers.py:32-64, sparse layers.py:125-173)
must agree, and the hand-derived CSR gradients must agree with stock torch
autograd through the dense formulation in fp64.
"""
import numpy as np
import pytest
import torch

from oracle import gat_oracle as O


def _params(H, Fin, Fo, skip, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(H, Fin, Fo, generator=g, dtype=torch.float64) * (1.414 * (2.0 / (Fin + Fo)) ** 0.5)
    a = torch.randn(H, 2 * Fo, generator=g, dtype=torch.float64) * (1.414 * (2.0 / (1 + 2 * Fo)) ** 0.5)
    Sk = torch.randn(H, Fin, Fo, generator=g, dtype=torch.float64) * 0.3 if skip else None
    c = lambda t: None if t is None else t.to(dtype)
    return c(W), c(a), c(Sk)


@pytest.mark.parametrize("concat", [True, False])
@pytest.mark.parametrize("skip", [False, True])
def test_dense_equals_sparse(concat, skip):
    N, Fin, Fo, H = 60, 16, 8, 3
    rowptr, col = O.random_symmetric_csr(N, 5, 1, hub=(7, 40))
    adj = O.dense_from_csr(rowptr, col, N, torch.float64)
    W, a, Sk = _params(H, Fin, Fo, skip, 2, torch.float64)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    yd = O.level_forward(x, adj, W, a, 0.2, concat, Sk, "dense")
    ys = O.level_forward(x, (rowptr, col), W, a, 0.2, concat, Sk, "sparse")
    assert yd.shape == ((N, H * Fo) if concat else (N, Fo))
    assert torch.allclose(yd, ys, atol=1e-12, rtol=0)


@pytest.mark.parametrize("concat", [True, False])
@pytest.mark.parametrize("skip", [False, True])
@pytest.mark.parametrize("formulation", ["dense", "sparse"])
def test_csr_grads_equal_autograd_fp64(concat, skip, formulation):
    N, Fin, Fo, H = 50, 12, 8, 2
    rowptr, col = O.random_symmetric_csr(N, 4, 11, hub=(3, 30))
    W, a, Sk = _params(H, Fin, Fo, skip, 12, torch.float64)
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo if concat else Fo, dtype=torch.float64, generator=gen)
    leaves = [t.clone().requires_grad_(True) for t in (x, W, a)] + ([Sk.clone().requires_grad_(True)] if skip else [])
    graph = O.dense_from_csr(rowptr, col, N, torch.float64) if formulation == "dense" else (rowptr, col)
    y = O.level_forward(leaves[0], graph, leaves[1], leaves[2], 0.2, concat,
                        leaves[3] if skip else None, formulation)
    grads = torch.autograd.grad(y, leaves, G)
    r = O.csr_layer_fwd_bwd(x.numpy(), rowptr, col, W.numpy(), a.numpy(), 0.2, concat, G.numpy(),
                            None if not skip else Sk.numpy())
    assert np.allclose(r["out"], y.detach().numpy(), atol=1e-12)
    assert np.allclose(r["dX"], grads[0].numpy(), atol=1e-11)
    assert np.allclose(r["dW"], grads[1].numpy(), atol=1e-11)
    assert np.allclose(r["da"], grads[2].numpy(), atol=1e-11)
    if skip:
        assert np.allclose(r["dW_skip"], grads[3].numpy(), atol=1e-11)


def test_dropout_masks_dense_equals_sparse():
    """Dropout order X -> Wh -> alpha (layers.py:34,37,43 / 132,136,153) with explicit masks."""
    N, Fin, Fo, H, p = 40, 10, 8, 2, 0.6
    rowptr, col = O.random_symmetric_csr(N, 4, 5)
    E = len(col)
    adj = O.dense_from_csr(rowptr, col, N, torch.float64)
    W, a, _ = _params(H, Fin, Fo, False, 6, torch.float64)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    keep = lambda *s: (torch.rand(*s, generator=gen) >= p).double() / (1 - p)
    mx, mwh, me = keep(H, N, Fin), keep(H, N, Fo), keep(H, E)
    src = np.repeat(np.arange(N), np.diff(rowptr))
    matt = torch.zeros(H, N, N, dtype=torch.float64)
    matt[:, torch.as_tensor(src), torch.as_tensor(col.astype(np.int64))] = me
    yd = O.level_forward(x, adj, W, a, 0.2, True, None, "dense", dict(x=mx, wh=mwh, att=matt))
    ys = O.level_forward(x, (rowptr, col), W, a, 0.2, True, None, "sparse", dict(x=mx, wh=mwh, att=me))
    assert torch.allclose(yd, ys, atol=1e-12, rtol=0)


def test_gradcheck_sparse_formulation():
    N, Fin, Fo = 12, 5, 4
    rowptr, col = O.random_symmetric_csr(N, 3, 21)
    W, a, _ = _params(1, Fin, Fo, False, 22, torch.float64)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=torch.Generator().manual_seed(23))
    f = lambda x_, W_, a_: O.sparse_head_forward(x_, rowptr, col, W_, a_, 0.2, True)
    assert torch.autograd.gradcheck(f, (x.requires_grad_(), W[0].requires_grad_(), a[0].reshape(1, -1).requires_grad_()),
                                    eps=1e-6, atol=1e-6)


def test_real_topologies_fp32_vs_fp64(topologies):
    """Cora-shaped level 1 (1433 -> 8 x 8 heads, row-normalised sparse features): the oracle's
    fp32 result sits within 1e-5 of its fp64 result, so 1e-5 is a meaningful GPU tolerance."""
    rowptr, col = topologies["cora"]
    N, Fin, Fo, H = len(rowptr) - 1, 1433, 8, 8
    gen = torch.Generator().manual_seed(72)
    x = (torch.rand(N, Fin, generator=gen) < 0.013).double()
    x = x / x.sum(1, keepdim=True).clamp(min=1)
    W, a, _ = _params(H, Fin, Fo, False, 72, torch.float64)
    y64 = O.level_forward(x, (rowptr, col), W, a, 0.2, True)
    y32 = O.level_forward(x.float(), (rowptr, col), W.float(), a.float(), 0.2, True)
    assert (y64 - y32.double()).abs().max() < 1e-5


def test_dense_v2_is_a_neighbour_mean():
    """layers.py:214-219: e is [N,1] and broadcasts along rows -> uniform attention;
    so dense V2 == mean of Wh2 over the neighbours, and a / W[:Fin] get exactly-zero gradients."""
    N, Fin, Fo = 30, 6, 4
    rowptr, col = O.random_symmetric_csr(N, 4, 31)
    adj = O.dense_from_csr(rowptr, col, N, torch.float64)
    gen = torch.Generator().manual_seed(32)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    W = torch.randn(2 * Fin, Fo, dtype=torch.float64, generator=gen).requires_grad_()
    a = torch.randn(Fo, 1, dtype=torch.float64, generator=gen).requires_grad_()
    y = O.dense_head_forward_v2(x, adj, W, a, 0.2, False)
    deg = adj.sum(1, keepdim=True)
    assert torch.allclose(y, (adj @ (x @ W[Fin:])) / deg, atol=1e-12)
    y.sum().backward()
    assert a.grad.abs().max() < 1e-12 and W.grad[:Fin].abs().max() < 1e-12 and W.grad[Fin:].abs().max() > 1e-3


def test_sparse_v2_gradcheck():
    N, Fin, Fo = 10, 4, 4
    rowptr, col = O.random_symmetric_csr(N, 3, 41)
    gen = torch.Generator().manual_seed(42)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen).requires_grad_()
    W = (torch.randn(2 * Fin, Fo, dtype=torch.float64, generator=gen) * 0.5).requires_grad_()
    a = torch.randn(1, Fo, dtype=torch.float64, generator=gen).requires_grad_()
    f = lambda x_, W_, a_: O.sparse_head_forward_v2(x_, rowptr, col, W_, a_, 0.2, True)  # noqa: E731
    assert torch.autograd.gradcheck(f, (x, W, a), eps=1e-6, atol=1e-6)
