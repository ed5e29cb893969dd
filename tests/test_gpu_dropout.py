"""Train-mode dropout with EXPLICIT masks: HIP path vs the oracle (fp64 autograd).

The reference draws per-head masks from torch's global RNG (layers.py:34,37,43 / 132,136,153);
no other implementation can reproduce that stream, so parity is defined on given masks.
"""
import numpy as np
import pytest
import torch

from oracle import gat_oracle as O
from test_gpu_parity import close, params, pg  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,Fin,Fo,skip,concat", [(3, 10, 8, False, True), (2, 7, 5, True, False), (8, 20, 8, True, True),
                                                  (3, 9, 128, True, True), (5, 6, 100, True, False), (6, 5, 128, True, True)])   # last two: head windows
def test_dropout_explicit_masks(pg, monkeypatch, H, Fin, Fo, skip, concat):  # noqa: F811
    from pygat_amd.dropout import gat_level_dropout
    monkeypatch.setenv("PYGAT_BWD_WINDOW_BYTES", "0")   # rows > 512 floats: backward in head windows
    N, p = 70, 0.6
    rowptr, col = O.random_symmetric_csr(N, 5, 3, hub=(2, 50))
    E = len(col)
    W, a, Sk = params(H, Fin, Fo, skip, 4)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo if concat else Fo, dtype=torch.float64, generator=gen)
    keep = lambda *s: (torch.rand(*s, generator=gen) >= p).double() / (1 - p)  # noqa: E731
    mx, mwh, matt = keep(H, N, Fin), keep(H, N, Fo), keep(E, H)
    # oracle (sparse formulation, masks per head; att mask is [H,E] there)
    leaves = [t.clone().requires_grad_(True) for t in (x, W, a)] + ([Sk.clone().requires_grad_(True)] if skip else [])
    y = O.level_forward(leaves[0], (rowptr, col), leaves[1], leaves[2], 0.2, concat, leaves[3] if skip else None,
                        "sparse", dict(x=mx, wh=mwh, att=matt.t().contiguous()))
    gr = torch.autograd.grad(y, leaves, G)
    dev = "cuda:0"
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev), slot_edges=16)
    xd = x.float().to(dev).requires_grad_(True)
    Ws = [W[h].float().to(dev).requires_grad_(True) for h in range(H)]
    As = [a[h].float().to(dev).reshape(1, -1).requires_grad_(True) for h in range(H)]
    Ss = [Sk[h].float().to(dev).requires_grad_(True) for h in range(H)] if skip else None
    masks = dict(x=mx.float().to(dev), wh=mwh.float().to(dev), att=matt.float().to(dev))
    out = gat_level_dropout(xd, g, Ws, As, Ss, 0.2, concat, p, masks=masks)
    out.backward(G.float().to(dev))
    close(out, y.detach().numpy(), "out")
    close(xd.grad, gr[0].numpy(), "dX")
    close(torch.stack([w.grad for w in Ws]), gr[1].numpy(), "dW")
    close(torch.stack([w.grad.reshape(-1) for w in As]), gr[2].numpy(), "da")
    if skip:
        close(torch.stack([w.grad for w in Ss]), gr[3].numpy(), "dW_skip")


def test_dropout_statistics_and_model_train_mode(pg, topologies):  # noqa: F811
    """models.GAT in train mode with dropout 0.6 on the Cora topology: runs, is random across calls,
    keeps the expected scale, and p=0 in train mode equals eval (layers.py F.dropout semantics)."""
    rowptr, col = topologies["cora"]
    N = len(rowptr) - 1
    torch.manual_seed(0)
    model = pg.GAT([64, 8, 7], [8, 1], 2, 0.6, 0.2, pg.SpGraphAttentionLayer).cuda()
    x = torch.randn(N, 64, device="cuda")
    g = pg.CSRGraph(torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda())
    model.train()
    y1 = model(x, g); y2 = model(x, g)
    assert y1.shape == (N, 7) and torch.isfinite(y1).all() and not torch.equal(y1, y2)
    y1.sum().backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in model.parameters())
    model.eval()
    with torch.no_grad():
        ye = model(x, g)
    model0 = pg.GAT([64, 8, 7], [8, 1], 2, 0.0, 0.2, pg.SpGraphAttentionLayer).cuda()
    model0.load_state_dict(model.state_dict())
    model0.train()
    assert torch.allclose(model0(x, g), ye, atol=1e-6)
