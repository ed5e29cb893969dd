"""Sparse input features (pygat_amd/features.py, csrc/k9_sparse.hip): the first level's projection and weight gradient on
the non-zeros of a bag-of-words X.  Checked against fp64 products under tests/parity.py's rule, against the dense HIP
kernels under the SAME dropout decisions (the two paths may differ in summation order only), and end to end through the
model on the Cora topology."""
import numpy as np
import pytest
import torch

from parity import close_grad

pytestmark = pytest.mark.gpu


def _features(n, fin, density, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, fin, generator=g) < density).float() * (torch.rand(n, fin, generator=g) + 0.1)
    x[3] = 0                                         # an empty row
    x[11] = torch.rand(fin, generator=g) + 0.1       # a full row: several 64-pair batches of the walk
    x[:, 5] = 0                                      # an empty column
    return x / x.sum(1, keepdim=True).clamp(min=1e-6)


@pytest.mark.parametrize("n,fin,H,Fo,skip,p", [(2708, 1433, 8, 8, False, 0.0), (2708, 1433, 8, 8, False, 0.6), (1000, 500, 8, 3, True, 0.5),
                                                (777, 300, 4, 64, True, 0.0), (500, 129, 1, 7, False, 0.3), (300, 50, 4, 121, True, 0.0)])
def test_sparse_projection_and_weight_gradient(n, fin, H, Fo, skip, p):
    import pygat_amd as pg
    from pygat_amd._lib import lib, check
    from pygat_amd.features import SparseFeatures
    dev = "cuda"
    Fp = pg.padded_width(Fo); R = H * Fp
    if R * (2 if skip else 1) + H > 512:
        pytest.skip("more output columns than the sparse kernels take")
    x = _features(n, fin, 0.02, n + fin)
    x[:, 7] = torch.rand(n) + 0.1                    # a column every row has: several segments
    x = x / x.sum(1, keepdim=True).clamp(min=1e-6)
    g = torch.Generator().manual_seed(H + Fo)
    W = torch.randn(H, fin, Fo, generator=g) * 0.3; a = torch.randn(H, 2 * Fo, generator=g) * 0.3
    Ws = torch.randn(H, fin, Fo, generator=g) * 0.3 if skip else None
    keep = 1.0 - p
    M = (torch.rand(H, n, fin, generator=g) < keep) if p > 0 and H <= 8 else None      # explicit decisions
    bits = None
    if M is not None:
        bits = sum((M[h].to(torch.int32) << h) for h in range(H)).to(torch.uint8).to(dev).contiguous()
    xd = x.to(dev); xs = SparseFeatures(xd)
    assert xs.nnz == int((x != 0).sum()) and abs(xs.density - 0.02) < 0.02 and xs.nseg > fin
    ldw = -(-(R * (2 if skip else 1) + 2 * H) // 4) * 4
    Wcat = torch.empty(fin, ldw, device=dev); a_pad = torch.empty(H, 2, Fp, device=dev)
    Wd, ad, Wsd = W.to(dev).contiguous(), a.to(dev).contiguous(), (Ws.to(dev).contiguous() if skip else None)   # (kept alive)
    check(lib.pygat_pack_params(H, fin, Fo, Wd.data_ptr(), ad.data_ptr(), Wsd.data_ptr() if skip else None, Wcat.data_ptr(), ldw,
                                a_pad.data_ptr(), None))
    Wh = torch.full((n, R), float("nan"), device=dev); Sk = torch.full((n, R), float("nan"), device=dev) if skip else None
    s = torch.full((n, H), float("nan"), device=dev) if p == 0 else None
    pe = p if M is not None else 0.0
    check(lib.pygat_project_sparse(n, fin, H, Fo, xs.rowptr.data_ptr(), xs.col.data_ptr(), xs.val.data_ptr(), Wcat.data_ptr(), ldw,
                                   pe, None, 0, bits.data_ptr() if bits is not None else None, Wh.data_ptr(),
                                   Sk.data_ptr() if skip else None, s.data_ptr() if s is not None else None, None), "project_sparse")
    torch.cuda.synchronize()
    scale = 1.0 / keep if M is not None else 1.0
    for h in range(H):
        xm = (x * M[h] if M is not None else x) * scale
        close_grad(Wh.view(n, H, Fp)[:, h, :Fo], (xm.double() @ W[h].double()).numpy(), (xm @ W[h]).double().numpy(), f"Wh head {h}")
        assert Fp == Fo or float(Wh.view(n, H, Fp)[:, h, Fo:].abs().max()) == 0.0
        if skip:
            close_grad(Sk.view(n, H, Fp)[:, h, :Fo], (xm.double() @ Ws[h].double()).numpy(), (xm @ Ws[h]).double().numpy(), f"Sk head {h}")
        if s is not None:
            ref = x.double() @ (W[h].double() @ a[h, :Fo].double())
            close_grad(s[:, h], ref.numpy(), (x @ (W[h] @ a[h, :Fo])).double().numpy(), f"s head {h}")
    # weight gradient
    dWh = torch.zeros(n, H, Fp); dWh[:, :, :Fo] = torch.randn(n, H, Fo, generator=g)
    RW = R + 4 * H
    GR = torch.randn(n, RW, generator=g)
    dW = torch.full((H, fin, Fo), float("nan"), device=dev); dWs = torch.full((H, fin, Fo), float("nan"), device=dev) if skip else None
    dWh_d, GR_d = dWh.view(n, R).to(dev).contiguous(), GR.to(dev).contiguous()
    wss = torch.empty(lib.pygat_wgrad_sparse_workspace_bytes(xs.nseg, H, Fo, int(skip)) // 4 + 4, device=dev)
    check(lib.pygat_wgrad_sparse(n, fin, H, Fo, xs.nseg, xs.colseg.data_ptr(), xs.seg_col.data_ptr(), xs.seg_begin.data_ptr(),
                                 xs.seg_end.data_ptr(), xs.trow.data_ptr(), xs.tval.data_ptr(), pe, None, 0,
                                 bits.data_ptr() if bits is not None else None, dWh_d.data_ptr(),
                                 GR_d.data_ptr() if skip else None, RW, wss.data_ptr(), dW.data_ptr(),
                                 dWs.data_ptr() if skip else None, None), "wgrad_sparse")
    torch.cuda.synchronize()
    for h in range(H):
        xm = (x * M[h] if M is not None else x) * scale
        d = dWh[:, h, :Fo]
        close_grad(dW[h], (xm.double().t() @ d.double()).numpy(), (xm.t() @ d).double().numpy(), f"dW head {h}")
        if skip:
            gph = GR[:, h * Fp:h * Fp + Fo]
            close_grad(dWs[h], (xm.double().t() @ gph.double()).numpy(), (xm.t() @ gph).double().numpy(), f"dWskip head {h}")


@pytest.mark.parametrize("dropout", [0.0, 0.6])
def test_model_on_sparse_features_equals_the_dense_path(topologies, monkeypatch, dropout):
    """The Cora-shaped model, training step and eval forward, with the first level on the sparse kernels and on the dense
    ones: same in-kernel dropout decisions (same seed), so loss, logits and every gradient agree to rounding."""
    import pygat_amd as pg
    from pygat_amd import features
    rowptr, col = topologies["cora"]
    N = len(rowptr) - 1
    x = _features(N, 1433, 0.013, 1).cuda()
    y = torch.randint(0, 7, (N,), generator=torch.Generator().manual_seed(2)).cuda()
    it = torch.arange(140).cuda()
    graph = pg.CSRGraph(torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda())
    crit = pg.EluLogSoftmaxNLL(it, y, N)
    res = []
    for sparse in (True, False):
        monkeypatch.setattr(features, "ENABLED", sparse)
        features._cache.clear()
        torch.manual_seed(11)
        model = pg.GAT([1433, 8, 7], [8, 1], 2, dropout, 0.2, pg.SpGraphAttentionLayer).cuda().train()
        torch.manual_seed(12)                          # the levels draw their mask seeds from torch's generator
        loss = crit(model(x, graph))
        loss.backward()
        with torch.no_grad():
            logits = model.eval()(x, graph)
        res.append((float(loss), [p.grad.clone() for p in model.parameters()], logits))
        assert (features.as_sparse_features(x, 72) is not None) == sparse
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * max(1.0, abs(res[1][0]))
    for gs, gd in zip(res[0][1], res[1][1]):
        assert float((gs - gd).abs().max()) <= 2e-6 * max(1.0, float(gd.abs().max()))
    assert float((res[0][2] - res[1][2]).abs().max()) <= 2e-6
