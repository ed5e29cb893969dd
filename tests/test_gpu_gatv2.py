"""GATv2 layers (reference layers.py:179-316) on the HIP path vs the oracle (fp64 autograd)."""
import numpy as np
import pytest
import torch

from oracle import gat_oracle as O
from parity import check_autograd, close_fwd
from test_gpu_parity import pg  # noqa: F401

pytestmark = pytest.mark.gpu


def v2params(H, Fin, Fo, skip, seed):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(H, 2 * Fin, Fo, generator=g, dtype=torch.float64) * (1.414 * (2.0 / (2 * Fin + Fo)) ** 0.5)
    a = torch.randn(H, Fo, generator=g, dtype=torch.float64) * (1.414 * (2.0 / (1 + Fo)) ** 0.5)
    Sk = torch.randn(H, Fin, Fo, generator=g, dtype=torch.float64) * 0.3 if skip else None
    return W, a, Sk


@pytest.mark.parametrize("H,Fin,Fo,skip,concat", [(8, 16, 8, False, True), (1, 12, 7, False, False), (4, 10, 16, True, True),
                                                 (2, 20, 64, True, False), (3, 9, 4, False, True), (4, 8, 128, False, True)])
@pytest.mark.parametrize("slot", [64, 8])
@pytest.mark.parametrize("symmetric", [True, False])
def test_sparse_v2_level(pg, H, Fin, Fo, skip, concat, slot, symmetric):  # noqa: F811
    N = 90
    if symmetric:
        rowptr, col = O.random_symmetric_csr(N, 5, 7 + H, hub=(4, 60))
    else:
        rng = np.random.default_rng(H)
        dense = (rng.random((N, N)) < 0.06) | np.eye(N, dtype=bool)
        rowptr = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
        col = np.nonzero(dense)[1].astype(np.int32)
    W, a, Sk = v2params(H, Fin, Fo, skip, 3 + Fo)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo if concat else Fo, dtype=torch.float64, generator=gen)
    leaves = [x, W, a] + ([Sk] if skip else [])
    oracle = lambda *lv: O.level_forward_v2(lv[0], (rowptr, col), lv[1], lv[2], 0.2, concat, lv[3] if skip else None)  # noqa: E731
    dev = "cuda:0"
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev), slot_edges=slot)
    xd = x.float().to(dev).requires_grad_(True)
    Wd = W.float().to(dev).requires_grad_(True)
    ad = a.float().to(dev).requires_grad_(True)
    Sd = Sk.float().to(dev).requires_grad_(True) if skip else None
    out = pg.GATv2LevelFn.apply(xd, Wd, ad, Sd, g, 0.2, concat)
    out.backward(G.float().to(dev))
    check_autograd(out, [xd.grad, Wd.grad, ad.grad] + ([Sd.grad] if skip else []), oracle, leaves, G,
                   ["dX", "dW", "da", "dW_skip"], what=f"v2[{H},{Fin},{Fo},{skip},{concat},slot {slot},sym {symmetric}]")


def test_v2_dropin_layers_and_model(pg, topologies):  # noqa: F811
    rowptr, col = topologies["cora"]
    N, Fin, Fo = len(rowptr) - 1, 32, 8
    adj = O.dense_from_csr(rowptr, col, N)
    x = torch.randn(N, Fin, generator=torch.Generator().manual_seed(1))
    # dense V2: the reference's row-broadcast logits = neighbour mean of h W[Fin:]
    torch.manual_seed(2)
    d = pg.GraphAttentionLayerV2(Fin, Fo, 0.0, 0.2, concat=True, skip_connection=True).cuda()
    yd = d(x.cuda(), adj.cuda())
    ref = O.dense_head_forward_v2(x.double(), adj.double(), d.W.detach().double().cpu(), d.a.detach().double().cpu(), 0.2,
                                  True, d.skip_projection.detach().double().cpu())
    ref32 = O.dense_head_forward_v2(x, adj, d.W.detach().cpu(), d.a.detach().cpu(), 0.2, True, d.skip_projection.detach().cpu())
    close_fwd(yd, ref.numpy(), "dense V2", ref32.double().numpy())
    yd.sum().backward()
    assert d.a.grad is not None and float(d.a.grad.abs().max()) == 0.0
    assert float(d.W.grad[:Fin].abs().max()) == 0.0 and float(d.W.grad[Fin:].abs().max()) > 0.0
    # sparse V2 single layer
    torch.manual_seed(3)
    s = pg.SpGraphAttentionLayerV2(Fin, Fo, 0.0, 0.2, concat=False).cuda()
    ys = s(x.cuda(), adj.cuda())
    ref = O.sparse_head_forward_v2(x.double(), rowptr, col, s.W.detach().double().cpu(), s.a.detach().double().cpu(), 0.2, False)
    ref32 = O.sparse_head_forward_v2(x, rowptr, col, s.W.detach().cpu(), s.a.detach().cpu(), 0.2, False)
    close_fwd(ys, ref.numpy(), "sparse V2", ref32.double().numpy())
    # model: --model GATv2_sparse (train.py:117), 8 heads then 1
    torch.manual_seed(4)
    m = pg.GAT([Fin, 8, 7], [8, 1], 2, 0.0, 0.2, pg.SpGraphAttentionLayerV2).cuda()
    sd = m.state_dict()
    assert tuple(sd["attention_layer_1_head_1.W"].shape) == (2 * Fin, 8) and tuple(sd["attention_layer_1_head_1.a"].shape) == (1, 8)
    y = m(x.cuda(), adj.cuda())
    h = O.level_forward_v2(x.double(), (rowptr, col),
                           torch.stack([sd[f"attention_layer_1_head_{j}.W"].double().cpu() for j in range(1, 9)]),
                           torch.stack([sd[f"attention_layer_1_head_{j}.a"].double().cpu().reshape(-1) for j in range(1, 9)]),
                           0.2, True)
    ref = O.level_forward_v2(h, (rowptr, col), sd["attention_layer_2_head_1.W"].double().cpu()[None],
                             sd["attention_layer_2_head_1.a"].double().cpu().reshape(1, -1), 0.2, False)
    h32 = O.level_forward_v2(x, (rowptr, col), torch.stack([sd[f"attention_layer_1_head_{j}.W"].cpu() for j in range(1, 9)]),
                             torch.stack([sd[f"attention_layer_1_head_{j}.a"].cpu().reshape(-1) for j in range(1, 9)]), 0.2, True)
    ref32 = O.level_forward_v2(h32, (rowptr, col), sd["attention_layer_2_head_1.W"].cpu()[None],
                               sd["attention_layer_2_head_1.a"].cpu().reshape(1, -1), 0.2, False)
    close_fwd(y, ref.numpy(), "GATv2_sparse model", ref32.double().numpy())
    y.sum().backward()
    assert all(p.grad is not None for p in m.parameters())
    # dense V2 in train mode (train.py:54,116 default for --model GATv2): runs (parity: test_dense_v2_dropout_explicit_masks)
    yd = pg.GraphAttentionLayerV2(Fin, Fo, 0.5, 0.2).cuda().train()(x.cuda(), adj.cuda())
    assert torch.isfinite(yd).all()
    # train-mode dropout of the sparse V2 layer: runs, differs between calls, has gradients
    m.dropout = 0.6
    for lay in m.modules():
        if hasattr(lay, "dropout"):
            lay.dropout = 0.6
    m.train()
    y1, y2 = m(x.cuda(), adj.cuda()), m(x.cuda(), adj.cuda())
    assert torch.isfinite(y1).all() and not torch.equal(y1, y2)
    y1.sum().backward()


@pytest.mark.parametrize("H,Fin,Fo,skip,concat", [(3, 10, 8, False, True), (2, 7, 5, True, False)])
def test_sparse_v2_dropout_explicit_masks(pg, H, Fin, Fo, skip, concat):  # noqa: F811
    """layers.py:266,271-272,293 with given masks: HIP path vs fp64 autograd through the oracle."""
    from pygat_amd.gatv2 import gatv2_level
    N, p = 60, 0.5
    rowptr, col = O.random_symmetric_csr(N, 5, 3, hub=(2, 40))
    E = len(col)
    W, a, Sk = v2params(H, Fin, Fo, skip, 4)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(N, Fin, dtype=torch.float64, generator=gen)
    G = torch.randn(N, H * Fo if concat else Fo, dtype=torch.float64, generator=gen)
    keep = lambda *s: (torch.rand(*s, generator=gen) >= p).double() / (1 - p)  # noqa: E731
    mx, mi, mj, matt = keep(H, N, Fin), keep(H, N, Fo), keep(H, N, Fo), keep(E, H)
    leaves = [x, W, a] + ([Sk] if skip else [])

    def oracle(*lv):
        c = lambda m: m.to(lv[0].dtype)  # noqa: E731
        outs = [O.sparse_head_forward_v2(lv[0], rowptr, col, lv[1][h], lv[2][h], 0.2, concat, lv[3][h] if skip else None,
                                         c(mx[h]), c(mi[h]), c(mj[h]), c(matt[:, h])) for h in range(H)]
        return torch.cat(outs, 1) if concat else torch.mean(torch.stack(outs, 1), 1)
    dev = "cuda:0"
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev), slot_edges=16)
    xd = x.float().to(dev).requires_grad_(True)
    Ws = [W[h].float().to(dev).requires_grad_(True) for h in range(H)]
    As = [a[h].float().to(dev).reshape(1, -1).requires_grad_(True) for h in range(H)]
    Ss = [Sk[h].float().to(dev).requires_grad_(True) for h in range(H)] if skip else None
    masks = dict(x=mx.float().to(dev), whi=mi.float().to(dev), whj=mj.float().to(dev), att=matt.float().to(dev))
    out = gatv2_level(xd, g, Ws, As, Ss, 0.2, concat, p, masks=masks)
    out.backward(G.float().to(dev))
    got = [xd.grad, torch.stack([w.grad for w in Ws]), torch.stack([w.grad.reshape(-1) for w in As])]
    if skip:
        got.append(torch.stack([w.grad for w in Ss]))
    check_autograd(out, got, oracle, leaves, G, ["dX", "dW", "da", "dW_skip"], what=f"v2 dropout[{H},{Fin},{Fo},{skip},{concat}]")


@pytest.mark.parametrize("skip,concat", [(False, True), (True, True), (True, False)])
def test_dense_v2_dropout_explicit_masks(pg, skip, concat):  # noqa: F811
    """GraphAttentionLayerV2 in train mode (the reference's `--model GATv2` default, train.py:54,116: dropout 0.6):
    explicit masks through the same kernels against oracle.dense_head_forward_v2 (layers.py:206-230), all gradients,
    including the exactly-zero ones of `a` and W[:Fin]."""
    N, Fin, Fo, p = 70, 12, 8, 0.6
    rowptr, col = O.random_symmetric_csr(N, 5, 3, hub=(2, 40))
    E = len(col)
    adj = O.dense_from_csr(rowptr, col, N, dtype=torch.float64)
    gen = torch.Generator().manual_seed(9)
    keep = 1.0 - p
    mk = lambda *s: (torch.rand(*s, generator=gen) < keep).double() / keep  # noqa: E731
    mx, mwh1, mwh2, matt_e = mk(N, Fin), mk(N, Fo), mk(N, Fo), mk(E)
    src = np.repeat(np.arange(N), np.diff(rowptr))
    matt = torch.zeros(N, N, dtype=torch.float64)
    matt[torch.as_tensor(src), torch.as_tensor(col.astype(np.int64))] = matt_e     # dense [N,N] mask, layers.py:220
    torch.manual_seed(4)
    layer = pg.GraphAttentionLayerV2(Fin, Fo, p, 0.2, concat=concat, skip_connection=skip).to("cuda:0").train()
    x = torch.randn(N, Fin, generator=gen, dtype=torch.float64)
    G = torch.randn(N, Fo, generator=gen, dtype=torch.float64)
    leaves = [x, layer.W.detach().double().cpu(), layer.a.detach().double().cpu()]
    if skip:
        leaves.append(layer.skip_projection.detach().double().cpu())

    def oracle(*lv):
        c = lambda m: m.to(lv[0].dtype)  # noqa: E731
        return O.dense_head_forward_v2(lv[0], c(adj), lv[1], lv[2], 0.2, concat, lv[3] if skip else None,
                                       mask_x=c(mx), mask_wh1=c(mwh1), mask_wh2=c(mwh2), mask_att=c(matt))
    dev = "cuda:0"
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev))
    xd = x.float().to(dev).requires_grad_(True)
    masks = {"x": mx.float().to(dev)[None], "wh": mwh2.float().to(dev)[None], "att": matt_e.float().to(dev)[:, None]}
    out = layer(xd, g, masks=masks)
    out.backward(G.float().to(dev))
    _, (_, gr) = check_autograd(out, [xd.grad, layer.W.grad, None] + ([layer.skip_projection.grad] if skip else []), oracle,
                                leaves, G, ["dX", "dW", "da", "dW_skip"], what=f"dense v2 dropout[{skip},{concat}]")
    assert float(layer.W.grad[:Fin].abs().max()) == 0.0 and float(layer.a.grad.abs().max()) == 0.0
    assert gr[2] is None or float(gr[2].abs().max()) < 1e-10      # mathematically zero (uniform attention)
    # in-kernel masks: runs, finite, differs from eval
    out2 = layer(xd, g)
    assert torch.isfinite(out2).all() and float((out2 - layer.eval()(xd, g)).abs().max()) > 1e-3
