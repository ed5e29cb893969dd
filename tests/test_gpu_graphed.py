"""HIP-graph replay of the hot path (pygat_amd/graphed.py): a captured level must be the eager level bit for
bit, and a captured epoch (reference train.py:151-179) must be the eager epochs, replay after replay."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gat_oracle as O
from test_gpu_parity import pg, params  # noqa: F401

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


@pytest.mark.parametrize("H,Fin,Fo,skip,concat,need_dx", [(8, 32, 16, False, True, False), (4, 24, 8, True, True, True),
                                                          (6, 40, 7, True, False, True), (1, 16, 16, False, True, False)])
def test_graphed_level_is_the_eager_level(pg, H, Fin, Fo, skip, concat, need_dx):  # noqa: F811
    N = 5000                                   # >= 4096 rows: the first-level backward folds ds into the dW GEMM
    rowptr, col = O.random_symmetric_csr(N, 7, 5, hub=(3, 900))
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=DEV), torch.as_tensor(col, device=DEV))
    W, a, Sk = params(H, Fin, Fo, skip, 1)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(N, Fin, generator=gen)
    lvl = pg.GraphedLevel(g, x.to(DEV), W.float().to(DEV), a.float().to(DEV), Sk.float().to(DEV) if skip else None,
                          alpha=0.2, concat=concat, need_dx=need_dx)
    for trial in range(3):                     # new values through the same captured graphs
        xs = (x * (1 + 0.1 * trial)).to(DEV).requires_grad_(need_dx)
        Ws = (W.float() * (1 - 0.05 * trial)).to(DEV).requires_grad_(True)
        As = (a.float() + 0.01 * trial).to(DEV).requires_grad_(True)
        Ss = (Sk.float() * (1 + 0.02 * trial)).to(DEV).requires_grad_(True) if skip else None
        G = torch.randn(N, H * Fo if concat else Fo, generator=gen).to(DEV)
        y = lvl(xs, Ws, As, Ss)
        y.backward(G)
        got = [y.detach().clone()] + [t.grad.clone() for t in (xs, Ws, As, Ss) if t is not None and t.grad is not None]
        xe = xs.detach().clone().requires_grad_(need_dx)
        We, Ae = Ws.detach().clone().requires_grad_(True), As.detach().clone().requires_grad_(True)
        Se = Ss.detach().clone().requires_grad_(True) if skip else None
        ye = pg.GATLevelFn.apply(xe, We, Ae, Se, g, 0.2, concat)
        ye.backward(G)
        want = [ye.detach()] + [t.grad for t in (xe, We, Ae, Se) if t is not None and t.grad is not None]
        assert len(got) == len(want) == 3 + int(need_dx) + int(skip)
        for k, (p, q) in enumerate(zip(got, want)):
            assert torch.equal(p, q), (trial, k, float((p - q).abs().max()))     # same kernels, same order: bitwise


def _citeseer(pg, topologies, dropout):  # noqa: F811
    rowptr, col = topologies["citeseer"]
    z = np.load(os.path.join(GOLDEN, "citeseer_labels.npz"))
    y = torch.as_tensor(z["labels"].astype(np.int64), device=DEV)
    itr = torch.as_tensor(z["idx_train"].astype(np.int64), device=DEV)
    ival = torch.as_tensor(z["idx_val"].astype(np.int64), device=DEV)
    N, C, Fin = len(rowptr) - 1, 6, 64
    gen = torch.Generator().manual_seed(72)
    centers = torch.randn(C, Fin, generator=gen)
    x = torch.relu(centers[y.cpu()] * 0.6 + torch.randn(N, Fin, generator=gen))
    x = (x / x.sum(1, keepdim=True).clamp(min=1e-6)).to(DEV)
    graph = pg.CSRGraph(torch.as_tensor(rowptr, device=DEV), torch.as_tensor(col, device=DEV))

    def make():
        torch.manual_seed(72)
        m = pg.GAT([Fin, 8, C], [8, 1], 2, dropout, 0.2, pg.SpGraphAttentionLayer).to(DEV)
        return m, torch.optim.Adam(m.parameters(), lr=5e-3, weight_decay=5e-4, capturable=True)     # train.py:64-66,122
    loss_fn = lambda out: F.nll_loss(F.log_softmax(F.elu(out), dim=1)[itr], y[itr])                 # noqa: E731  train.py:151-159
    eval_fn = lambda out: F.nll_loss(F.log_softmax(F.elu(out), dim=1)[ival], y[ival])               # noqa: E731  train.py:169-171
    return x, graph, make, loss_fn, eval_fn


def test_fused_epoch_replays_equal_eager_epochs(pg, topologies):  # noqa: F811
    """dropout 0: n replays of the captured epoch == n eager epochs, loss by loss and parameter by parameter."""
    x, graph, make, loss_fn, eval_fn = _citeseer(pg, topologies, 0.0)
    warm, n = 2, 12
    m1, o1 = make()
    ep = pg.FusedEpoch(m1, o1, x, graph, loss_fn, eval_fn, warmup=warm)
    fused = []
    for _ in range(n):
        lt, lv = ep.run()
        fused.append((float(lt), float(lv)))
    m2, o2 = make()
    ref = pg.FusedEpoch(m2, o2, x, graph, loss_fn, eval_fn, capture=False)     # the same epoch body, never captured
    eager = []
    for _ in range(warm + n):
        lt, lv = ref.run()
        eager.append((float(lt), float(lv)))
    f, e = np.array(fused), np.array(eager[warm:])
    assert e[-1, 0] < e[0, 0], "the eager run does not learn"
    assert np.abs(f - e).max() <= 1e-6, (f[:3], e[:3], np.abs(f - e).max())
    for (k, p), (_, q) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert float((p - q).abs().max()) <= 1e-6, k
    assert ep.epochs == n


def test_fused_epoch_dropout_masks_are_fresh_on_every_replay(pg, topologies):  # noqa: F811
    """dropout 0.6 (train.py's default): the in-kernel masks come from a seed in device memory, so replays of ONE
    captured graph see different masks (train loss changes with frozen weights, eval loss does not); and with a
    learning rate the captured epoch trains."""
    x, graph, make, loss_fn, eval_fn = _citeseer(pg, topologies, 0.6)
    m, _ = make()
    frozen = torch.optim.Adam(m.parameters(), lr=0.0, capturable=True)
    ep = pg.FusedEpoch(m, frozen, x, graph, loss_fn, eval_fn, warmup=2)
    seen = [tuple(float(v) for v in ep.run()) for _ in range(4)]
    train, val = [s[0] for s in seen], [s[1] for s in seen]
    assert all(np.isfinite(train)) and len(set(train)) == 4, train          # fresh masks every replay
    assert len(set(val)) == 1, val                                           # eval mode: no dropout, same weights
    m2, o2 = make()
    ep2 = pg.FusedEpoch(m2, o2, x, graph, loss_fn, eval_fn, warmup=2)
    vals = [float(ep2.run()[1]) for _ in range(150)]
    assert np.isfinite(vals).all() and vals[-1] < 0.93 * vals[0], (vals[0], vals[-1])     # 60 epochs measured: -7.5 %


@pytest.mark.parametrize("nchunks", [2, 5])
def test_row_chunked_forward_is_the_forward(pg, nchunks):  # noqa: F811
    """GATLevelFn(pipeline=...): K2 chunk of rows by chunk of rows (what pygat_amd/dist.py overlaps with the
    all-gather) gives the unchunked result bit for bit, gradients included; the callback sees every row once, in
    order, and a chunk border never cuts a row."""
    N, H, Fin, Fo = 6000, 4, 24, 16
    rowptr, col = O.random_symmetric_csr(N, 9, 11, hub=(1500, 2500))      # a 2500-edge row in the middle
    g = pg.CSRGraph(torch.as_tensor(rowptr, device=DEV), torch.as_tensor(col, device=DEV), slot_edges=16)
    W, a, _ = params(H, Fin, Fo, False, 3)
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(N, Fin, generator=gen).to(DEV)
    G = torch.randn(N, H * Fo, generator=gen).to(DEV)
    seen = []

    def on_chunk(c, r0, r1, out):
        seen.append((c, r0, r1))

    res = []
    for pipe in (None, (nchunks, on_chunk)):
        Wd, ad = W.float().to(DEV).requires_grad_(True), a.float().to(DEV).requires_grad_(True)
        y = pg.GATLevelFn.apply(x, Wd, ad, None, g, 0.2, True, None, pipe)
        y.backward(G)
        res.append((y.detach(), Wd.grad, ad.grad))
    for p, q in zip(*res):
        assert torch.equal(p, q)
    assert [c for c, _, _ in seen] == list(range(len(seen))) and 2 <= len(seen) <= nchunks
    assert seen[0][1] == 0 and seen[-1][2] == N and all(a1[2] == b1[1] for a1, b1 in zip(seen, seen[1:]))
