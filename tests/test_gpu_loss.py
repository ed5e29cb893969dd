"""The fused training loss (pygat_amd.EluLogSoftmaxNLL, csrc/k8_loss.hip) against the oracle's restatement of
train.py:151-152,159 (oracle.train_loss: elu -> log_softmax -> nll_loss on an index subset), value and gradient, through
tests/parity.py's rule; repeated indices, a single class, logits of both signs and large magnitude, replays."""
import numpy as np
import pytest
import torch

from oracle import gat_oracle as O
from parity import close_fwd, close_grad

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,C,m,scale", [(2708, 7, 140, 1.0), (19717, 3, 60, 0.05), (3327, 6, 120, 30.0), (100, 1, 100, 1.0),
                                          (1000, 64, 5000, 3.0)])
def test_fused_loss_value_and_gradient(n, C, m, scale):
    import pygat_amd as pg
    g = torch.Generator().manual_seed(n + C)
    out = torch.randn(n, C, generator=g, dtype=torch.float64) * scale
    idx = torch.randint(0, n, (m,), generator=g)            # m > n: every row repeated several times
    y = torch.randint(0, C, (n,), generator=g)
    gup = 0.7

    def oracle(dtype):
        o = out.to(dtype).clone().requires_grad_(True)
        loss = O.train_loss(o, idx, y)
        (loss * gup).backward()
        return loss.detach().double(), o.grad.double()
    l64, d64 = oracle(torch.float64)
    l32, d32 = oracle(torch.float32)
    crit = pg.EluLogSoftmaxNLL(idx.cuda(), y.cuda(), n)
    od = out.float().cuda().requires_grad_(True)
    for _ in range(3):                                       # the workspace counter is left ready for the next launch
        od.grad = None
        loss = crit(od)
        (loss * gup).backward()
    torch.cuda.synchronize()
    assert loss.dim() == 0
    close_grad(loss.reshape(1), l64.reshape(1).numpy(), l32.reshape(1).numpy(), "loss")
    close_grad(od.grad, d64.numpy(), d32.numpy(), "d loss / d out")
    untouched = torch.ones(n, dtype=torch.bool); untouched[idx] = False
    assert float(od.grad[untouched.cuda()].abs().max() if untouched.any() else 0.0) == 0.0


def test_fused_loss_in_a_training_step_matches_the_aten_sequence(topologies):
    """The Cora-shaped model's step with the fused loss == the same step with the ATen sequence of train.py, gradient by
    gradient (both run the HIP levels; dropout 0 so that the two steps see the same masks)."""
    import torch.nn.functional as F
    import pygat_amd as pg
    rowptr, col = topologies["cora"]
    N = len(rowptr) - 1
    g = torch.Generator().manual_seed(72)
    x = (torch.rand(N, 300, generator=g) < 0.02).float()
    x = (x / x.sum(1, keepdim=True).clamp(min=1)).cuda()
    y = torch.randint(0, 7, (N,), generator=g).cuda()
    it = torch.arange(140).cuda()
    graph = pg.CSRGraph(torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda())
    torch.manual_seed(5)
    model = pg.GAT([300, 8, 7], [8, 1], 2, 0.0, 0.2, pg.SpGraphAttentionLayer).cuda()
    crit = pg.EluLogSoftmaxNLL(it, y, N)
    grads = []
    for fused in (False, True):
        model.zero_grad(set_to_none=True)
        out = model(x, graph)
        loss = crit(out) if fused else F.nll_loss(F.log_softmax(F.elu(out), dim=1)[it], y[it])
        loss.backward()
        grads.append((float(loss), [p.grad.detach().double().cpu().numpy() for p in model.parameters()]))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6
    for a_, b_ in zip(grads[0][1], grads[1][1]):
        assert np.abs(a_ - b_).max() <= 1e-6 * max(1.0, np.abs(a_).max())


@pytest.mark.parametrize("n,C,scale", [(3144, 121, 1.0), (1612, 121, 20.0), (7, 3, 1.0), (5000, 50, 0.01)])
def test_fused_bce_with_logits(n, C, scale):
    """pygat_amd.BCEWithLogits against nn.BCEWithLogitsLoss(reduction='mean') (train_ppi.py:114,157) in fp64 (priced by the
    fp32 CPU evaluation of the same formula), value and gradient, logits of both signs and large magnitude, replays."""
    import pygat_amd as pg
    g = torch.Generator().manual_seed(n + C)
    x = torch.randn(n, C, generator=g, dtype=torch.float64) * scale
    y = (torch.rand(n, C, generator=g) < 0.3).double()
    gup = 1.3

    def ref(dtype):
        xx = x.to(dtype).clone().requires_grad_(True)
        loss = torch.nn.BCEWithLogitsLoss(reduction="mean")(xx, y.to(dtype))
        (loss * gup).backward()
        return loss.detach().double(), xx.grad.double()
    l64, d64 = ref(torch.float64)
    l32, d32 = ref(torch.float32)
    crit = pg.BCEWithLogits(y.float().cuda())
    xd = x.float().cuda().requires_grad_(True)
    for rep in range(3):                                        # the workspace counter is left ready for the next call
        xd.grad = None
        loss = crit(xd)
        (loss * gup).backward()
        close_fwd(loss.reshape(1), l64.reshape(1), f"bce loss rep {rep}", ref32=l32.reshape(1))
        close_grad(xd.grad, d64, d32, f"bce gradient rep {rep}", floor=1e-9)
