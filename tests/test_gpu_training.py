"""End-to-end training equivalence (the substitute SURVEY.md 8(c) defines for the "Cora accuracy +-0.3"
target, which cannot be measured offline: Cora's features and labels are missing blobs).

Citeseer topology + its REAL labels and train/val/test split (tests/golden/citeseer_labels.npz) +
seeded class-conditional synthetic features; the 2-level model of train.py:60-72 (8 heads x 8 -> 6,
1 head), ELU + log_softmax + NLL on idx_train (train.py:151-159), Adam lr 5e-3 wd 5e-4, dropout 0 so
both sides are deterministic.  Trained from the same initial weights once through the CPU oracle
(torch autograd, fp32 like the reference) and once through the HIP path: the loss must agree step by
step and the final test accuracy within 0.3 points.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gat_oracle as O
from test_gpu_parity import pg  # noqa: F401

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_citeseer_training_matches_oracle(pg, topologies):  # noqa: F811
    rowptr, col = topologies["citeseer"]
    z = np.load(os.path.join(GOLDEN, "citeseer_labels.npz"))
    y = torch.as_tensor(z["labels"].astype(np.int64))
    itr, ite = torch.as_tensor(z["idx_train"].astype(np.int64)), torch.as_tensor(z["idx_test"].astype(np.int64))
    N, C, Fin, steps = len(rowptr) - 1, 6, 96, 40
    gen = torch.Generator().manual_seed(72)
    centers = torch.randn(C, Fin, generator=gen)
    x = torch.relu(centers[y] * 0.6 + torch.randn(N, Fin, generator=gen))      # noisy class-conditional features
    x = x / x.sum(1, keepdim=True).clamp(min=1e-6)                              # utils.normalize_features
    nfeat, nheads = [Fin, 8, C], [8, 1]
    torch.manual_seed(72)
    model = pg.GAT(nfeat, nheads, 2, 0.0, 0.2, pg.SpGraphAttentionLayer).cuda()
    init = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    def loss_of(out, yy, idx):
        return F.nll_loss(F.log_softmax(F.elu(out), dim=1)[idx], yy[idx])       # train.py:151-152,159

    # ---- HIP path
    graph = pg.CSRGraph(torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda())
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, weight_decay=5e-4)
    xd, yd, itd = x.cuda(), y.cuda(), itr.cuda()
    gpu_losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = loss_of(model(xd, graph), yd, itd)
        loss.backward(); opt.step()
        gpu_losses.append(float(loss))
    with torch.no_grad():
        acc_gpu = float((model.eval()(xd, graph).argmax(1)[ite.cuda()] == yd[ite.cuda()]).float().mean())

    # ---- CPU oracle from the same initial weights
    levels = []
    for li, nh in enumerate(nheads):
        Ws = torch.stack([init[f"attention_layer_{li+1}_head_{j+1}.W"] for j in range(nh)]).clone().requires_grad_()
        As = torch.stack([init[f"attention_layer_{li+1}_head_{j+1}.a"].reshape(-1) for j in range(nh)]).clone().requires_grad_()
        levels.append(dict(W=Ws, a=As))
    opt_c = torch.optim.Adam([t for lv in levels for t in (lv["W"], lv["a"])], lr=5e-3, weight_decay=5e-4)
    cpu_losses = []
    for _ in range(steps):
        opt_c.zero_grad()
        loss = loss_of(O.model_forward(x, (rowptr, col), levels, 0.2), y, itr)
        loss.backward(); opt_c.step()
        cpu_losses.append(float(loss))
    with torch.no_grad():
        acc_cpu = float((O.model_forward(x, (rowptr, col), levels, 0.2).argmax(1)[ite] == y[ite]).float().mean())

    g, c = np.array(gpu_losses), np.array(cpu_losses)
    assert c[-1] < 0.8 * c[0], "the oracle run did not learn"            # the setup trains at all
    assert np.abs(g - c).max() < 2e-3 * c[0], (g[:5], c[:5], g[-3:], c[-3:])   # same trajectory
    assert abs(acc_gpu - acc_cpu) <= 0.003 + 1e-9, (acc_gpu, acc_cpu)    # +-0.3 points
    print(f"citeseer: loss {c[0]:.4f} -> {c[-1]:.4f} (cpu) / {g[-1]:.4f} (gpu); test acc cpu {acc_cpu:.4f} gpu {acc_gpu:.4f}")
