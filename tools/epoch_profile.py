#!/usr/bin/env python3
"""The reference's epoch (train step + eval forward, train.py:151-179 / train_ppi.py:112-152) on one of bench.py's small
configurations, launched EAGERLY so that a profiler attributes every kernel (bench.py replays it from one HIP graph).

    rocprofv3 --kernel-trace --stats -d gpurun_out/X -- python3 tools/epoch_profile.py ppi --epochs 30
    python3 tools/epoch_profile.py ppi --spans          # HIP-event spans of the level's own launches (ops.KernelTimer)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import bench  # noqa: E402
import pygat_amd as pg  # noqa: E402
from pygat_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("name", choices=["cora", "pubmed", "ppi"])
ap.add_argument("--epochs", type=int, default=30)
ap.add_argument("--spans", action="store_true")
ap.add_argument("--aten-loss", action="store_true", help="the loss as the ATen sequence of train.py instead of the fused kernels")
args = ap.parse_args()
dev = torch.device("cuda", 0)
c = bench.EPOCH_CFG[args.name]
g = torch.Generator().manual_seed(72)
if args.name == "ppi":
    graph = bench.ppi_batch(pg, dev)
    N = graph.n
    x = torch.randn(N, c["nfeats"][0], generator=g).to(dev)
    y = (torch.rand(N, c["nfeats"][-1], generator=g) < 0.3).float().to(dev)
    loss_fn = pg.BCEWithLogits(y)          # train_ppi.py:114,157
else:
    z = np.load(os.path.join(ROOT, "tests", "golden", f"{args.name}_csr.npz"), allow_pickle=False)
    N = len(z["rowptr"]) - 1
    x = (torch.rand(N, c["nfeats"][0], generator=g) < c["density"]).float()
    x = (x / x.sum(1, keepdim=True).clamp(min=1)).to(dev)
    y = torch.randint(0, c["nfeats"][-1], (N,), generator=g).to(dev)
    it = torch.arange(c["ntrain"], device=dev)
    graph = pg.CSRGraph(torch.as_tensor(z["rowptr"], device=dev), torch.as_tensor(z["col"], device=dev))
    # train.py:151-152,159: nll_loss(log_softmax(elu(out))[idx_train], labels[idx_train]) -- one launch forward, one backward
    loss_fn = pg.EluLogSoftmaxNLL(it, y, N) if "--aten-loss" not in sys.argv else \
        (lambda out: F.nll_loss(F.log_softmax(F.elu(out), dim=1)[it], y[it]))
torch.manual_seed(72)
model = pg.GAT(c["nfeats"], c["nheads"], len(c["nheads"]), c["dropout"], 0.2, pg.SpGraphAttentionLayer,
               skip_connection=(args.name == "ppi")).to(dev)
opt = pg.Adam(model.parameters(), lr=c["lr"], weight_decay=c["wd"])


one = torch.ones((), device=dev)


def epoch():
    model.train()
    opt.zero_grad(set_to_none=True)
    loss = loss_fn(model(x, graph))
    loss.backward(one)
    opt.step()
    model.eval()
    with torch.no_grad():
        loss_fn(model(x, graph))


for _ in range(5):
    epoch()
torch.cuda.synchronize()
if args.spans:
    ops.TIMER = ops.KernelTimer()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.epochs):
    epoch()
e1.record()
torch.cuda.synchronize()
print(f"{args.name}: {graph.n} nodes, {graph.nnz} edges, eager epoch {e0.elapsed_time(e1) / args.epochs:.3f} ms")
if args.spans:
    for k, v in ops.TIMER.times_ms().items():
        print(f"  {k:18s} {len(v) // args.epochs:3d} launches/epoch  {sum(v) / args.epochs * 1e3:9.1f} us/epoch")
