#!/usr/bin/env python3
"""Epoch-time benchmark on the reference's small configurations (BASELINE.json configs 1-4).

    python tools/epoch_bench.py [--dataset cora|citeseer|pubmed|ppi] [--epochs 50] [--graph-capture]

Reproduces what the reference calls an epoch (train.py:154-179: one training step -- forward,
ELU + log_softmax + NLL on idx_train, backward, Adam with weight decay -- plus one eval forward
unless --fastmode), with the per-dataset hyper-parameters of train.py:47-87 / train_ppi.py:43-57,
on the REAL topology (tests/golden/*_csr.npz) and seeded synthetic features / labels (the
reference's feature and label blobs are missing).  PPI: two synthetic graphs with
real node counts, block-diagonal batch (load_data_ppi.py:71-88), BCE-with-logits, skip connections.

--graph-capture replays the whole epoch from one HIP graph (small graphs are launch-bound).
Prints one JSON line; `cpu_oracle_epoch_s` is the same epoch through oracle/gat_oracle.py (torch CPU
autograd, sparse formulation) -- a stand-in for the reference's CPU path, which cannot be imported.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

CFG = {  # train.py:47-87, train_ppi.py:43-57
    "cora": dict(nlayers=2, nheads=[8, 1], nfeats=[1433, 8, 7], alpha=0.2, dropout=0.6, lr=5e-3, wd=5e-4, skip=False),
    "citeseer": dict(nlayers=2, nheads=[8, 1], nfeats=[3703, 8, 6], alpha=0.2, dropout=0.6, lr=5e-3, wd=5e-4, skip=False),
    "pubmed": dict(nlayers=2, nheads=[8, 8], nfeats=[500, 8, 3], alpha=0.2, dropout=0.6, lr=1e-2, wd=1e-3, skip=False),
    "ppi": dict(nlayers=3, nheads=[4, 4, 6], nfeats=[50, 256, 256, 121], alpha=0.2, dropout=0.0, lr=5e-3, wd=0.0, skip=True),
}


def load(dataset, seed=72):
    g = torch.Generator().manual_seed(seed)
    if dataset == "ppi":
        from oracle import gat_oracle as O
        sizes = np.load(os.path.join(ROOT, "tests", "golden", "ppi_graph_sizes.npz"))["train"][:2]
        rps, cols, off = [np.zeros(1, dtype=np.int64)], [], 0
        for k, n in enumerate(sizes):                 # block-diagonal batch of 2 graphs, mean degree ~28
            rp, c = O.random_symmetric_csr(int(n), 28, 100 + k)
            rps.append(rp[1:].astype(np.int64) + rps[-1][-1]); cols.append(c.astype(np.int64) + off); off += int(n)
        rowptr = np.concatenate(rps).astype(np.int32); col = np.concatenate(cols).astype(np.int32)
        N = off
        x = torch.randn(N, 50, generator=g)
        y = (torch.rand(N, 121, generator=g) < 0.3).float()
        return rowptr, col, x, y, torch.arange(N)
    z = np.load(os.path.join(ROOT, "tests", "golden", f"{dataset}_csr.npz"))
    rowptr, col = z["rowptr"], z["col"]
    N, c = len(rowptr) - 1, CFG[dataset]
    x = (torch.rand(N, c["nfeats"][0], generator=g) < 0.013).float()
    x = x / x.sum(1, keepdim=True).clamp(min=1)       # utils.normalize_features
    y = torch.randint(0, c["nfeats"][-1], (N,), generator=g)
    ntrain = {"cora": 140, "citeseer": 120, "pubmed": 60}[dataset]
    return rowptr, col, x, y, torch.arange(ntrain)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="cora")
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--graph-capture", action="store_true")
    ap.add_argument("--fastmode", action="store_true")
    ap.add_argument("--fused-adam", action="store_true", help="torch's single-kernel Adam (same update rule)")
    ap.add_argument("--cpu-epochs", type=int, default=2)
    args = ap.parse_args()
    c = CFG[args.dataset]
    import pygat_amd as pg
    dev = torch.device("cuda", 0)
    rowptr, col, x, y, idx_train = load(args.dataset)
    graph = pg.CSRGraph(torch.as_tensor(rowptr, device=dev), torch.as_tensor(col, device=dev))
    torch.manual_seed(72)
    model = pg.GAT(c["nfeats"], c["nheads"], c["nlayers"], c["dropout"], c["alpha"], pg.SpGraphAttentionLayer,
                   skip_connection=c["skip"]).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=c["lr"], weight_decay=c["wd"], capturable=args.graph_capture,
                           **({"fused": True} if args.fused_adam else {}))
    xd, yd, it = x.to(dev), y.to(dev), idx_train.to(dev)

    def loss_fn(out):
        if args.dataset == "ppi":
            return F.binary_cross_entropy_with_logits(out, yd)          # train_ppi.py:104
        return F.nll_loss(F.log_softmax(F.elu(out), dim=1)[it], yd[it])  # train.py:151-152,159

    def epoch():
        model.train()
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(model(xd, graph))
        loss.backward()
        opt.step()
        if not args.fastmode:
            model.eval()
            with torch.no_grad():
                loss_val = loss_fn(model(xd, graph))
            return loss, loss_val
        return loss, loss

    losses = []
    for _ in range(5):
        losses.append(float(epoch()[0]))
    torch.cuda.synchronize()
    run = epoch
    if args.graph_capture:
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                epoch()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(gr):
            static = epoch()
        run = lambda: (gr.replay(), static)[1]  # noqa: E731
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.epochs):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.epochs
    losses.append(float(out[0]))
    res = {"dataset": args.dataset, "N": len(rowptr) - 1, "E": len(col), "epoch_s": dt, "graph_capture": args.graph_capture,
           "loss_first_last": [losses[0], losses[-1]], "edges_per_s_epoch": len(col) / dt}
    if args.cpu_epochs:
        from oracle import gat_oracle as O
        torch.manual_seed(72)
        levels = []
        nh = [1] + c["nheads"]
        for i in range(c["nlayers"]):
            fin, fo, H = c["nfeats"][i] * nh[i], c["nfeats"][i + 1], nh[i + 1]
            levels.append(dict(W=(torch.randn(H, fin, fo) * 0.1).requires_grad_(), a=(torch.randn(H, 2 * fo) * 0.1).requires_grad_(),
                               skip=(torch.randn(H, fin, fo) * 0.1).requires_grad_() if c["skip"] else None))
        ts = []
        for _ in range(args.cpu_epochs):
            t0 = time.perf_counter()
            out = O.model_forward(x, (rowptr, col), levels, c["alpha"])
            (F.binary_cross_entropy_with_logits(out, y) if args.dataset == "ppi"
             else F.nll_loss(F.log_softmax(F.elu(out), 1)[idx_train], y[idx_train])).backward()
            with torch.no_grad():
                O.model_forward(x, (rowptr, col), levels, c["alpha"])
            ts.append(time.perf_counter() - t0)
        res["cpu_oracle_epoch_s"] = min(ts)
        res["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
