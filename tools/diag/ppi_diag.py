"""Diagnostic: per-level error of the PPI-shaped model against the fp64 oracle (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ppi_case as P
from oracle import gat_oracle as O
import pygat_amd as pg
dev = "cuda:0"
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11
torch.manual_seed(seed)
m = pg.GAT(P.NFEAT, P.NHEADS, 3, 0.0, 0.2, pg.SpGraphAttentionLayer, skip_connection=True)
parts = P.graphs(); rp, col = P.batch_csr(parts)
g = pg.CSRGraph(torch.as_tensor(rp, device=dev), torch.as_tensor(col, device=dev))
x = torch.as_tensor(P.features()).double()
rng = np.random.default_rng(7)
lev64 = P.oracle_levels(m, torch.float64)
for li, lv in enumerate(lev64):
    concat = li < 2
    H, Fin, Fo = lv["W"].shape
    G = torch.as_tensor(rng.standard_normal((x.shape[0], H * Fo if concat else Fo)))
    res = {}
    for dt in (torch.float64, torch.float32):
        xx = x.detach().clone().to(dt).requires_grad_(True)
        W = lv["W"].detach().to(dt).requires_grad_(True); a = lv["a"].detach().to(dt).requires_grad_(True)
        S = lv["skip"].detach().to(dt).requires_grad_(True)
        y = O.level_forward(xx, (rp, col), W, a, 0.2, concat, S, "sparse")
        y.backward(G.to(dt))
        res[dt] = [t.detach().double() for t in (y, xx.grad, W.grad, a.grad, S.grad)]
    xd = x.float().to(dev).requires_grad_(True)
    Wd = lv["W"].detach().float().to(dev).requires_grad_(True); ad = lv["a"].detach().float().to(dev).requires_grad_(True)
    Sd = lv["skip"].detach().float().to(dev).requires_grad_(True)
    out = pg.GATLevelFn.apply(xd, Wd, ad, Sd, g, 0.2, concat)
    out.backward(G.float().to(dev))
    got = [t.detach().double().cpu() for t in (out, xd.grad, Wd.grad, ad.grad, Sd.grad)]
    for name, gt, r64, r32 in zip(("out", "dX", "dW", "da", "dSk"), got, res[torch.float64], res[torch.float32]):
        e = (gt - r64).abs(); own = (r32 - r64).abs()
        print(f"level {li+1} {name:4s} max|ref| {float(r64.abs().max()):9.3g}  hip err max {float(e.max()):.3e} p99.9 {float(e.flatten().quantile(0.999)):.3e}"
              f"   fp32-oracle err max {float(own.max()):.3e} p99.9 {float(own.flatten().quantile(0.999)):.3e}")
        if name == "dX":
            rows = e.max(1).values
            print("      rows with err > 10 x median:", int((rows > 10 * rows.median()).sum()), "of", len(rows), " worst rows", rows.topk(5).indices.tolist())
    x = res[torch.float64][0]
