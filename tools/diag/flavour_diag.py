"""Diagnostic: full-size gradient errors of the three backward flavours against the fp64 C oracle."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pygat_amd as pg
from pygat_amd.rmat import rmat_csr
from oracle import c_oracle
subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rowptr, col = rmat_csr(scale, 5_000_000 * (1 << scale) // (1 << 20), seed=1, device=dev)
graph = pg.CSRGraph(rowptr, col)
H, Fo, Fin = 8, 16, 128
g = torch.Generator(device=dev).manual_seed(2)
X = torch.randn(graph.n, Fin, generator=g, device=dev)
W = torch.randn(H, Fin, Fo, generator=g, device=dev) * (1.414 * (2.0 / (Fin + Fo)) ** 0.5)
a = torch.randn(H, 2 * Fo, generator=g, device=dev) * (1.414 * (2.0 / (1 + 2 * Fo)) ** 0.5)
G = torch.randn(graph.n, H * Fo, generator=g, device=dev)
args = (X.cpu().numpy(), rowptr.cpu().numpy(), col.cpu().numpy(), W.cpu().numpy(), a.cpu().numpy(), 0.2, True, G.cpu().numpy())
tp = c_oracle.transpose_pattern(args[1], args[2])
r64 = c_oracle.level(*args, want_dx=True, tp=tp, dtype=np.float64)
r32 = c_oracle.level(*args, want_dx=True, tp=tp)
for k in ("out", "dW", "da", "dX"):
    print(f"fp32 oracle {k}: err {np.abs(r32[k].astype(np.float64) - r64[k]).max():.3e}  max|ref| {np.abs(r64[k]).max():.4g}")
for fl in ("rowlocal", "rowsum", "two-gather"):
    pg.ops.BACKWARD_FLAVOUR = fl
    Xd = X.clone().requires_grad_(True); Wd = W.clone().requires_grad_(True); ad = a.clone().requires_grad_(True)
    out = pg.GATLevelFn.apply(Xd, Wd, ad, None, graph, 0.2, True)
    out.backward(G)
    torch.cuda.synchronize()
    for k, t in (("out", out), ("dW", Wd.grad), ("da", ad.grad), ("dX", Xd.grad)):
        e = np.abs(t.detach().double().cpu().numpy() - r64[k])
        print(f"{fl:10s} {k}: err max {e.max():.3e} p99.99 {np.quantile(e, 0.9999):.3e} median {np.median(e):.3e}")
