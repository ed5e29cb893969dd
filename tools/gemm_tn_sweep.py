import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import pygat_amd as pg
N, M, C = 1 << 20, 128, 128
X = torch.randn(N, M, device="cuda"); D = torch.randn(N, C, device="cuda"); out = torch.empty(M, C, device="cuda")
def run(split):
    ts = []
    for i in range(13):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); pg.gemm(True, False, M, C, N, X, M, D, C, [(C, out, C)], split_k=split); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[3:]))
print(os.environ.get("PYGAT_TN_NOWIDE"), os.environ.get("PYGAT_TN_UK"), {s: round(run(s), 3) for s in (256, 512, 768, 1024, 2048)})
