import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import pygat_amd as pg
from pygat_amd import ops, _lib
dev = torch.device("cuda", 0)
N = 1 << 20
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(N, 128, generator=g, device=dev); W = torch.randn(128, 128, generator=g, device=dev) * 0.1
C1 = torch.empty(N, 128, device=dev); C2 = torch.empty(N, 128, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
blk = _lib.ColBlocks(128, N * 128)
a = t(lambda: ops.gemm(False, False, N, 128, 128, X, 128, W, 128, [(128, C1, 128)]))
b = t(lambda: ops.gemm(False, False, N, 128, 128, X, 128, W, 128, [(128, C2, 128)], a_blocks=blk))
print(f"smallk_x3 (fast path) {a*1e3:.1f} us   x3gw (general, W re-staged per tile) {b*1e3:.1f} us   max diff {float((C1-C2).abs().max()):.2e}")
