import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, numpy as np
import pygat_amd as pg
from pygat_amd import ops
dev = torch.device("cuda:0")
def timed(fn, iters=30, gap=0.005):
    ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)); time.sleep(gap)
    return float(np.median(ts[5:]))
for (M, N, K, sk) in [(3144, 2056, 1024, 1), (4096, 2048, 1024, 1), (3072, 2048, 1024, 1), (3144, 1024, 1024, 1), (4096, 1024, 1024, 1), (4096, 1024, 1024, 2), (3144, 1024, 1024, 2),
                      (2048, 1024, 1024, 4), (1024, 1024, 3144, 8), (1024, 2048, 3144, 4)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    f = lambda: ops.gemm(False, False, M, N, K, A, K, B, N, [(N, C, N)], split_k=sk)
    t = timed(f)
    tiles = -(-M // 128) * -(-N // 128)
    print(f"M {M} N {N} K {K} split {sk}: tiles {tiles} x {sk} = {tiles*sk:4d} work-groups  {t*1e3:7.1f} us  {2.0*M*N*K/t/1e9:6.1f} TF", flush=True)
