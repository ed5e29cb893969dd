import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import pygat_amd as pg
for name, tA, tB, M, N, K in [("L2 project", False, False, 3144, 2056, 1024), ("L2 dW", True, False, 1024, 1024, 3144), ("L2 dX", False, True, 3144, 1024, 1024),
                              ("L3 project", False, False, 3144, 1548, 1024)]:
    A = torch.randn((K, M) if tA else (M, K), device="cuda"); B = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    for sk in (1, 2, 3, 4, 6, 8):
        ts = []
        for _ in range(10):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); pg.gemm(tA, tB, M, N, K, A, A.shape[1], B, B.shape[1], [(N, C, N)], split_k=sk); b.record()
            torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ms = float(np.median(ts[3:]))
        print(f"{name} split {sk}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF", flush=True)
