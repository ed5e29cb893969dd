#!/usr/bin/env python3
"""Times pygat_gemm_f32 on a few shapes of the reference's configurations (development tool).

    python tools/gemm_bench.py            # PPI level 2/3 projections, weight and input gradients; Cora eval projection
    python tools/gemm_bench.py --only ppi --splits 1,2,4,8     # the same shapes at forced split-K factors
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pygat_amd as pg  # noqa: E402

SHAPES = [  # (name, transA, transB, M, N, K)
    ("cfg5 project     X[1M,128] W[128,128]", False, False, 1 << 20, 128, 128),
    ("cfg5 dW          X^T[128,1M] dWh[1M,128]", True, False, 128, 128, 1 << 20),
    ("cfg5 dX          dWh[1M,128] W^T[128,128]", False, True, 1 << 20, 128, 128),
    ("hidden project   X[1M,64] W[64,128]", False, False, 1 << 20, 128, 64),
    ("rank project     X[1M,128] W[128,32]", False, False, 1 << 20, 32, 128),
    ("ppi L2 project   X[3144,1024] Wcat[1024,2056]", False, False, 3144, 2056, 1024),
    ("ppi L2 dW        X^T[1024,3144] dWh[3144,1024]", True, False, 1024, 1024, 3144),
    ("ppi L2 dX        dWh[3144,1024] Wcat^T[1024,1024]", False, True, 3144, 1024, 1024),
    ("ppi L1 project   X[3144,50] Wcat[50,2056]", False, False, 3144, 2056, 50),
    ("ppi L3 project   X[3144,1024] Wcat[1024,1548]", False, False, 3144, 1548, 1024),
    ("ppi L3 dX        dWh[3144,768] Wcat^T[1024,768]", False, True, 3144, 1024, 768),
    ("F'64 project     X[1M,128] Wcat[128,520]", False, False, 1 << 20, 520, 128),
    ("F'64 dW          X^T[128,1M] dWh[1M,512]", True, False, 128, 512, 1 << 20),
    ("F'64 dX          dWh[1M,512] W^T[128,512]", False, True, 1 << 20, 128, 512),
    ("v2 cfg5 project  X[1M,128] Wcat[128,256]", False, False, 1 << 20, 256, 128),
    ("cora eval project X[2708,1433] Wcat[1433,72]", False, False, 2708, 72, 1433),
    ("pubmed project   X[19717,500] Wcat[500,72]", False, False, 19717, 72, 500),
]
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
splits = [int(v) for v in sys.argv[sys.argv.index("--splits") + 1].split(",")] if "--splits" in sys.argv else [None]
modes = sys.argv[sys.argv.index("--modes") + 1].split(",") if "--modes" in sys.argv else ["fp32-mfma", "split-bf16"]
for mode, sk, (name, tA, tB, M, N, K) in [(m, q, sh) for sh in SHAPES for m in modes for q in splits if only in sh[0]]:
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    B = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    lda, ldb = A.shape[1], B.shape[1]
    ts = []
    for _ in range(12):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); pg.gemm(tA, tB, M, N, K, A, lda, B, ldb, [(N, C, N)], split_k=sk, mode=mode); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts[3:]))
    ref = (A.t() if tA else A).double() @ (B.t() if tB else B).double()
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    used = sk if sk is not None else pg.ops._split_k(M, N, K, streamed_k=tA and not tB, mode=mode)
    print(f"{mode:10s} split_k {used:3d} {name:52s} {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF  rel err {err:.1e}", flush=True)
