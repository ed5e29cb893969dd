#!/usr/bin/env python3
"""Slab-count sweep of the streamed-K weight-gradient GEMM dW = X^T dWh (development tool).

    python tools/gemm_tn_sweep.py [--cols 128 512 1024]

Prints, per output width, the median time for several split_k values."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pygat_amd as pg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cols", type=int, nargs="+", default=[128, 512, 1024])
ap.add_argument("--splits", type=int, nargs="+", default=[16, 32, 64, 128, 256])
args = ap.parse_args()
N, M = 1 << 20, 128
X = torch.randn(N, M, device="cuda")


def run(C, split):
    D = torch.randn(N, C, device="cuda")
    out = torch.empty(M, C, device="cuda")
    ts = []
    for _ in range(9):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); pg.gemm(True, False, M, C, N, X, M, D, C, [(C, out, C)], split_k=split); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[3:]))


for C in args.cols:
    print(C, {s: round(run(C, s), 3) for s in args.splits}, flush=True)
