#!/usr/bin/env python3
"""Development aid: where the projection GEMM (K1, gemm_smallk_x3_kernel) spends a launch, from in-kernel stamps of a diagnostic
build (tools/build_variant.sh <name> k1_gemm_x3.hip "-DPYGAT_DIAG_K1=16", PYGAT_AMD_LIB=<that .so>): shader clock (s_memtime
against the 100 MHz s_memrealtime), MFMA phase and epilogue per tile, and when the waves start and end.

Here: the projection INSIDE the headline training step (bench.py's level: RMAT scale 20, 8 x 16, F 128) -- the clock it runs at
there is the one the kernels before it leave behind.  By itself: tools/gemm_headline_bench.py --stamps [--gap-ms 20].
"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pygat_amd as pg  # noqa: E402
from pygat_amd._lib import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gap-ms", type=float, default=0.0)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--scale", type=int, default=20)
a_ = ap.parse_args()
if not hasattr(lib, "pygat_diag_k1_stamps"):
    sys.exit("this library has no stamps: build one with -DPYGAT_DIAG_K1=16 and set PYGAT_AMD_LIB")
lib.pygat_diag_k1_stamps.restype = ctypes.c_int
lib.pygat_diag_k1_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
H, Fo, Fin = 8, 16, 128
from pygat_amd.rmat import rmat_csr  # noqa: E402
rowptr, col = rmat_csr(a_.scale, 5_000_000, seed=1, device=dev)
graph = pg.CSRGraph(rowptr, col)
N = graph.n
g = torch.Generator(device=dev).manual_seed(2)
X = torch.randn(N, Fin, generator=g, device=dev)
W = (torch.randn(H, Fin, Fo, generator=g, device=dev) * 0.17).requires_grad_(True)
a = (torch.randn(H, 2 * Fo, generator=g, device=dev) * 0.3).requires_grad_(True)
G = torch.randn(N, H * Fo, generator=g, device=dev)


def step():
    W.grad = a.grad = None
    pg.GATLevelFn.apply(X, W, a, None, graph, 0.2, True).backward(G)


def read():
    buf = np.zeros(2048 * 64, dtype=np.uint64)
    n = lib.pygat_diag_k1_stamps(buf.ctypes.data, buf.size)
    assert n == buf.size, n
    raw = buf.reshape(-1, 8).astype(np.int64)
    return raw[raw[:, 3] > 0]


fn = step
for _ in range(5):
    fn()
torch.cuda.synchronize()
rows = []
for _ in range(a_.iters):
    fn()
    torch.cuda.synchronize()
    raw = read()
    tot, mf, ep, nt = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3]
    t0 = raw[:, 4].min()
    span = (raw[:, 6].max() - t0) * 0.01
    clock = np.median(tot / np.maximum(raw[:, 6] - raw[:, 5], 1)) * 0.1
    rows.append((clock, span, np.median(tot / nt), np.median(mf / nt), np.median(ep / nt), nt.sum()))
    if a_.gap_ms:
        time.sleep(a_.gap_ms * 1e-3)
r = np.median(np.array(rows), axis=0)
tiles = r[5]
mfma_cycles = tiles * 288 * 32 / 1024          # per SIMD: 288 MFMAs of 32 cycles per 32-row tile, 1024 SIMDs
print(f"K1 inside the training step: shader clock {r[0]:.2f} GHz, first entry -> last wave's end {r[1]:.1f} us; "
      f"per tile and wave {r[2]:.0f} cycles (MFMA phase {r[3]:.0f}, epilogue {r[4]:.0f}); the MFMA pipe's own work "
      f"{mfma_cycles:.0f} cycles per SIMD = {mfma_cycles / (r[0] * 1e3):.1f} us at this clock = {mfma_cycles / (r[0] * 1e3) / r[1]:.2f} of the span")
