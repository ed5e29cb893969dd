#!/usr/bin/env python3
"""What a row chunk of the pipelined forward costs (pygat_amd/dist.py, bench.py --chunks): the config-5 level's forward with K2 cut
into c = 1, 2, 4, 8 row chunks and a no-op hand-off, timed with HIP events (forward only, and forward + backward).
    python3 tools/chunk_cost.py [--scale 20] [--heads 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygat_amd as pg  # noqa: E402
from pygat_amd.rmat import rmat_csr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=20)
ap.add_argument("--draws", type=int, default=5_000_000)
ap.add_argument("--heads", type=int, default=8)
ap.add_argument("--fout", type=int, default=16)
ap.add_argument("--chunks", type=int, nargs="*", default=[1, 2, 4, 8])
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
rowptr, col = rmat_csr(a.scale, a.draws, seed=1, device=dev)
graph = pg.CSRGraph(rowptr, col)
N = graph.n
g = torch.Generator(device=dev).manual_seed(2)
X = torch.randn(N, 128, generator=g, device=dev)
W = (torch.randn(a.heads, 128, a.fout, generator=g, device=dev) * 0.17).requires_grad_(True)
av = (torch.randn(a.heads, 2 * a.fout, generator=g, device=dev) * 0.3).requires_grad_(True)
G = torch.randn(N, a.heads * a.fout, generator=g, device=dev)
buf = torch.empty(N, a.heads * a.fout, device=dev)
for c in a.chunks:
    pipe = None if c == 0 else (c, lambda *_: None, buf)

    def fwd():
        W.grad = av.grad = None
        return pg.GATLevelFn.apply(X, W, av, None, graph, 0.2, True, None, pipe)
    for _ in range(3):
        fwd().backward(G)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(a.steps):
        e[0].record(); out = fwd(); e[1].record(); out.backward(G); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"chunks {c}: forward {tf / a.steps:.3f} ms, backward {tb / a.steps:.3f} ms", flush=True)
