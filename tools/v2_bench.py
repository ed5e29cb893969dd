#!/usr/bin/env python3
"""GATv2 (SpGraphAttentionLayerV2) level forward+backward on the config-5 R-MAT graph (development tool)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import pygat_amd as pg
from pygat_amd.rmat import rmat_csr
dev = torch.device("cuda", 0)
rowptr, col = rmat_csr(20, 5_000_000, seed=1, device=dev)
graph = pg.CSRGraph(rowptr, col)
N, E, H, Fo, Fin = graph.n, graph.nnz, 8, 16, 128
g = torch.Generator(device=dev).manual_seed(2)
X = torch.randn(N, Fin, generator=g, device=dev)
W = (torch.randn(H, 2 * Fin, Fo, generator=g, device=dev) * 0.12).requires_grad_()
a = (torch.randn(H, Fo, generator=g, device=dev) * 0.3).requires_grad_()
G = torch.randn(N, H * Fo, generator=g, device=dev)
def step():
    W.grad = a.grad = None
    out = pg.GATv2LevelFn.apply(X, W, a, None, graph, 0.2, True, None)
    out.backward(G)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
print(json.dumps({"layer": "SpGraphAttentionLayerV2", "N": N, "E": E, "heads": H, "f_out": Fo, "ms_per_step": ms, "edges_per_s": E / ms * 1e3}))
