#!/usr/bin/env python3
"""VERDICT round 4 item 5, the bound before the build: how much can ANY processing order buy the gather kernels on the config-5
graph?  The level's kernels are timed (HIP events per launch, ops.KernelTimer) on the SAME R-MAT pattern under four node
numberings -- the generator's (hubs at low ids), a random one, reverse Cuthill-McKee (scipy: the bandwidth-minimising order,
neighbours get nearby ids = nearby table rows AND nearby slots) and degree-descending.  A renumbering moves both the slot order
and the table layout, so it bounds what a slot -> work-group permutation alone (which leaves the table where it is) could gain.
    python3 tools/locality_probe.py [--scale 20]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygat_amd as pg  # noqa: E402
from pygat_amd import ops  # noqa: E402
from pygat_amd.rmat import rmat_csr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=20)
ap.add_argument("--draws", type=int, default=5_000_000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--orders", nargs="*", default=["generator", "random", "rcm", "degree"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
rowptr, col = rmat_csr(a.scale, a.draws, seed=1, device="cpu")
N = rowptr.numel() - 1
rp, cc = rowptr.numpy().astype(np.int64), col.numpy().astype(np.int64)
rows = np.repeat(np.arange(N), np.diff(rp))


def renumber(new_id):
    """CSR of the pattern with node v renamed new_id[v]."""
    r2, c2 = new_id[rows], new_id[cc]
    order = np.lexsort((c2, r2))
    r2, c2 = r2[order], c2[order]
    rp2 = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(r2, minlength=N), out=rp2[1:])
    return torch.from_numpy(rp2.astype(np.int32)), torch.from_numpy(c2.astype(np.int32))


def numbering(kind):
    if kind == "generator":
        return np.arange(N)
    if kind == "random":
        return np.random.default_rng(0).permutation(N)
    if kind == "degree":
        inv = np.argsort(-np.diff(rp), kind="stable")
    elif kind.startswith("hot_first_"):      # the K most-gathered nodes first, everything else in the generator's order
        K = int(kind.rsplit("_", 1)[1])
        deg = np.diff(rp)
        hot = np.zeros(N, dtype=bool); hot[np.argsort(-deg, kind="stable")[:K]] = True
        inv = np.concatenate([np.nonzero(hot)[0], np.nonzero(~hot)[0]])
    elif kind == "degree_bucket":            # by floor(log2(degree)) descending, the generator's order inside a bucket
        inv = np.argsort(-np.floor(np.log2(np.diff(rp))).astype(np.int64), kind="stable")
    elif kind == "cold_last":                # the self-loop-only nodes last, everything else in the generator's order
        inv = np.argsort((np.diff(rp) == 1).astype(np.int64), kind="stable")
    else:
        import scipy.sparse as sp
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        inv = reverse_cuthill_mckee(sp.csr_matrix((np.ones(len(cc), dtype=np.int8), cc, rp), shape=(N, N)), symmetric_mode=True)
    new_id = np.empty(N, dtype=np.int64)
    new_id[inv] = np.arange(N)
    return new_id


g = torch.Generator().manual_seed(2)
X = torch.randn(N, 128, generator=g).to(dev)
W = (torch.randn(8, 128, 16, generator=g) * 0.17).to(dev).requires_grad_(True)
av = (torch.randn(8, 32, generator=g) * 0.3).to(dev).requires_grad_(True)
G = torch.randn(N, 128, generator=g).to(dev)
for kind in a.orders:
    rp2, c2 = renumber(numbering(kind))
    graph = pg.CSRGraph(rp2.to(dev), c2.to(dev))

    def step():
        W.grad = av.grad = None
        pg.GATLevelFn.apply(X, W, av, None, graph, 0.2, True).backward(G)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.TIMER = timer
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ops.TIMER = None
    kt = {k: round(float(np.mean(v)), 4) for k, v in timer.times_ms().items()}
    print(json.dumps({"numbering": kind, "kernels_ms": kt, "sum_ms": round(sum(kt.values()), 4)}), flush=True)
    del graph
