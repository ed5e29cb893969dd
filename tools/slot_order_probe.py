#!/usr/bin/env python3
"""ONE locality experiment on the gather kernels, with a kill criterion (keep only if K2 + K4 drop by
>= 0.10 ms at config 5).  Nothing is renumbered and no table row moves: only the ORDER in which the main launches of K2 / K4 walk
the slots changes (pygat_graph.slot_order, a permutation built once per graph), i.e. which rows' gathers are in flight together:
  identity     grid order = slot order = CSR row order (the shipped default)
  random       a random permutation (what "no locality at all" costs)
  hubs_first   slots by descending degree of their first row (long rows first, the self-loop-only rows last)
  mean_col     slots by the mean neighbour id of their edges (concurrent waves gather from nearby table rows)
  min_col      slots by their smallest neighbour id
  xcd_blocks   work-groups are dealt round-robin to the 8 XCDs: give every XCD (= every L2) one contiguous eighth of the slots
  local_sort_W inside windows of W consecutive slots, by the number of rows ending in the slot (like-structured slots share a wave)
    python3 tools/slot_order_probe.py [--scale 20] [--orders ...]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygat_amd as pg  # noqa: E402
from pygat_amd import graph as G_, ops  # noqa: E402
from pygat_amd.rmat import rmat_csr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=20)
ap.add_argument("--draws", type=int, default=5_000_000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--orders", nargs="*", default=["identity", "random", "hubs_first", "mean_col", "min_col", "xcd_blocks", "identity"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
rowptr, col = rmat_csr(a.scale, a.draws, seed=1, device=dev)
N = rowptr.numel() - 1


def order_fn(kind):
    if kind == "identity":
        return None

    def f(pat, meta):
        ns = meta.shape[0]
        m = meta.long()
        if kind == "random":
            perm = torch.randperm(ns, generator=torch.Generator().manual_seed(0)).to(meta.device)
        elif kind == "hubs_first":
            rp = pat.rowptr.long()
            deg = rp[m[:, 2] + 1] - rp[m[:, 2]]
            perm = torch.argsort(deg, descending=True, stable=True)
        elif kind.startswith("local_sort_"):
            # inside every window of W consecutive slots: order by the number of rows that END in the slot (what a lane group pays
            # in row-finish branches), so that the lane groups of a wave -- and the waves of a work-group -- walk slots of like
            # structure, while the rows of a window stay neighbours in memory (PMC under a degree RENUMBERING: -9 % K2 time at
            # -3.6 % traffic: the gain there is less divergence, not fewer bytes)
            W = int(kind.rsplit("_", 1)[1])
            nxt = torch.cat([m[1:, 2], torch.tensor([pat.n], device=meta.device)])
            rows_in = (nxt - m[:, 2]).clamp(min=0)
            win = torch.arange(ns, device=meta.device) // W
            key = win * (1 << 20) + rows_in.clamp(max=(1 << 20) - 1)
            perm = torch.argsort(key, stable=True)
        elif kind in ("mean_col", "min_col"):
            e = torch.arange(pat.nnz, device=meta.device)
            sid = torch.searchsorted(m[:, 0].contiguous(), e, right=True) - 1
            cj = pat.col.long()
            if kind == "mean_col":
                key = torch.zeros(ns, device=meta.device, dtype=torch.float64).scatter_add_(0, sid, cj.double()) / (m[:, 1] - m[:, 0]).clamp(min=1)
            else:
                key = torch.full((ns,), 1 << 40, device=meta.device, dtype=torch.int64).scatter_reduce_(0, sid, cj, "amin")
            perm = torch.argsort(key, stable=True)
        else:   # xcd_blocks: grid position q -> work-group q // 8 (eight lane groups), XCD = work-group % 8
            q = torch.arange(ns, device=meta.device)
            wg, lg = q // 8, q % 8
            nwg = -(-ns // 8)
            per = -(-nwg // 8)
            slot = ((wg % 8) * per + wg // 8) * 8 + lg
            # positions whose slot falls past the end (ragged tail): hand them the unused ids in order
            ok = slot < ns
            used = torch.zeros(ns, dtype=torch.bool, device=meta.device)
            used[slot[ok]] = True
            free = torch.nonzero(~used).flatten()
            slot[~ok] = free[: int((~ok).sum())]
            perm = slot
        assert torch.equal(torch.sort(perm).values, torch.arange(ns, device=perm.device)), kind
        return perm.to(torch.int32).contiguous()
    return f


g = torch.Generator(device=dev).manual_seed(2)
X = torch.randn(N, 128, generator=g, device=dev)
W = (torch.randn(8, 128, 16, generator=g, device=dev) * 0.17).requires_grad_(True)
av = (torch.randn(8, 32, generator=g, device=dev) * 0.3).requires_grad_(True)
Gr = torch.randn(N, 128, generator=g, device=dev)
ref = None
for kind in a.orders:
    G_.SLOT_ORDER_FN = order_fn(kind)
    graph = pg.CSRGraph(rowptr, col)

    def step():
        W.grad = av.grad = None
        out = pg.GATLevelFn.apply(X, W, av, None, graph, 0.2, True)
        out.backward(Gr)
        return out
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    res = (out.detach().clone(), W.grad.clone(), av.grad.clone())
    if ref is None:
        ref = res
    same = all(torch.equal(x, y) for x, y in zip(res, ref))     # the order must not change a bit of the results
    timer = ops.KernelTimer()
    ops.TIMER = timer
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ops.TIMER = None
    kt = {k: round(float(np.mean(v)), 4) for k, v in timer.times_ms().items()}
    print(json.dumps({"slot_order": kind, "k2_ms": kt["k2_forward"], "k4_ms": kt["k4_backward_col"],
                      "k2_plus_k4_ms": round(kt["k2_forward"] + kt["k4_backward_col"], 4), "bitwise_equal_to_identity": same}), flush=True)
    del graph
G_.SLOT_ORDER_FN = None
